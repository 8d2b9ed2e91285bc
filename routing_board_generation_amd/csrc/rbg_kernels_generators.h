// rbg_kernels_generators.h -- the board generators as kernels: ParallelRandomWalk (jax stream and
// the opt-in Philox stream), SequentialRandomWalk, SeedExtension, UniformRandomGenerator.
#pragma once
#include "rbg_kernels_common.h"
#include "rbg_seqwalk.h"
#include "rbg_seedext.h"
#include "rbg_walk_philox.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Kernels
// ---------------------------------------------------------------------------------------------
struct GenArgs {
  const uint32_t *keys;
  int B, N, split_first;
  Geo g;
  int32_t *solved_grid;
  int32_t *train_grid;
  int32_t *start;
  int32_t *target;
  int32_t *position;
  int32_t *step_count;
  uint32_t *key_out;
  int32_t *iters;
  int seq_hist = 0;  // SequentialRandomWalk: LDS words of step history per board (0 = none)
};


#ifndef RBG_SEQ_HISTORY_BYTES
#define RBG_SEQ_HISTORY_BYTES (40 * 1024)  // per workgroup, boards included; 0 = never (tests)
#endif

// LPB lanes per board: 64 / LPB boards share one wavefront (the stencil keeps N lanes busy and
// the Threefry pipeline 2N + 2, so small N would leave most of a wave idle).
template <int LPB>
__global__ __launch_bounds__(kBlock) void prw_generate_kernel(GenArgs a) {
  constexpr int kSubs = SubWave<LPB>::kSubs;
  const SubWave<LPB> sw(threadIdx.x & (kWaveSize - 1));
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int first = (blockIdx.x * kWavesPerBlock + wib) * kSubs;
  if (first >= a.B) return;  // wave-uniform; waves of a block never synchronise with each other
#if defined(RBG_PROBE_TIMES)  // developer probe (tools/probe_walk.py --times): wave start / end
  const uint32_t probe_t0 = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
#if defined(RBG_PROBE_STAGE) && RBG_PROBE_STAGE == 0
  return;
#endif
  const Geo g = a.g;
  // boards past the end of the batch redo the wave's first board and store nothing
  const bool valid = first + sw.h < a.B;
  const int b = valid ? first + sw.h : first;
  uint32_t *board = g_lds + (wib * kSubs + sw.h) * slab_words<LPB>(g);
  uint32_t *W = board + board_words(g);

  Key2 key;
  key.a = a.keys[2 * (size_t)b];
  key.b = a.keys[2 * (size_t)b + 1];
  if (a.split_first) {  // key, pos_key = split(key)  (generator :52)
    Key2 unused;
    split2_lanes(sw, key, key, unused);
  }

  Walker w;
#if defined(RBG_PROBE_TIMES)  // stamps go to the solved_grid buffer (8 per wave); no solved grid
  w.stamp = a.solved_grid + (size_t)(blockIdx.x * kWavesPerBlock + wib) * 8;
  if (sw.lane == 0) w.stamp[0] = (int32_t)probe_t0;
  RBG_STAMP(w, 1);
#endif
  prw_walk(sw, board, W, g, a.N, key, w);
  if (!valid) return;

  const size_t gofs = (size_t)b * g.n;
#if !defined(RBG_PROBE_TIMES)
  if (a.solved_grid) store_grid<LPB, false>(board, g, sw.l, a.solved_grid + gofs);
#endif
  if (a.train_grid) store_grid<LPB, true>(board, g, sw.l, a.train_grid + gofs);
  if (sw.l < a.N) {
    const size_t o = ((size_t)b * a.N + sw.l) * 2;
    *reinterpret_cast<int2 *>(&a.start[o]) = make_int2(w.r0, w.c0);
    *reinterpret_cast<int2 *>(&a.target[o]) = make_int2(w.r, w.c);
    if (a.position) *reinterpret_cast<int2 *>(&a.position[o]) = make_int2(w.r0, w.c0);
  }
  if (sw.l == 0) {
    if (a.key_out) *reinterpret_cast<uint2 *>(&a.key_out[2 * (size_t)b]) = make_uint2(key.a, key.b);
    if (a.step_count) a.step_count[b] = 0;
#if defined(RBG_PROBE_HWID)  // developer probe (tools/probe_walk.py --placement): where the wave ran
    if (a.iters)
      a.iters[b] = (int)(((__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu) << 16) |
                         (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xFFFFu));
#elif defined(RBG_PROBE_TIMES)
    if (a.iters) a.iters[b] = w.iters;
    if (a.step_count)  // where the wave ran: XCC_ID << 16 | HW_ID[15:0]
      a.step_count[b] = (int)(((__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu) << 16) |
                              (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xFFFFu));
    RBG_STAMP(w, 7);
#else
    if (a.iters) a.iters[b] = w.iters;
#endif
  }
}

// SequentialRandomWalkBoard.generate / SequentialRandomWalkGenerator.__call__ (one board per wave)
__global__ __launch_bounds__(kBlock) void seq_rw_generate_kernel(GenArgs a) {
  const int lane = threadIdx.x & (kWaveSize - 1);
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x * kWavesPerBlock + wib;
  if (b >= a.B) return;
  const SubWave<64> sw(lane);
  const Geo g = a.g;
  uint32_t *board = g_lds + wib * slab_words<64>(g);
  Key2 key;
  key.a = a.keys[2 * (size_t)b];
  key.b = a.keys[2 * (size_t)b + 1];
  if (a.split_first) {  // key, pos_key = split(key)  (sequential_random_walk_generator.py:40)
    Key2 unused;
    split2_lanes(sw, key, key, unused);
  }
  int first_cell, last_cell;
  uint32_t *hist = a.seq_hist > 0
                       ? g_lds + kWavesPerBlock * slab_words<64>(g) + wib * a.seq_hist
                       : nullptr;
  const int attempts = seq_rw_generate(board, hist, g, a.N, key, lane, first_cell, last_cell);
  const size_t gofs = (size_t)b * g.n;
  if (a.solved_grid) store_grid<64, false>(board, g, lane, a.solved_grid + gofs);
  if (attempts == 0 && lane == 0) {
    // total failure: every start/end is (0, 0); the scatter order of duplicate indices is
    // unspecified in the reference -- here updates apply in agent order (last agent's target)
    board[pad_index(g, 0, 0)] = 3u * (uint32_t)a.N;
  }
  wave_lds_fence();
  if (a.train_grid) store_grid<64, true>(board, g, lane, a.train_grid + gofs);
  if (lane < a.N) {
    const int r0 = div_m(g, first_cell), r1 = div_m(g, last_cell);
    const int2 s2 = make_int2(r0, first_cell - r0 * g.m), t2 = make_int2(r1, last_cell - r1 * g.m);
    const size_t o = ((size_t)b * a.N + lane) * 2;
    *reinterpret_cast<int2 *>(&a.start[o]) = s2;
    *reinterpret_cast<int2 *>(&a.target[o]) = t2;
    if (a.position) *reinterpret_cast<int2 *>(&a.position[o]) = s2;
  }
  if (lane == 0) {
    if (a.key_out) *reinterpret_cast<uint2 *>(&a.key_out[2 * (size_t)b]) = make_uint2(key.a, key.b);
    if (a.step_count) a.step_count[b] = 0;
    if (a.iters) a.iters[b] = attempts;
  }
}


// ParallelRandomWalkGenerator(stream = "philox") (rbg_walk_philox.h): N lanes per board suffice
template <int LPB>
__global__ __launch_bounds__(kBlock) void prw_philox_generate_kernel(GenArgs a) {
  constexpr int kSubs = SubWave<LPB>::kSubs;
  const SubWave<LPB> sw(threadIdx.x & (kWaveSize - 1));
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int first = (blockIdx.x * kWavesPerBlock + wib) * kSubs;
  if (first >= a.B) return;
  const Geo g = a.g;
  const bool valid = first + sw.h < a.B;
  const int b = valid ? first + sw.h : first;
  uint32_t *board = g_lds + (wib * kSubs + sw.h) * slab_words<LPB>(g);
  uint32_t *W = board + board_words(g);
  Key2 key;
  key.a = a.keys[2 * (size_t)b];
  key.b = a.keys[2 * (size_t)b + 1];
  Walker w;
  prw_walk_philox(sw, board, W, g, a.N, key, w);
  if (!valid) return;
  const size_t gofs = (size_t)b * g.n;
  if (a.solved_grid) store_grid<LPB, false>(board, g, sw.l, a.solved_grid + gofs);
  if (a.train_grid) store_grid<LPB, true>(board, g, sw.l, a.train_grid + gofs);
  if (sw.l < a.N) {
    const size_t o = ((size_t)b * a.N + sw.l) * 2;
    *reinterpret_cast<int2 *>(&a.start[o]) = make_int2(w.r0, w.c0);
    *reinterpret_cast<int2 *>(&a.target[o]) = make_int2(w.r, w.c);
    if (a.position) *reinterpret_cast<int2 *>(&a.position[o]) = make_int2(w.r0, w.c0);
  }
  if (sw.l == 0) {
    if (a.key_out) {  // State.key (tag 3)
      const Philox4 p = philox4x32_10(key.a, key.b, 0u, 0u, 0u, 3u);
      *reinterpret_cast<uint2 *>(&a.key_out[2 * (size_t)b]) = make_uint2(p.w[0], p.w[1]);
    }
    if (a.step_count) a.step_count[b] = 0;
    if (a.iters) a.iters[b] = w.iters;
  }
}

struct SeArgs {
  GenArgs gen;  // iters = sweeps per board
  SeParams p;
};

// SeedExtensionBoard.generate_starts_ends / return_solved_board and
// SeedExtensionGenerator.__call__ (rbg_seedext.h); blockDim.x / 64 waves per workgroup
template <int LPB>
__global__ __launch_bounds__(kBlock) void seed_extension_kernel(SeArgs args) {
  constexpr int kSubs = SubWave<LPB>::kSubs;
  const GenArgs &a = args.gen;
  const SubWave<LPB> sw(threadIdx.x & (kWaveSize - 1));
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wpb = (int)(blockDim.x >> 6);
  const int first = (blockIdx.x * wpb + wib) * kSubs;
  if (first >= a.B) return;  // wave-uniform; waves never synchronise with each other
  const Geo g = a.g;
  // boards past the end of the batch redo the wave's first board and store nothing
  const bool valid = first + sw.h < a.B;
  const int b = valid ? first + sw.h : first;
  const int per_board = slab_words<LPB>(g) + args.p.extra_words;
  uint32_t *board = g_lds + (wib * kSubs + sw.h) * per_board;
  uint32_t *scr = board + board_words(g);
  uint32_t *xtra = board + slab_words<LPB>(g);

  Key2 key;
  key.a = a.keys[2 * (size_t)b];
  key.b = a.keys[2 * (size_t)b + 1];
  if (a.split_first) {  // key, pos_key = split(key)  (random_seed_generator.py:34)
    Key2 unused;
    split2_lanes(sw, key, key, unused);
  }
  int pr, pc, tr, tc;
  const int sweeps = seed_extension_generate(sw, board, scr, xtra, g, args.p, key, pr, pc, tr, tc);
  if (!valid) return;

  const size_t gofs = (size_t)b * g.n;
  if (a.solved_grid) store_grid<LPB, false>(board, g, sw.l, a.solved_grid + gofs);
  if (a.train_grid) store_grid<LPB, true>(board, g, sw.l, a.train_grid + gofs);
  if (sw.l < a.N) {
    const size_t o = ((size_t)b * a.N + sw.l) * 2;
    *reinterpret_cast<int2 *>(&a.start[o]) = make_int2(pr, pc);
    *reinterpret_cast<int2 *>(&a.target[o]) = make_int2(tr, tc);
    if (a.position) *reinterpret_cast<int2 *>(&a.position[o]) = make_int2(pr, pc);
  }
  if (sw.l == 0) {
    if (a.key_out) *reinterpret_cast<uint2 *>(&a.key_out[2 * (size_t)b]) = make_uint2(key.a, key.b);
    if (a.step_count) a.step_count[b] = 0;
    if (a.iters) a.iters[b] = sweeps;
  }
}

struct UniformArgs {
  const uint32_t *keys;
  int B, N;
  Geo g;
  int32_t *train_grid;
  int32_t *start;
  int32_t *target;
  int32_t *position;
  int32_t *step_count;
  uint32_t *key_out;
};

// UniformRandomGenerator.__call__ (uniform_generator.py:70-109)
__global__ __launch_bounds__(kBlock) void uniform_generate_kernel(UniformArgs a) {
  const int lane = threadIdx.x & (kWaveSize - 1);
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x * kWavesPerBlock + wib;
  if (b >= a.B) return;
  const SubWave<64> sw(lane);
  uint32_t *board = g_lds + wib * slab_words<64>(a.g);
  uint32_t *W = board + board_words(a.g);
  const Geo g = a.g;
  Key2 key;
  key.a = a.keys[2 * (size_t)b];
  key.b = a.keys[2 * (size_t)b + 1];
  Key2 kept, pos_key, unused, sub;
  split2_lanes(sw, key, kept, pos_key);
  split2_lanes(sw, pos_key, unused, sub);
  int sel0, sel1;
  shuffle_prefix(sw, board, W, g, sub, 2 * a.N, g.thr, sel0, sel1);
  // entry e of the prefix: e < N -> start of agent e, else target of agent e - N
  const int e_t = lane + a.N;  // prefix entry holding this agent's target
  const int t_lo = __shfl(sel0, e_t & 63), t_hi = __shfl(sel1, e_t & 63);
  const int tcell = e_t < kWaveSize ? t_lo : t_hi;
  board_clear<64>(board, g, lane);
  if (lane < a.N) {
    const int r0 = div_m(g, sel0), c0 = sel0 - r0 * g.m;
    const int tr = div_m(g, tcell), tc = tcell - tr * g.m;
    board[pad_index(g, r0, c0)] = 3u * lane + 2u;
    board[pad_index(g, tr, tc)] = 3u * lane + 3u;  // distinct cells: no ordering issue
    const size_t o = ((size_t)b * a.N + lane) * 2;
    *reinterpret_cast<int2 *>(&a.start[o]) = make_int2(r0, c0);
    *reinterpret_cast<int2 *>(&a.target[o]) = make_int2(tr, tc);
    if (a.position) *reinterpret_cast<int2 *>(&a.position[o]) = make_int2(r0, c0);
  }
  wave_lds_fence();
  if (a.train_grid) store_grid<64, false>(board, g, lane, a.train_grid + (size_t)b * g.n);
  if (lane == 0) {
    if (a.key_out) *reinterpret_cast<uint2 *>(&a.key_out[2 * (size_t)b]) = make_uint2(kept.a, kept.b);
    if (a.step_count) a.step_count[b] = 0;
  }
}

}  // namespace
