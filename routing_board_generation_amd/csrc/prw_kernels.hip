// prw_kernels.hip -- hand-written gfx950 (CDNA4) kernels + the C-ABI of include/rbg_prw.h.
//
// Device code lives in the headers next to this file:
//   rbg_device.h   jax's Threefry-2x32, sub-wave helpers, the padded LDS board
//   rbg_walk.h     ParallelRandomWalk (one board per sub-wave of 16/32/64 lanes)
//   rbg_env.h      Connector reset / step / observation pieces (one env per sub-wave)
//   rbg_seqwalk.h  SequentialRandomWalk (one board per wavefront)
//   rbg_walk_philox.h  the opt-in Philox stream of the walk (project-defined; N lanes per board)
//   rbg_seedext.h  SeedExtension (one board per sub-wave; chain phase + event phase + exact BFS)
// Common design:
//   * a board lives in LDS as a "padded" image (2 wall rows above/below, 4 wall columns shared
//     between consecutive rows) so that the 4-neighbour and neighbour-of-neighbour stencils need
//     no bounds checks;
//   * sub-lane i < N is agent i (availability stencil, weighted choice, move);
//   * the reference's "N private grid copies -> max-join -> correction mask" collision rule
//     (JAXI/parallel_random_walk.py:119-145) collapses to ONE `ds_max_u32` per mover on the
//     contested cell: the highest value (= highest agent id) wins, losers never left;
//   * termination is a wave ballot; no __syncthreads anywhere (waves never talk to each other);
//   * the PRNG is jax's Threefry-2x32 with jax's counter layout, in registers.
// HBM traffic is the algorithmic minimum: keys in, dense int32 grids / agents out, coalesced.
//
// No MFMA: this is integer stencil / indexing work.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/rbg_prw.h"
#include "rbg_device.h"
#include "rbg_walk.h"
#include "rbg_env.h"
#include "rbg_seqwalk.h"
#include "rbg_seedext.h"
#include "rbg_walk_philox.h"
#include "rbg_pieces.h"

namespace {

using namespace rbg;

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

extern __shared__ __attribute__((aligned(16))) uint32_t g_lds[];

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

#ifndef RBG_TU_RECORD  // (SequentialRandomWalk kernel: not part of the recording-rollout unit)
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

#endif  // !RBG_TU_RECORD

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
  const int per_board = slab_words<LPB>(g) + se_extra_words(g.words, g.n);
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

#ifndef RBG_TU_RECORD  // (not part of the recording-rollout unit)
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

#endif  // !RBG_TU_RECORD

// Pre-generated next-episode states (include/rbg_prw.h::rbg_stash).  `next` of env b is the
// State the auto-reset of an episode with State.key == tag[b] must produce.
struct StashPtrs {
  StatePtrs next;        // depth * B rows; slot s of env b = row s * B + b
  uint32_t *tag;         // [depth*B][2]
  int32_t *ready_at;     // [depth*B] first launch tick that may consume the row
  uint32_t *refill_key;  // [depth*B][2] the State.key whose successor the row should hold next
  int32_t *hits;         // [B] nullable diagnostics: resets served from the stash
  int depth;             // 1 or 2
};

struct EnvArgs {
  const uint32_t *keys;  // reset only
  int B, N;
  Geo g;
  StatePtrs in;   // step / observe
  StatePtrs out;
  const int32_t *action;
  rbg_env_params p;
  TimestepPtrs ts;
  int mode;  // 0 reset (generate), 1 observe existing state, 2 step
  StashPtrs stash;  // all NULL when resets are always generated inline
  int tick, ready_delay;
  const uint32_t *policy_keys;  // step only: non-NULL = the action comes from the masked-uniform
  uint32_t policy_t;            // random policy (rbg_random_actions) of global step policy_t
};

// Row of the stash that holds the successor of the episode with State.key `key`, or -1.
__device__ __forceinline__ int stash_lookup(const StashPtrs &st, int B, int b, Key2 key,
                                            int tick) {
  int row = -1;
  if (st.tag) {
    for (int s = 0; s < st.depth; ++s) {
      const int r = s * B + b;
      const uint2 tag = *reinterpret_cast<const uint2 *>(&st.tag[2 * (size_t)r]);
      if (tick >= st.ready_at[r] && tag.x == key.a && tag.y == key.b) row = r;
    }
  }
  return row;
}

// The same lookup with its loads issued early: lane s < depth fetches (tag, ready_at) of row
// s * B + b when the wave starts, so that an env that ends its episode a microsecond later finds
// the answer in registers instead of starting two dependent global loads on the launch's tail.
struct StashProbe {
  uint2 tag;
  int ready_at;
};
__device__ __forceinline__ StashProbe stash_probe_issue(const StashPtrs &st, int B, int b,
                                                        int lane) {
  StashProbe p = {make_uint2(0u, 0u), 0x7FFFFFFF};
  if (st.tag && lane < st.depth) {
    const int r = lane * B + b;
    p.tag = *reinterpret_cast<const uint2 *>(&st.tag[2 * (size_t)r]);
    p.ready_at = st.ready_at[r];
  }
  return p;
}
__device__ __forceinline__ int stash_probe_row(const StashPtrs &st, const StashProbe &p, int B,
                                               int b, int lane, Key2 key, int tick) {
  const bool ok = st.tag && lane < st.depth && tick >= p.ready_at && p.tag.x == key.a &&
                  p.tag.y == key.b;
  const uint64_t m = __ballot(ok) & 3ull;  // depth <= 2; the later slot wins, as in stash_lookup
  return m == 0ull ? -1 : (m & 2ull ? B + b : b);
}

// An episode that ends on the time limit is known to end as soon as step_count is loaded: its
// wave fetches the stash row while it steps (registers only; one dwordx4 per lane covers grids
// up to 16x16) and finds the next episode's State ready when it gets there.  Other terminations
// (every agent connected or blocked) take the plain load_state path.
struct RowPrefetch {
  int row;  // -1: nothing prefetched
  int4 grid;
  int2 s, t, p;
  uint2 key;
};
__device__ __forceinline__ RowPrefetch stash_row_issue(const StatePtrs &next, const Geo &g, int N,
                                                       int row, int lane) {
  RowPrefetch pf;
  pf.row = row;
  pf.grid = make_int4(0, 0, 0, 0);
  pf.s = pf.t = pf.p = make_int2(0, 0);
  pf.key = make_uint2(0u, 0u);
  if (row >= 0) {
    if (lane * 4 < g.n)
      pf.grid = *reinterpret_cast<const int4 *>(&next.grid[(size_t)row * g.n + lane * 4]);
    if (lane < N) {
      const size_t o = ((size_t)row * N + lane) * 2;
      pf.s = *reinterpret_cast<const int2 *>(&next.start[o]);
      pf.t = *reinterpret_cast<const int2 *>(&next.target[o]);
      pf.p = *reinterpret_cast<const int2 *>(&next.position[o]);
    }
    pf.key = *reinterpret_cast<const uint2 *>(&next.key[2 * (size_t)row]);
  }
  return pf;
}
__device__ __forceinline__ void stash_row_commit(const SubWave<64> &sw, uint32_t *board,
                                                 const Geo &g, const RowPrefetch &pf,
                                                 AgentRegs &ag, Key2 &key) {
  const int lane = sw.l;
  for (int w = lane; w < g.words; w += kWaveSize) board[w] = kWall;
  wave_lds_fence();
  if (lane * 4 < g.n) {
    const int r = div_m(g, lane * 4);
    *reinterpret_cast<uint4 *>(&board[pad_index(g, r, lane * 4 - r * g.m)]) =
        make_uint4((uint32_t)pf.grid.x, (uint32_t)pf.grid.y, (uint32_t)pf.grid.z,
                   (uint32_t)pf.grid.w);
  }
  wave_lds_fence();
  ag.r0 = pf.s.x; ag.c0 = pf.s.y; ag.tr = pf.t.x; ag.tc = pf.t.y; ag.r = pf.p.x; ag.c = pf.p.y;
  key.a = pf.key.x;
  key.b = pf.key.y;
}

// After a (re)start with State.key `key`: ask the next refill for what comes after it
// (include/rbg_prw.h::rbg_stash).  depth 1: the env's one row is requested for `key`.  depth 2:
// one row is reserved for `key` unless one already is, and the other -- just consumed, or never
// used -- is requested for the key of the episode after the next, split(split(key)[0])[0]
// (VmapAutoResetWrapper's split, then the generator's).  Called by the whole sub-wave.
template <int LPB>
__device__ __forceinline__ void request_refill(const SubWave<LPB> &sw, const StashPtrs &st, int B,
                                               int b, Key2 key, int ready_at) {
  if (!st.refill_key) return;
  if (st.depth == 1) {
    if (sw.l == 0) {
      *reinterpret_cast<uint2 *>(&st.refill_key[2 * (size_t)b]) = make_uint2(key.a, key.b);
      st.ready_at[b] = ready_at;
    }
    return;
  }
  Key2 rk, after, unused;
  split2_lanes(sw, key, rk, unused);
  split2_lanes(sw, rk, after, unused);
  if (sw.l == 0) {
    const int r0 = b, r1 = B + b;
    const uint2 k0 = *reinterpret_cast<const uint2 *>(&st.refill_key[2 * (size_t)r0]);
    const uint2 k1 = *reinterpret_cast<const uint2 *>(&st.refill_key[2 * (size_t)r1]);
    int other = r1;
    if (k0.x == key.a && k0.y == key.b) {
      other = r1;
    } else if (k1.x == key.a && k1.y == key.b) {
      other = r0;
    } else {
      *reinterpret_cast<uint2 *>(&st.refill_key[2 * (size_t)r0]) = make_uint2(key.a, key.b);
      st.ready_at[r0] = ready_at;
    }
    *reinterpret_cast<uint2 *>(&st.refill_key[2 * (size_t)other]) = make_uint2(after.a, after.b);
    st.ready_at[other] = ready_at;
  }
}

#ifndef RBG_ENV_WPB
#define RBG_ENV_WPB 4
#endif
constexpr int kEnvWavesPerBlock = RBG_ENV_WPB;  // waves (envs) per workgroup of the env kernels

// One body, three kernels (reset / observe / step) so that every launch population has its own
// name in a rocprofv3 kernel trace.
enum { kModeReset = 0, kModeObserve = 1, kModeStep = 2 };

template <int MODE>
__device__ __forceinline__ void connector_body(const EnvArgs &a) {
#if defined(RBG_PROBE_STEP_TIMES)  // developer probe (tools/probe_step.py --times)
  const uint32_t probe_t0 = (uint32_t)__builtin_amdgcn_s_memrealtime();  // 100 MHz, chip-wide
#endif
  const int lane = threadIdx.x & (kWaveSize - 1);
  const SubWave<64> sw(lane);
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x * kEnvWavesPerBlock + wib;
  if (b >= a.B) return;
  uint32_t *board = g_lds + wib * slab_words<64>(a.g);
  uint32_t *W = board + board_words(a.g);
  const Geo g = a.g;
  const int N = a.N;
  AgentRegs ag = {0, 0, 0, 0, 0, 0};

  if constexpr (MODE == kModeReset) {  // ---- Connector.reset ----------------------------------
    Key2 kin;
    kin.a = a.keys[2 * (size_t)b];
    kin.b = a.keys[2 * (size_t)b + 1];
    const Key2 key = regenerate(sw, board, W, g, N, kin, ag);
    store_state(sw, board, g, N, b, ag, 0, key, a.out);
    request_refill(sw, a.stash, a.B, b, key, a.tick + a.ready_delay);
    write_observation(sw, board, g, N, b, ag, 0, a.ts,
                      lane < N ? action_mask_bits(board, g, ag, lane) : 1u);
    write_extras(sw, board, g, N, b, ag, a.ts);
    write_restart(N, b, lane, a.ts);
    return;
  }

  int step_count;
  Key2 key;
  load_state(sw, board, g, N, b, a.in, ag, step_count, key);

  if constexpr (MODE == kModeObserve) {  // ---- observe an existing state (restart timestep) ---
    write_observation(sw, board, g, N, b, ag, step_count, a.ts,
                      lane < N ? action_mask_bits(board, g, ag, lane) : 1u);
    write_extras(sw, board, g, N, b, ag, a.ts);
    write_restart(N, b, lane, a.ts);
    return;
  }

  // ---- Connector.step --------------------------------------------------------------------------
  int act = 0;
  if (a.policy_keys) {
    // policy fused into the step launch: the mask rbg_random_actions would read from the previous
    // timestep is the mask of the state just loaded
    Key2 pk;
    pk.a = a.policy_keys[2 * (size_t)b];
    pk.b = a.policy_keys[2 * (size_t)b + 1];
    if (lane < N)
      act = random_policy_action(pk, a.policy_t, (uint32_t)lane, action_mask_bits(board, g, ag, lane));
  } else if (lane < N) {
    act = a.action[(size_t)b * N + lane];
  }
  const StashProbe probe = stash_probe_issue(a.stash, a.B, b, lane);  // in flight during the step
  RowPrefetch pf;
  pf.row = -1;
  if (a.p.auto_reset && a.stash.tag && step_count + 1 >= a.p.time_limit && (g.m & 3) == 0 &&
      g.n <= 4 * kWaveSize)  // wave-uniform
    pf = stash_row_issue(a.stash.next, g, N,
                         stash_probe_row(a.stash, probe, a.B, b, lane, key, a.tick), lane);
#if defined(RBG_PROBE_SKIP_STEP)  // developer probe: load -> store only (wrong results)
  StepResult res;
  res.was_connected = res.connected = res.done = res.last = false;
  res.bits = 1u + (uint32_t)act;
#else
  const StepResult res = env_step_core(sw, board, g, N, ag, step_count, act, a.p.time_limit);
#endif

  if (!(res.last && a.p.auto_reset)) {
    // The common case.  Every wave of a launch reaches this point at about the same time, and
    // until the first of them stores, the HBM write path -- what bounds the launch -- is idle:
    // the 10 KB observation goes out first, then the state, and the small outputs (reward,
    // extras: ballots and a pass over the board) are computed while those stores drain.
    write_observation(sw, board, g, N, b, ag, step_count, a.ts, res.bits);
    store_state(sw, board, g, N, b, ag, step_count, key, a.out);
    write_step_outcome(sw, res, a.p, N, b, a.ts);
    write_extras(sw, board, g, N, b, ag, a.ts);
  } else {
    // the few waves that restart an episode have the longest path through this kernel and
    // every wave of the launch started together: let them go first wherever they compete
    __builtin_amdgcn_s_setprio(3);
    write_step_outcome(sw, res, a.p, N, b, a.ts);
    write_extras(sw, board, g, N, b, ag, a.ts);  // of the terminal state
    // VmapAutoResetWrapper._auto_reset: key, _ = split(state.key); state, ts = reset(key);
    // only the observation of the terminal timestep is replaced.
    // With a stash the new episode's State was generated ahead of time (stash_refill_kernel)
    // and is only copied here; the entry is valid iff it was generated for exactly this
    // episode's key and the refill that wrote it is known to have completed (ready_at: ordered
    // by the host through the stream, never through in-kernel flags).
    const int row = pf.row >= 0 ? pf.row
                                : stash_probe_row(a.stash, probe, a.B, b, lane, key, a.tick);
    if (pf.row >= 0) {
      stash_row_commit(sw, board, g, pf, ag, key);
      if (a.stash.hits && lane == 0) a.stash.hits[b] += 1;
    } else if (row >= 0) {
      load_state(sw, board, g, N, row, a.stash.next, ag, step_count, key);
      if (a.stash.hits && lane == 0) a.stash.hits[b] += 1;
    } else {
      Key2 rk, unused;
      split2_lanes(sw, key, rk, unused);
      key = regenerate(sw, board, W, g, N, rk, ag);
    }
    step_count = 0;
    request_refill(sw, a.stash, a.B, b, key, a.tick + a.ready_delay);
    const uint32_t bits = lane < N ? action_mask_bits(board, g, ag, lane) : 1u;
    store_state(sw, board, g, N, b, ag, step_count, key, a.out);
    write_observation(sw, board, g, N, b, ag, step_count, a.ts, bits);
  }
#if defined(RBG_PROBE_STEP_TIMES)
  // wave start / end (10 ns ticks) and where it ran, in place of three extras (probe builds only)
  __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): the stores have left the wave
  if (lane == 0) {
    a.ts.num_connections[b] = (int32_t)probe_t0;
    a.ts.total_path_length[b] = (int32_t)(uint32_t)__builtin_amdgcn_s_memrealtime();
    a.ts.obs_step_count[b] = (int)(((__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu) << 16) |
                                   (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xFFFFu));
  }
#endif
}

#ifndef RBG_TU_RECORD  // (not part of the recording-rollout unit)
__global__ __launch_bounds__(kWaveSize * kEnvWavesPerBlock) void connector_reset_kernel(EnvArgs a) {
  connector_body<kModeReset>(a);
}
__global__ __launch_bounds__(kWaveSize * kEnvWavesPerBlock) void connector_observe_kernel(EnvArgs a) {
  connector_body<kModeObserve>(a);
}
__global__ __launch_bounds__(kWaveSize * kEnvWavesPerBlock) void connector_step_kernel(EnvArgs a) {
  connector_body<kModeStep>(a);
}

#endif  // !RBG_TU_RECORD

// Producer of the stash: for every env whose entry is not the one its running episode will need
// (tag != refill_key), generate G(split(refill_key)[0]) and publish it.  One env per sub-wave
// (two per wave at 16x16/10); sub-waves with nothing to do retire at once.
struct RefillArgs {
  int B, N;
  Geo g;
  StashPtrs stash;
};

template <int LPB>
__global__ __launch_bounds__(kBlock) void stash_refill_kernel(RefillArgs a) {
  constexpr int kSubs = SubWave<LPB>::kSubs;
  const SubWave<LPB> sw(threadIdx.x & (kWaveSize - 1));
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = (blockIdx.x * kWavesPerBlock + wib) * kSubs + sw.h;
  if (b >= a.B) return;
  const uint2 want = *reinterpret_cast<const uint2 *>(&a.stash.refill_key[2 * (size_t)b]);
  const uint2 have = *reinterpret_cast<const uint2 *>(&a.stash.tag[2 * (size_t)b]);
  if (want.x == have.x && want.y == have.y) return;  // nothing to do for this env
  const Geo g = a.g;
  uint32_t *board = g_lds + (wib * kSubs + sw.h) * slab_words<LPB>(g);
  uint32_t *W = board + board_words(g);
  AgentRegs ag = {0, 0, 0, 0, 0, 0};
  Key2 key = {want.x, want.y}, rk, unused;
  split2_lanes(sw, key, rk, unused);  // VmapAutoResetWrapper: key, _ = split(state.key)
  const Key2 next_key = regenerate(sw, board, W, g, a.N, rk, ag);
  store_state(sw, board, g, a.N, b, ag, 0, next_key, a.stash.next);
  if (sw.l == 0) *reinterpret_cast<uint2 *>(&a.stash.tag[2 * (size_t)b]) = want;
}

// ---------------------------------------------------------------------------------------------
// Random-policy rollout: n_steps of VmapAutoResetWrapper(Connector).step under the masked-uniform
// policy in ONE launch.  The env stays in LDS/registers between steps, so HBM sees the state once
// in and once out (+ a few counters) instead of ~13 KB per env-step, and envs that regenerate do
// so while the others keep stepping (no per-step launch boundary to wait at).
// ---------------------------------------------------------------------------------------------
struct RolloutStatsPtrs {
  int32_t *episodes;
  int32_t *connections_made;
  int32_t *unconnected_agent_steps;
  int32_t *final_connections;
  int32_t *final_path_length;
};

struct RolloutArgs {
  int B, N;
  Geo g;
  StatePtrs in, out;
  const uint32_t *policy_keys;
  uint32_t t0;
  int n_steps;
  rbg_env_params p;
  RolloutStatsPtrs stats;
  TimestepPtrs ts;  // timestep of the LAST step of the rollout (all members nullable)
  StashPtrs stash;  // optional next-episode stash
  int tick, ready_delay;
  TimestepPtrs traj;     // kRecord: every step's timestep, arrays [n_steps][B]... (members nullable)
  int32_t *traj_action;  // kRecord: the actions taken, [n_steps][B][N] (nullable)
};

// One env per sub-wave of LPB lanes (the host picks the smallest LPB that holds the walk's 2N + 2
// Threefry slots, like the generator): at 16x16/10 a wave steps two envs with one instruction
// stream.  Per-env values (step_count, keys, counters) live in VGPRs, uniform inside the sub-wave;
// an env that finishes an episode regenerates under its sub-wave's exec mask while the other
// lanes of the wave wait -- every cross-lane operation of the env and walk code stays inside a
// sub-wave, so that divergence is legal.
// (LPB 64: 4 waves per SIMD = one launch of 4096 envs resident at once, VGPR budget 128.)
#ifndef RBG_ROLLOUT64_WAVES
#define RBG_ROLLOUT64_WAVES 4
#endif
// Per-env values are uniform inside the sub-wave; for a full-wave env that is wave-uniform, and
// saying so moves them to SGPRs (the 128-VGPR budget of 4 waves per SIMD is tight: without this
// connector_rollout_kernel<64> spilled 8 dwords).
template <int LPB>
__device__ __forceinline__ int uni(int v) {
  if constexpr (LPB == kWaveSize) return __builtin_amdgcn_readfirstlane(v);
  return v;
}
template <int LPB>
__device__ __forceinline__ Key2 uni(Key2 k) {
  if constexpr (LPB == kWaveSize) {
    k.a = (uint32_t)__builtin_amdgcn_readfirstlane((int)k.a);
    k.b = (uint32_t)__builtin_amdgcn_readfirstlane((int)k.b);
  }
  return k;
}
// kRecord: the launch also writes the timestep of EVERY step (the trajectory an acting loop
// collects: observation, mask, reward, discount, step_type, extras, action) -- n_steps x 11.7 KB
// per env at 16x16/10, which makes it an HBM stream instead of an LDS-resident loop.
// (a recording full-wave env needs a few registers more than the 128 of four waves per SIMD
// leave; it is HBM-bound, so three waves per SIMD cost it nothing)
template <int LPB, bool kRecord>
__global__ __launch_bounds__(kBlock, (LPB == kWaveSize ? (kRecord ? 3 : RBG_ROLLOUT64_WAVES) : 2)) void connector_rollout_kernel(
    RolloutArgs a) {
  constexpr int kSubs = SubWave<LPB>::kSubs;
  const SubWave<LPB> sw(threadIdx.x & (kWaveSize - 1));
  const int l = sw.l;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = (blockIdx.x * kWavesPerBlock + wib) * kSubs + sw.h;
  if (b >= a.B) return;  // per sub-wave: the rest of the wave carries on
  const Geo g = a.g;
  uint32_t *board = g_lds + (wib * kSubs + sw.h) * slab_words<LPB>(g);
  uint32_t *W = board + board_words(g);
  const int N = a.N;
  const bool is_agent = l < N;
  AgentRegs ag = {0, 0, 0, 0, 0, 0};
  int step_count;
  Key2 key, pk;
  load_state(sw, board, g, N, b, a.in, ag, step_count, key);
  pk.a = a.policy_keys[2 * (size_t)b];
  pk.b = a.policy_keys[2 * (size_t)b + 1];
  step_count = uni<LPB>(step_count);
  key = uni<LPB>(key);
  pk = uni<LPB>(pk);

  int episodes = 0, conn_made = 0, pen_steps = 0, fin_conn = 0, fin_plen = 0;
  uint32_t bits = is_agent ? action_mask_bits(board, g, ag, l) : 1u;
  // the policy's random word does not depend on the env: the one of step s + 1 is computed
  // inside step s (rbg_env.h::PolicyFill)
  PolicyFill pf;
  pf.k = pk;
  pf.i = (uint32_t)l;
  pf.t = a.t0;
  pf.first();
  pf.second();
  for (int s = 0; s < a.n_steps; ++s) {
    const int act = policy_pick(pf.y0, bits);
    pf.t = a.t0 + (uint32_t)s + 1u;
    const StepResult res =
        env_step_core(sw, board, g, N, ag, step_count, act, a.p.time_limit, pf);
    bits = res.bits;
    step_count = uni<LPB>(step_count);
    conn_made = uni<LPB>(conn_made + __popcll(sw.ballot(!res.was_connected && res.connected)));
    pen_steps = uni<LPB>(pen_steps + __popcll(sw.ballot(is_agent && !res.was_connected)));
    const bool final_step = s == a.n_steps - 1;
    if (final_step) {
      write_step_outcome(sw, res, a.p, N, b, a.ts);
      write_extras(sw, board, g, N, b, ag, a.ts);
    }
    const int row = s * a.B + b;  // of the trajectory arrays
    if constexpr (kRecord) {
      __builtin_amdgcn_sched_barrier(0);  // (keeps the writers' address arithmetic out of the step)
      if (a.traj_action && is_agent) a.traj_action[(size_t)row * N + l] = act;
      write_step_outcome(sw, res, a.p, N, row, a.traj);
      write_extras(sw, board, g, N, row, ag, a.traj);  // of the terminal state when it is one
      __builtin_amdgcn_sched_barrier(0);
    }
    if (res.last) {  // per env
      episodes = uni<LPB>(episodes + 1);
      fin_conn = uni<LPB>(fin_conn + __popcll(sw.ballot(res.connected)));
      if (a.stats.final_path_length) fin_plen = uni<LPB>(fin_plen + count_path_cells(sw, board, g));
      if (a.p.auto_reset) {
        // (rows do not change during a launch: a second reset finds its own row or none)
        const int row = stash_lookup(a.stash, a.B, b, key, a.tick);
        if (row >= 0) {
          load_state(sw, board, g, N, row, a.stash.next, ag, step_count, key);
          if (a.stash.hits && l == 0) a.stash.hits[b] += 1;
        } else {
          Key2 rk, unused;
          split2_lanes(sw, key, rk, unused);
          key = regenerate(sw, board, W, g, N, rk, ag);
        }
        step_count = 0;
        key = uni<LPB>(key);
        request_refill(sw, a.stash, a.B, b, key, a.tick + a.ready_delay);
        bits = is_agent ? action_mask_bits(board, g, ag, l) : 1u;
      }
    }
    // (after an auto-reset: the observation of the episode that starts, like the wrapper's)
    if constexpr (kRecord) {
      __builtin_amdgcn_sched_barrier(0);
      write_observation(sw, board, g, N, row, ag, step_count, a.traj, bits);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  store_state(sw, board, g, N, b, ag, step_count, key, a.out);
  write_observation(sw, board, g, N, b, ag, step_count, a.ts, bits);
  if (l == 0) {
    if (a.stats.episodes) a.stats.episodes[b] = episodes;
    if (a.stats.connections_made) a.stats.connections_made[b] = conn_made;
    if (a.stats.unconnected_agent_steps) a.stats.unconnected_agent_steps[b] = pen_steps;
    if (a.stats.final_connections) a.stats.final_connections[b] = fin_conn;
    if (a.stats.final_path_length) a.stats.final_path_length[b] = fin_plen;
  }
}

#ifndef RBG_TU_RECORD  // (not part of the recording-rollout unit)
// BoardDatasetGeneratorJAX.__call__ (rl_training/offline_generation/dataset_generator_jax.py
// :112-141): key, _ = split(key); which = randint(key, (), 0, n_boards); State from the stored
// heads/targets of board `which`.  [3P] jax/_src/random.py::_randint: k1, k2 = split(key);
// hi, lo = random_bits(k1), random_bits(k2); offset = ((hi % span) * mult + lo % span) % span in
// wrapping uint32 arithmetic, mult = ((2^16 % span)^2) % span (host-computed).
struct DatasetArgs {
  const uint32_t *keys;
  int B, N, n_boards;
  uint32_t mult;
  Geo g;
  const int32_t *heads;    // [n_boards][N][2]
  const int32_t *targets;  // [n_boards][N][2]
  StatePtrs out;
  int32_t *which;
};

__global__ __launch_bounds__(kBlock) void dataset_sample_kernel(DatasetArgs a) {
  const int lane = threadIdx.x & (kWaveSize - 1);
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x * kWavesPerBlock + wib;
  if (b >= a.B) return;
  const SubWave<64> sw(lane);
  const Geo g = a.g;
  uint32_t *board = g_lds + wib * slab_words<64>(g);
  Key2 key, k0, unused, k1, k2;
  key.a = a.keys[2 * (size_t)b];
  key.b = a.keys[2 * (size_t)b + 1];
  split2_lanes(sw, key, k0, unused);
  split2_lanes(sw, k0, k1, k2);
  const uint2 hi = threefry_call(k1.a, k1.b, 0u, 0u), lo = threefry_call(k2.a, k2.b, 0u, 0u);
  const uint32_t span = (uint32_t)a.n_boards;
  const uint32_t which = ((hi.x % span) * a.mult + (lo.x % span)) % span;
  board_clear<64>(board, g, lane);
  AgentRegs ag = {0, 0, 0, 0, 0, 0};
  if (lane < a.N) {
    const size_t o = ((size_t)which * a.N + lane) * 2;
    const int2 h = *reinterpret_cast<const int2 *>(&a.heads[o]);
    const int2 t = *reinterpret_cast<const int2 *>(&a.targets[o]);
    ag.r0 = ag.r = h.x; ag.c0 = ag.c = h.y; ag.tr = t.x; ag.tc = t.y;
    board[pad_index(g, h.x, h.y)] = 3u * lane + 2u;
  }
  wave_lds_fence();
  if (lane < a.N) board[pad_index(g, ag.tr, ag.tc)] = 3u * lane + 3u;  // targets written last
  wave_lds_fence();
  store_state(sw, board, g, a.N, b, ag, 0, k0, a.out);
  if (lane == 0 && a.which) a.which[b] = (int32_t)which;
}

// split(key, num) for one key: thread t writes child t
__global__ void split_kernel(uint32_t k0, uint32_t k1, int num, uint32_t *__restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= num) return;
  Key2 k = {k0, k1};
  const Key2 c = split_child(k, (uint32_t)t, (uint32_t)num);
  *reinterpret_cast<uint2 *>(&out[2 * (size_t)t]) = make_uint2(c.a, c.b);
}

__global__ void split_batch_kernel(const uint32_t *__restrict__ keys, int B, int num,
                                   uint32_t *__restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)B * num) return;
  const size_t b = t / num;
  const uint32_t i = (uint32_t)(t - b * num);
  Key2 k = {keys[2 * b], keys[2 * b + 1]};
  const Key2 c = split_child(k, i, (uint32_t)num);
  *reinterpret_cast<uint2 *>(&out[2 * t]) = make_uint2(c.a, c.b);
}

// masked-uniform random policy (defined by this project; see include/rbg_prw.h)
__global__ void random_actions_kernel(const uint32_t *__restrict__ keys, int B, int N, uint32_t t,
                                      const uint8_t *__restrict__ mask,
                                      int32_t *__restrict__ action) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)B * N) return;
  const size_t b = idx / N;
  const uint32_t i = (uint32_t)(idx - b * N);
  const Key2 k = {keys[2 * b], keys[2 * b + 1]};
  const uint8_t *mk = mask + idx * 5;
  uint32_t bits = 0u;
#pragma unroll
  for (int a = 0; a < 5; ++a) bits |= (uint32_t)(mk[a] != 0) << a;
  action[idx] = random_policy_action(k, t, i, bits);
}

// ---------------------------------------------------------------------------------------------
// The walk taken apart (rbg_pieces.h): the methods of ParallelRandomWalkBoard that the reference's
// unit tests call one by one.  Not on the hot path; one wave per board or one thread per query.
// ---------------------------------------------------------------------------------------------
struct PiecesArgs {
  const uint32_t *keys;
  int B, N;
  Geo g;
  const int32_t *grid_in;
  const int32_t *target;
  const int32_t *position;
  uint32_t *key_out;
  int32_t *grid_out;
  int32_t *position_out;
  int32_t *action_out;
};

// _initialise_agents(key, grid) (:147-186): start cells = first N of permutation(key, m * m);
// every agent sets its head on a private copy of `grid`, the copies are max-joined
__global__ __launch_bounds__(kBlock) void prw_initialise_agents_kernel(PiecesArgs a) {
  const int lane = threadIdx.x & (kWaveSize - 1);
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x * kWavesPerBlock + wib;
  if (b >= a.B) return;
  const SubWave<64> sw(lane);
  const Geo g = a.g;
  uint32_t *board = g_lds + wib * slab_words<64>(g);
  uint32_t *W = board + board_words(g);
  Key2 key = {a.keys[2 * (size_t)b], a.keys[2 * (size_t)b + 1]}, unused, sub;
  split2_lanes(sw, key, unused, sub);  // _shuffle: key, subkey = split(key)
  int sel0, sel1;
  shuffle_prefix(sw, board, W, g, sub, a.N, g.thr, sel0, sel1);
  if (a.grid_in) load_grid(sw, board, g, a.grid_in + (size_t)b * g.n);
  else board_clear<64>(board, g, lane);
  if (lane < a.N) {
    const int r = div_m(g, sel0), c = sel0 - r * g.m;
    const uint32_t head = 3u * (uint32_t)lane + 2u;
    uint32_t *cell = &board[pad_index(g, r, c)];
    // the other agents' copies still hold the old value: it survives the max when it is larger
    *cell = (a.N > 1 && (int)*cell > (int)head) ? *cell : head;
    *reinterpret_cast<int2 *>(&a.position_out[((size_t)b * a.N + lane) * 2]) = make_int2(r, c);
  }
  wave_lds_fence();
  store_grid<64, false>(board, g, lane, a.grid_out + (size_t)b * g.n);
}

// _step((key, grid, agents)) (:92-145): keys = split(key, N); actions by _select_action; every
// agent steps on a private grid, the grids are max-joined, colliding agents with the lower id
// stay (env_step_core is that rule for given actions); next_key = split(key)[1]
__global__ __launch_bounds__(kBlock) void prw_step_kernel(PiecesArgs a) {
  const int lane = threadIdx.x & (kWaveSize - 1);
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x * kWavesPerBlock + wib;
  if (b >= a.B) return;
  const SubWave<64> sw(lane);
  const Geo g = a.g;
  uint32_t *board = g_lds + wib * slab_words<64>(g);
  const int m = g.m;
  AgentRegs ag = {0, 0, -1, -1, 0, 0};
  if (lane < a.N) {
    const size_t o = ((size_t)b * a.N + lane) * 2;
    const int2 p = *reinterpret_cast<const int2 *>(&a.position[o]);
    ag.r = p.x < 0 ? 0 : p.x >= m ? m - 1 : p.x;
    ag.c = p.y < 0 ? 0 : p.y >= m ? m - 1 : p.y;
    if (a.target) {
      const int2 t = *reinterpret_cast<const int2 *>(&a.target[o]);
      ag.tr = t.x;
      ag.tc = t.y;
    }
  }
  load_grid(sw, board, g, a.grid_in + (size_t)b * g.n);
  const Key2 key = {a.keys[2 * (size_t)b], a.keys[2 * (size_t)b + 1]};
  int act = 0;
  if (lane < a.N) {
    const LdsGrid grid = {board, g.S};
    act = ref_select_action(split_child(key, (uint32_t)lane, (uint32_t)a.N), grid, m,
                            ag.r * m + ag.c);
    if (a.action_out) a.action_out[(size_t)b * a.N + lane] = act;
  }
  int step_count = 0;
  env_step_core<64>(sw, board, g, a.N, ag, step_count, act, 0x7FFFFFFF);
  store_grid<64, false>(board, g, lane, a.grid_out + (size_t)b * g.n);
  if (lane < a.N)
    *reinterpret_cast<int2 *>(&a.position_out[((size_t)b * a.N + lane) * 2]) = make_int2(ag.r, ag.c);
  Key2 k0, k1;
  split2_lanes(sw, key, k0, k1);
  if (lane == 0 && a.key_out)
    *reinterpret_cast<uint2 *>(&a.key_out[2 * (size_t)b]) = make_uint2(k1.a, k1.b);
}

// _select_action(key, grid, agent) for every agent of every board, one key per agent
__global__ void prw_select_action_kernel(const uint32_t *__restrict__ agent_keys, int B, int m,
                                         int N, const int32_t *__restrict__ grid,
                                         const int32_t *__restrict__ position,
                                         int32_t *__restrict__ action) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)B * N) return;
  const size_t b = idx / N;
  const HbmGrid g = {grid + b * (size_t)(m * m), m};
  const Key2 key = {agent_keys[2 * idx], agent_keys[2 * idx + 1]};
  action[idx] = ref_select_action(key, g, m, position[2 * idx] * m + position[2 * idx + 1]);
}

// _adjacent_cells / _available_cells / _is_cell_free / _is_cell_doubling_back of Q cells per board
__global__ void prw_cell_query_kernel(const int32_t *__restrict__ grid, int B, int m,
                                      const int32_t *__restrict__ cells, int Q,
                                      const int32_t *__restrict__ wire_id,
                                      int32_t *__restrict__ adjacent,
                                      int32_t *__restrict__ available,
                                      uint8_t *__restrict__ is_free,
                                      uint8_t *__restrict__ not_doubling_back) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)B * Q) return;
  const int cell = cells[idx];
  int out[4];
  if (adjacent) {
    ref_adjacent_cells(cell, m, out);
#pragma unroll
    for (int k = 0; k < 4; ++k) adjacent[idx * 4 + k] = out[k];
  }
  if (!grid) return;
  const HbmGrid g = {grid + (idx / Q) * (size_t)(m * m), m};
  if (available) {
    ref_available_cells(g, m, cell, out);
#pragma unroll
    for (int k = 0; k < 4; ++k) available[idx * 4 + k] = out[k];
  }
  if (is_free) is_free[idx] = (uint8_t)ref_is_cell_free(g, m, cell);
  if (not_doubling_back) {
    const int wire = wire_id ? wire_id[idx] : floor_div(ref_gather(g, m, cell) - 1, 3);
    not_doubling_back[idx] = (uint8_t)ref_not_doubling_back(g, m, wire, cell);
  }
}

// _step_agent(agent, grid, action) (:376-429) for every agent on its own copy of the grid
__global__ void prw_step_agent_kernel(const int32_t *__restrict__ grid, int B, int m, int N,
                                      const int32_t *__restrict__ target,
                                      const int32_t *__restrict__ position,
                                      const int32_t *__restrict__ action,
                                      int32_t *__restrict__ new_position,
                                      int32_t *__restrict__ new_grids,
                                      uint8_t *__restrict__ moved) {
  const size_t idx = blockIdx.x;  // (board, agent)
  const size_t b = idx / N;
  const int i = (int)(idx - b * N), n = m * m;
  const int32_t *g = grid + b * (size_t)n;
  const int r = position[2 * idx], c = position[2 * idx + 1];
  int act = action[idx];
  act = act < 0 ? 0 : act > 4 ? 4 : act;  // lax.switch clamps its index (move_position)
  const int nr = r + (act == 1 ? -1 : act == 3 ? 1 : 0), nc = c + (act == 2 ? 1 : act == 4 ? -1 : 0);
  const bool connected = target && r == target[2 * idx] && c == target[2 * idx + 1];
  const bool in_bounds = 0 <= nr && nr < m && 0 <= nc && nc < m;
  bool ok = false;
  if (in_bounds) {
    const int v = g[nr * m + nc];
    ok = (v == 0 || v == 3 * i + 3) && !connected && act != 0;
  }
  if (threadIdx.x == 0) {
    new_position[2 * idx] = ok ? nr : r;
    new_position[2 * idx + 1] = ok ? nc : c;
    if (moved) moved[idx] = (uint8_t)ok;
  }
  if (!new_grids) return;
  int32_t *out = new_grids + idx * (size_t)n;
  for (int k = threadIdx.x; k < n; k += blockDim.x) {
    int v = g[k];
    if (ok && k == nr * m + nc) v = 3 * i + 2;     // move_agent: the new cell is the head ...
    else if (ok && k == r * m + c) v = 3 * i + 1;  // ... the old one a path cell
    out[k] = v;
  }
}

#endif  // !RBG_TU_RECORD

// The recording rollout's launch, in a function of its own so that it can live in a translation
// unit of its own (csrc/prw_record.hip): it is the one kernel of the library that the max-ilp
// scheduler of the main unit slows down (9.1 instead of 8.1 us per step), see _native.py.
#if defined(RBG_TU_RECORD) || !defined(RBG_RECORD_TU_SEPARATE)
int launch_rollout_record(const RolloutArgs &a, int lpb, unsigned grid, size_t lds,
                          hipStream_t st) {
  if (lpb == 16) {
    hipLaunchKernelGGL((connector_rollout_kernel<16, true>), dim3(grid), dim3(kBlock), lds, st, a);
  } else if (lpb == 32) {
    hipLaunchKernelGGL((connector_rollout_kernel<32, true>), dim3(grid), dim3(kBlock), lds, st, a);
  } else {
    hipLaunchKernelGGL((connector_rollout_kernel<64, true>), dim3(grid), dim3(kBlock), lds, st, a);
  }
  return (int)hipGetLastError();
}
#endif

#ifndef RBG_TU_RECORD  // (not part of the recording-rollout unit)
// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------
thread_local char g_err[256] = "";

int fail(int code, const char *msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

int check_hip(hipError_t e, const char *what) {
  if (e == hipSuccess) return 0;
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return (int)e;
}

// Candidate threshold of shuffle_prefix: about (count + LPB) / 2 of the n uniform 32-bit sort
// keys are expected below it (so that count <= #candidates <= LPB almost always); 0 disables the
// fast path (count too close to LPB, or the threshold would not fit).
uint32_t prefix_threshold(int n, int count, int lpb) {
  if (count * 5 > lpb * 3) return 0u;
  const int expect = (count + lpb) / 2 - 1;
  if (expect <= count || expect >= n) return 0u;
  return (uint32_t)((((uint64_t)expect) << 32) / (uint64_t)n);
}

int make_geo(int m, int N, Geo *g, int prefix_count = -1, int lpb = kWaveSize) {
  if (m < 2 || m > kMaxM) return fail(RBG_E_SHAPE, "grid size m must be in [2, 40]");
  if (N < 1 || N > kMaxN) return fail(RBG_E_SHAPE, "num_agents N must be in [1, 64]");
  if (N > m * m) return fail(RBG_E_SHAPE, "num_agents N exceeds m*m cells");
  g->m = m;
  g->n = m * m;
  g->S = m + 4;
  g->words = (m + 4) * (m + 4);
  g->magic = ((1u << 20) + (uint32_t)m - 1u) / (uint32_t)m;
  g->magic_s = ((1u << 20) + (uint32_t)g->S - 1u) / (uint32_t)g->S;
  g->step_table = ((uint32_t)(-g->S) & 0xFFu) | ((uint32_t)g->S << 8) | (0xFFu << 16) | (1u << 24);
  g->thr = prefix_threshold(m * m, prefix_count < 0 ? N : prefix_count, lpb);
  return 0;
}

// The m % 4 == 0 fast path moves grids with 16-byte loads/stores.
int check_align(const Geo &g, const void *p0, const void *p1, const void *p2) {
  if ((g.m & 3) != 0) return 0;
  const uintptr_t bits = (uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2;
  if (bits & 15u) return fail(RBG_E_PARAM, "grid buffers must be 16-byte aligned when m % 4 == 0");
  return 0;
}

// dynamic LDS of one 256-thread workgroup: 4 waves x (64 / lpb) boards x (board + scratch)
size_t lds_bytes(const Geo &g, int lpb = kWaveSize) {
  const size_t slab = (size_t)((g.words + 3) & ~3) + (size_t)(4 * lpb);
  return (size_t)kWavesPerBlock * (size_t)(kWaveSize / lpb) * slab * sizeof(uint32_t);
}

// Lanes per board for the generator: the smallest sub-wave that holds the 2N + 2 slots of the
// Threefry pipeline, as long as the workgroup's boards fit 64 KiB of LDS.
int pick_lpb(int m, int N) {
  Geo g;
  g.words = (m + 4) * (m + 4);
  for (int lpb = 16; lpb < kWaveSize; lpb *= 2)
    if (2 * N + 2 <= lpb && lds_bytes(g, lpb) <= 64u * 1024u) return lpb;
  return kWaveSize;
}

StatePtrs to_ptrs(const rbg_state *s) {
  StatePtrs p = {s->grid, s->step_count, s->start, s->target, s->position, s->key};
  return p;
}
TimestepPtrs to_ptrs(const rbg_timestep *t) {
  TimestepPtrs p = {t->step_type,      t->reward,          t->discount,
                    t->obs_grid,       t->action_mask,     t->obs_step_count,
                    t->num_connections, t->ratio_connections, t->total_path_length,
                    t->reward_total,   t->discount_max};
  return p;
}
bool state_complete(const rbg_state *s) {
  return s && s->grid && s->step_count && s->start && s->target && s->position && s->key;
}

#endif  // !RBG_TU_RECORD

}  // namespace

#if defined(RBG_TU_RECORD)
// (RolloutArgs is a type of this unit's anonymous namespace: the same source, the same layout)
extern "C" __attribute__((visibility("hidden"))) int rbg_internal_rollout_record(
    const void *args, int lpb, unsigned grid, size_t lds, void *stream) {
  return launch_rollout_record(*static_cast<const RolloutArgs *>(args), lpb, grid, lds,
                               (hipStream_t)stream);
}
#else
#if defined(RBG_RECORD_TU_SEPARATE)
extern "C" __attribute__((visibility("hidden"))) int rbg_internal_rollout_record(
    const void *args, int lpb, unsigned grid, size_t lds, void *stream);
#endif

extern "C" {

int rbg_abi_version(void) { return RBG_ABI_VERSION; }
const char *rbg_version(void) { return "rbg_prw 0.1.0 (gfx950)"; }
const char *rbg_last_error(void) { return g_err; }

int rbg_split(uint32_t key0, uint32_t key1, int num, uint32_t *out, void *stream) {
  if (!out) return fail(RBG_E_NULL, "rbg_split: out is NULL");
  if (num < 0) return fail(RBG_E_SHAPE, "rbg_split: num < 0");
  if (num == 0) return 0;
  hipLaunchKernelGGL(split_kernel, dim3((num + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     key0, key1, num, out);
  return check_hip(hipGetLastError(), "rbg_split");
}

int rbg_split_batch(const uint32_t *keys, int B, int num, uint32_t *out, void *stream) {
  if (B < 0 || num < 0) return fail(RBG_E_SHAPE, "rbg_split_batch: negative size");
  const size_t total = (size_t)B * (size_t)num;
  if (total == 0) return 0;
  if (!keys || !out) return fail(RBG_E_NULL, "rbg_split_batch: NULL pointer");
  hipLaunchKernelGGL(split_batch_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, keys, B, num, out);
  return check_hip(hipGetLastError(), "rbg_split_batch");
}

int rbg_prw_generate(const uint32_t *keys, int B, int m, int N, int split_first,
                     int32_t *solved_grid, int32_t *train_grid, int32_t *start, int32_t *target,
                     int32_t *position, int32_t *step_count, uint32_t *key_out, int32_t *iters,
                     void *stream) {
  if (B < 0) return fail(RBG_E_SHAPE, "rbg_prw_generate: B < 0");
  GenArgs a;
  if (int rc = make_geo(m, N, &a.g)) return rc;
  const int lpb = pick_lpb(m, N);
  if (int rc = make_geo(m, N, &a.g, N, lpb)) return rc;
  if (B == 0) return 0;  // an empty batch may come with NULL (empty-tensor) pointers
  if (!keys || !start || !target) return fail(RBG_E_NULL, "rbg_prw_generate: NULL pointer");
  if (int rc = check_align(a.g, solved_grid, train_grid, nullptr)) return rc;
  a.keys = keys; a.B = B; a.N = N; a.split_first = split_first != 0;
  a.solved_grid = solved_grid; a.train_grid = train_grid; a.start = start; a.target = target;
  a.position = position; a.step_count = step_count; a.key_out = key_out; a.iters = iters;
  const int boards_per_block = kWavesPerBlock * (kWaveSize / lpb);
  const dim3 grid((B + boards_per_block - 1) / boards_per_block), block(kBlock);
  const size_t lds = lds_bytes(a.g, lpb);
  hipStream_t st = (hipStream_t)stream;
  if (lpb == 16) {
    hipLaunchKernelGGL(prw_generate_kernel<16>, grid, block, lds, st, a);
  } else if (lpb == 32) {
    hipLaunchKernelGGL(prw_generate_kernel<32>, grid, block, lds, st, a);
  } else {
    hipLaunchKernelGGL(prw_generate_kernel<64>, grid, block, lds, st, a);
  }
  return check_hip(hipGetLastError(), "rbg_prw_generate");
}

int rbg_prw_generate_philox(const uint32_t *keys, int B, int m, int N, int32_t *solved_grid,
                            int32_t *train_grid, int32_t *start, int32_t *target,
                            int32_t *position, int32_t *step_count, uint32_t *key_out,
                            int32_t *iters, void *stream) {
  if (B < 0) return fail(RBG_E_SHAPE, "rbg_prw_generate_philox: B < 0");
  GenArgs a;
  if (int rc = make_geo(m, N, &a.g)) return rc;
  // lane i = agent i: N lanes per board.  Four boards per wavefront are the most efficient
  // packing (B = 65536: 173 us against 188 with two), but a batch that then gives fewer than two
  // waves per SIMD is latency-bound on every wave: two boards per wave put two waves on each
  // SIMD, which hide each other's LDS round trips (B = 4096: 30.4 us against 39.0).
  int lpb = (B + 3) / 4 < 2 * 1024 ? 32 : 16;
  while (lpb < N || (lpb < kWaveSize && lds_bytes(a.g, lpb) > 64u * 1024u)) lpb *= 2;
  if (int rc = make_geo(m, N, &a.g, N, lpb)) return rc;
  if (B == 0) return 0;
  if (!keys || !start || !target) return fail(RBG_E_NULL, "rbg_prw_generate_philox: NULL pointer");
  if (int rc = check_align(a.g, solved_grid, train_grid, nullptr)) return rc;
  a.keys = keys; a.B = B; a.N = N; a.split_first = 0;
  a.solved_grid = solved_grid; a.train_grid = train_grid; a.start = start; a.target = target;
  a.position = position; a.step_count = step_count; a.key_out = key_out; a.iters = iters;
  const int boards_per_block = kWavesPerBlock * (kWaveSize / lpb);
  const dim3 grid((B + boards_per_block - 1) / boards_per_block), block(kBlock);
  const size_t lds = lds_bytes(a.g, lpb);
  hipStream_t st = (hipStream_t)stream;
  if (lpb == 16) {
    hipLaunchKernelGGL(prw_philox_generate_kernel<16>, grid, block, lds, st, a);
  } else if (lpb == 32) {
    hipLaunchKernelGGL(prw_philox_generate_kernel<32>, grid, block, lds, st, a);
  } else {
    hipLaunchKernelGGL(prw_philox_generate_kernel<64>, grid, block, lds, st, a);
  }
  return check_hip(hipGetLastError(), "rbg_prw_generate_philox");
}

int rbg_seq_rw_generate(const uint32_t *keys, int B, int m, int N, int split_first,
                        int32_t *solved_grid, int32_t *train_grid, int32_t *start,
                        int32_t *target, int32_t *position, int32_t *step_count,
                        uint32_t *key_out, int32_t *attempts, void *stream) {
  if (B < 0) return fail(RBG_E_SHAPE, "rbg_seq_rw_generate: B < 0");
  if (m < 3) return fail(RBG_E_SHAPE, "rbg_seq_rw_generate: m must be >= 3");
  GenArgs a;
  if (int rc = make_geo(m, N, &a.g)) return rc;
  if (B == 0) return 0;
  if (!keys || !start || !target) return fail(RBG_E_NULL, "rbg_seq_rw_generate: NULL pointer");
  if (int rc = check_align(a.g, solved_grid, train_grid, nullptr)) return rc;
  a.keys = keys; a.B = B; a.N = N; a.split_first = split_first != 0;
  a.solved_grid = solved_grid; a.train_grid = train_grid; a.start = start; a.target = target;
  a.position = position; a.step_count = step_count; a.key_out = key_out; a.iters = attempts;
  // step history for resumed retries (rbg_seqwalk.h) when it leaves room for >= 4 workgroups per CU
  const size_t hist_words = (size_t)N * (size_t)(2 * m - 1) * 4;
  size_t lds = lds_bytes(a.g);
  a.seq_hist = 0;
  if (lds + kWavesPerBlock * hist_words * sizeof(uint32_t) <= (size_t)RBG_SEQ_HISTORY_BYTES) {
    a.seq_hist = (int)hist_words;
    lds += kWavesPerBlock * hist_words * sizeof(uint32_t);
  }
  hipLaunchKernelGGL(seq_rw_generate_kernel, dim3((B + kWavesPerBlock - 1) / kWavesPerBlock),
                     dim3(kBlock), lds, (hipStream_t)stream, a);
  return check_hip(hipGetLastError(), "rbg_seq_rw_generate");
}

int rbg_seed_extension_generate(const uint32_t *keys, int B, int m, int N, int split_first,
                                float randomness, int two_sided, int extension_iterations,
                                int max_extension_steps, int32_t *solved_grid, int32_t *train_grid,
                                int32_t *start, int32_t *target, int32_t *position,
                                int32_t *step_count, uint32_t *key_out, int32_t *sweeps,
                                void *stream) {
  if (B < 0) return fail(RBG_E_SHAPE, "rbg_seed_extension_generate: B < 0");
  SeArgs args;
  GenArgs &a = args.gen;
  if (int rc = make_geo(m, N, &a.g)) return rc;
  if (N > (m / 2) * (m / 2))
    return fail(RBG_E_SHAPE, "rbg_seed_extension_generate: N exceeds the (m/2)^2 seed cells");
  if (extension_iterations < -1 || max_extension_steps < 0)
    return fail(RBG_E_PARAM, "rbg_seed_extension_generate: bad iteration / step count");
  if (extension_iterations == -1 && split_first)
    return fail(RBG_E_PARAM, "rbg_seed_extension_generate: seed-only takes the seed keys");
  if (B == 0) return 0;
  if (!keys || !start || !target)
    return fail(RBG_E_NULL, "rbg_seed_extension_generate: NULL pointer");
  if (int rc = check_align(a.g, solved_grid, train_grid, nullptr)) return rc;
  a.keys = keys; a.B = B; a.N = N; a.split_first = split_first != 0;
  a.solved_grid = solved_grid; a.train_grid = train_grid; a.start = start; a.target = target;
  a.position = position; a.step_count = step_count; a.key_out = key_out; a.iters = sweeps;
  SeParams &p = args.p;
  p.N = N;
  p.iterations = extension_iterations < 0 ? 0 : extension_iterations;
  p.seed_only = extension_iterations < 0;
  p.max_steps = max_extension_steps;
  p.two_sided = two_sided != 0;
  p.randomness = randomness;
  // use_random = randomness > uniform, uniform in [0, 1)
  p.rnd_mode = !(randomness > 0.0f) ? 0 : (randomness >= 1.0f ? 1 : 2);
  p.bfs_last = solved_grid != nullptr;  // only the solved board can see the last re-routing
  p.half = m / 2;
  // lanes per board: lane w owns wire w; fewer, fuller waves as long as the workgroup's LDS fits
  int lpb = 16;
  while (lpb < N) lpb *= 2;
  int wpb = kWavesPerBlock;
  size_t lds = 0;
  for (;;) {
    const size_t per_board =
        (size_t)((a.g.words + 3) & ~3) + (size_t)(4 * lpb) + (size_t)se_extra_words(a.g.words, a.g.n);
    lds = (size_t)wpb * (size_t)(kWaveSize / lpb) * per_board * sizeof(uint32_t);
    if (lds <= 64u * 1024u) break;
    if (lpb < kWaveSize) lpb *= 2; else if (wpb > 1) wpb /= 2; else break;
  }
  if (lds > 64u * 1024u) return fail(RBG_E_SHAPE, "rbg_seed_extension_generate: board too large");
  const int boards_per_block = wpb * (kWaveSize / lpb);
  const dim3 grid((B + boards_per_block - 1) / boards_per_block), block(kWaveSize * wpb);
  hipStream_t st = (hipStream_t)stream;
  if (lpb == 16) {
    hipLaunchKernelGGL(seed_extension_kernel<16>, grid, block, lds, st, args);
  } else if (lpb == 32) {
    hipLaunchKernelGGL(seed_extension_kernel<32>, grid, block, lds, st, args);
  } else {
    hipLaunchKernelGGL(seed_extension_kernel<64>, grid, block, lds, st, args);
  }
  return check_hip(hipGetLastError(), "rbg_seed_extension_generate");
}

int rbg_uniform_generate(const uint32_t *keys, int B, int m, int N, int32_t *train_grid,
                         int32_t *start, int32_t *target, int32_t *position,
                         int32_t *step_count, uint32_t *key_out, void *stream) {
  if (B < 0) return fail(RBG_E_SHAPE, "rbg_uniform_generate: B < 0");
  UniformArgs a;
  if (int rc = make_geo(m, N, &a.g, 2 * N)) return rc;
  if (2 * N > m * m) return fail(RBG_E_SHAPE, "rbg_uniform_generate: 2N exceeds m*m cells");
  if (B == 0) return 0;
  if (!keys || !start || !target) return fail(RBG_E_NULL, "rbg_uniform_generate: NULL pointer");
  if (int rc = check_align(a.g, train_grid, nullptr, nullptr)) return rc;
  a.keys = keys; a.B = B; a.N = N;
  a.train_grid = train_grid; a.start = start; a.target = target; a.position = position;
  a.step_count = step_count; a.key_out = key_out;
  hipLaunchKernelGGL(uniform_generate_kernel, dim3((B + kWavesPerBlock - 1) / kWavesPerBlock),
                     dim3(kBlock), lds_bytes(a.g), (hipStream_t)stream, a);
  return check_hip(hipGetLastError(), "rbg_uniform_generate");
}

static int launch_env(EnvArgs &a, void *stream, const char *what) {
  const dim3 grid((a.B + kEnvWavesPerBlock - 1) / kEnvWavesPerBlock),
      block(kWaveSize * kEnvWavesPerBlock);
  const size_t lds = lds_bytes(a.g) / kWavesPerBlock * kEnvWavesPerBlock;
  hipStream_t st = (hipStream_t)stream;
  if (a.mode == kModeReset) {
    hipLaunchKernelGGL(connector_reset_kernel, grid, block, lds, st, a);
  } else if (a.mode == kModeObserve) {
    hipLaunchKernelGGL(connector_observe_kernel, grid, block, lds, st, a);
  } else {
    hipLaunchKernelGGL(connector_step_kernel, grid, block, lds, st, a);
  }
  return check_hip(hipGetLastError(), what);
}

int rbg_connector_reset(const uint32_t *keys, int B, int m, int N, const rbg_state *state,
                        const rbg_timestep *timestep, void *stream) {
  if (B < 0) return fail(RBG_E_SHAPE, "rbg_connector_reset: B < 0");
  EnvArgs a;
  memset(&a, 0, sizeof(a));
  if (int rc = make_geo(m, N, &a.g)) return rc;
  if (B == 0) return 0;
  if (!keys || !timestep) return fail(RBG_E_NULL, "rbg_connector_reset: NULL pointer");
  if (!state_complete(state)) return fail(RBG_E_NULL, "rbg_connector_reset: incomplete state");
  if (int rc = check_align(a.g, state->grid, timestep->obs_grid, nullptr)) return rc;
  a.keys = keys; a.B = B; a.N = N; a.out = to_ptrs(state); a.in = a.out;
  a.ts = to_ptrs(timestep); a.mode = 0;
  return launch_env(a, stream, "rbg_connector_reset");
}

static int fill_stash(const rbg_stash *stash, StashPtrs *out, const char *who) {
  if (!stash) return 0;
  if (!state_complete(&stash->next) || !stash->tag || !stash->ready_at || !stash->refill_key) {
    snprintf(g_err, sizeof(g_err), "%s: incomplete stash", who);
    return RBG_E_NULL;
  }
  out->next = to_ptrs(&stash->next);
  out->tag = stash->tag;
  out->ready_at = stash->ready_at;
  out->refill_key = stash->refill_key;
  out->hits = stash->hits;
  out->depth = stash->depth;
  if (stash->depth != 1 && stash->depth != 2) {
    snprintf(g_err, sizeof(g_err), "%s: stash depth must be 1 or 2", who);
    return RBG_E_PARAM;
  }
  return 0;
}

int rbg_connector_reset_prefetch(const uint32_t *keys, int B, int m, int N, const rbg_state *state,
                                 const rbg_timestep *timestep, const rbg_stash *stash, int tick,
                                 int ready_delay, void *stream) {
  if (B < 0) return fail(RBG_E_SHAPE, "rbg_connector_reset_prefetch: B < 0");
  EnvArgs a;
  memset(&a, 0, sizeof(a));
  if (int rc = make_geo(m, N, &a.g)) return rc;
  if (B == 0) return 0;
  if (!keys || !timestep) return fail(RBG_E_NULL, "rbg_connector_reset_prefetch: NULL pointer");
  if (!state_complete(state)) return fail(RBG_E_NULL, "rbg_connector_reset_prefetch: incomplete state");
  if (int rc = check_align(a.g, state->grid, timestep->obs_grid, stash ? stash->next.grid : nullptr))
    return rc;
  if (int rc = fill_stash(stash, &a.stash, "rbg_connector_reset_prefetch")) return rc;
  a.keys = keys; a.B = B; a.N = N; a.out = to_ptrs(state); a.in = a.out;
  a.ts = to_ptrs(timestep); a.mode = 0; a.tick = tick; a.ready_delay = ready_delay;
  return launch_env(a, stream, "rbg_connector_reset_prefetch");
}

int rbg_connector_step_prefetch(int B, int m, int N, const rbg_state *state_in,
                                const rbg_state *state_out, const int32_t *action,
                                const rbg_env_params *params, const rbg_timestep *timestep,
                                const rbg_stash *stash, int tick, int ready_delay, void *stream) {
  if (B < 0) return fail(RBG_E_SHAPE, "rbg_connector_step_prefetch: B < 0");
  if (!params) return fail(RBG_E_NULL, "rbg_connector_step_prefetch: NULL params");
  if (params->time_limit < 1) return fail(RBG_E_PARAM, "rbg_connector_step_prefetch: time_limit < 1");
  EnvArgs a;
  memset(&a, 0, sizeof(a));
  if (int rc = make_geo(m, N, &a.g)) return rc;
  if (B == 0) return 0;
  if (!action || !timestep) return fail(RBG_E_NULL, "rbg_connector_step_prefetch: NULL pointer");
  if (!state_complete(state_in) || !state_complete(state_out))
    return fail(RBG_E_NULL, "rbg_connector_step_prefetch: incomplete state");
  if (int rc = check_align(a.g, state_in->grid, state_out->grid, timestep->obs_grid)) return rc;
  if (stash)
    if (int rc = check_align(a.g, stash->next.grid, nullptr, nullptr)) return rc;
  if (int rc = fill_stash(stash, &a.stash, "rbg_connector_step_prefetch")) return rc;
  a.B = B; a.N = N; a.in = to_ptrs(state_in); a.out = to_ptrs(state_out); a.action = action;
  a.p = *params; a.ts = to_ptrs(timestep); a.mode = 2; a.tick = tick; a.ready_delay = ready_delay;
  return launch_env(a, stream, "rbg_connector_step_prefetch");
}

int rbg_stash_refill(int B, int m, int N, const rbg_stash *stash, void *stream) {
  if (B < 0) return fail(RBG_E_SHAPE, "rbg_stash_refill: B < 0");
  RefillArgs a;
  memset(&a, 0, sizeof(a));
  const int lpb = (m >= 2 && m <= kMaxM && N >= 1 && N <= kMaxN) ? pick_lpb(m, N) : kWaveSize;
  if (int rc = make_geo(m, N, &a.g, -1, lpb)) return rc;
  if (B == 0) return 0;
  if (!stash) return fail(RBG_E_NULL, "rbg_stash_refill: NULL stash");
  if (int rc = check_align(a.g, stash->next.grid, nullptr, nullptr)) return rc;
  if (int rc = fill_stash(stash, &a.stash, "rbg_stash_refill")) return rc;
  a.B = B * stash->depth;  // the kernel works on rows
  a.N = N;
  const int envs_per_block = kWavesPerBlock * (kWaveSize / lpb);
  const dim3 grid((a.B + envs_per_block - 1) / envs_per_block), block(kBlock);
  const size_t lds = lds_bytes(a.g, lpb);
  hipStream_t st = (hipStream_t)stream;
  if (lpb == 16) {
    hipLaunchKernelGGL(stash_refill_kernel<16>, grid, block, lds, st, a);
  } else if (lpb == 32) {
    hipLaunchKernelGGL(stash_refill_kernel<32>, grid, block, lds, st, a);
  } else {
    hipLaunchKernelGGL(stash_refill_kernel<64>, grid, block, lds, st, a);
  }
  return check_hip(hipGetLastError(), "rbg_stash_refill");
}

int rbg_connector_observe(int B, int m, int N, const rbg_state *state,
                          const rbg_timestep *timestep, void *stream) {
  if (B < 0) return fail(RBG_E_SHAPE, "rbg_connector_observe: B < 0");
  EnvArgs a;
  memset(&a, 0, sizeof(a));
  if (int rc = make_geo(m, N, &a.g)) return rc;
  if (B == 0) return 0;
  if (!timestep) return fail(RBG_E_NULL, "rbg_connector_observe: NULL timestep");
  if (!state_complete(state)) return fail(RBG_E_NULL, "rbg_connector_observe: incomplete state");
  if (int rc = check_align(a.g, state->grid, timestep->obs_grid, nullptr)) return rc;
  a.B = B; a.N = N; a.in = to_ptrs(state); a.out = a.in; a.ts = to_ptrs(timestep); a.mode = 1;
  return launch_env(a, stream, "rbg_connector_observe");
}

int rbg_connector_step(int B, int m, int N, const rbg_state *state_in, const rbg_state *state_out,
                       const int32_t *action, const rbg_env_params *params,
                       const rbg_timestep *timestep, void *stream) {
  if (B < 0) return fail(RBG_E_SHAPE, "rbg_connector_step: B < 0");
  if (!params) return fail(RBG_E_NULL, "rbg_connector_step: NULL params");
  if (params->time_limit < 1) return fail(RBG_E_PARAM, "rbg_connector_step: time_limit < 1");
  EnvArgs a;
  memset(&a, 0, sizeof(a));
  if (int rc = make_geo(m, N, &a.g)) return rc;
  if (B == 0) return 0;
  if (!action || !timestep) return fail(RBG_E_NULL, "rbg_connector_step: NULL pointer");
  if (!state_complete(state_in) || !state_complete(state_out))
    return fail(RBG_E_NULL, "rbg_connector_step: incomplete state");
  if (int rc = check_align(a.g, state_in->grid, state_out->grid, timestep->obs_grid)) return rc;
  a.B = B; a.N = N; a.in = to_ptrs(state_in); a.out = to_ptrs(state_out); a.action = action;
  a.p = *params; a.ts = to_ptrs(timestep); a.mode = 2;
  return launch_env(a, stream, "rbg_connector_step");
}

int rbg_connector_rollout(int B, int m, int N, const rbg_state *state_in,
                          const rbg_state *state_out, const uint32_t *policy_keys, uint32_t t0,
                          int n_steps, const rbg_env_params *params,
                          const rbg_rollout_stats *stats, const rbg_timestep *last_timestep,
                          void *stream) {
  return rbg_connector_rollout_prefetch(B, m, N, state_in, state_out, policy_keys, t0, n_steps,
                                        params, stats, last_timestep, nullptr, 0, 0, stream);
}

int rbg_connector_rollout_prefetch(int B, int m, int N, const rbg_state *state_in,
                                   const rbg_state *state_out, const uint32_t *policy_keys,
                                   uint32_t t0, int n_steps, const rbg_env_params *params,
                                   const rbg_rollout_stats *stats,
                                   const rbg_timestep *last_timestep, const rbg_stash *stash,
                                   int tick, int ready_delay, void *stream) {
  return rbg_connector_rollout_record(B, m, N, state_in, state_out, policy_keys, t0, n_steps,
                                      params, stats, last_timestep, nullptr, nullptr, stash, tick,
                                      ready_delay, stream);
}

int rbg_connector_rollout_record(int B, int m, int N, const rbg_state *state_in,
                                 const rbg_state *state_out, const uint32_t *policy_keys,
                                 uint32_t t0, int n_steps, const rbg_env_params *params,
                                 const rbg_rollout_stats *stats,
                                 const rbg_timestep *last_timestep,
                                 const rbg_timestep *trajectory, int32_t *trajectory_action,
                                 const rbg_stash *stash, int tick, int ready_delay,
                                 void *stream) {
  if (B < 0) return fail(RBG_E_SHAPE, "rbg_connector_rollout: B < 0");
  if (n_steps < 1) return fail(RBG_E_PARAM, "rbg_connector_rollout: n_steps < 1");
  if (!params) return fail(RBG_E_NULL, "rbg_connector_rollout: NULL params");
  if (params->time_limit < 1) return fail(RBG_E_PARAM, "rbg_connector_rollout: time_limit < 1");
  RolloutArgs a;
  memset(&a, 0, sizeof(a));
  const int lpb = (m >= 2 && m <= kMaxM && N >= 1 && N <= kMaxN) ? pick_lpb(m, N) : kWaveSize;
  if (int rc = make_geo(m, N, &a.g, -1, lpb)) return rc;
  if (B == 0) return 0;
  if (!policy_keys) return fail(RBG_E_NULL, "rbg_connector_rollout: NULL policy_keys");
  if (!state_complete(state_in) || !state_complete(state_out))
    return fail(RBG_E_NULL, "rbg_connector_rollout: incomplete state");
  if (int rc = check_align(a.g, state_in->grid, state_out->grid,
                           last_timestep ? last_timestep->obs_grid : nullptr))
    return rc;
  a.B = B; a.N = N; a.in = to_ptrs(state_in); a.out = to_ptrs(state_out);
  a.policy_keys = policy_keys; a.t0 = t0; a.n_steps = n_steps; a.p = *params;
  if (stats) {
    RolloutStatsPtrs sp = {stats->episodes, stats->connections_made,
                           stats->unconnected_agent_steps, stats->final_connections,
                           stats->final_path_length};
    a.stats = sp;
  }
  if (last_timestep) a.ts = to_ptrs(last_timestep);
  const bool record = trajectory != nullptr || trajectory_action != nullptr;
  if (record) {
    if ((long long)n_steps * (long long)B > 0x7FFFFFFFll)
      return fail(RBG_E_SHAPE, "rbg_connector_rollout_record: n_steps * B too large");
    if (trajectory) {
      if (int rc = check_align(a.g, trajectory->obs_grid, nullptr, nullptr)) return rc;
      a.traj = to_ptrs(trajectory);
    }
    a.traj_action = trajectory_action;
  }
  if (stash)
    if (int rc = check_align(a.g, stash->next.grid, nullptr, nullptr)) return rc;
  if (int rc = fill_stash(stash, &a.stash, "rbg_connector_rollout_prefetch")) return rc;
  a.tick = tick; a.ready_delay = ready_delay;
  const bool wants_stats = stats && (stats->episodes || stats->connections_made ||
                                     stats->unconnected_agent_steps || stats->final_connections ||
                                     stats->final_path_length);
  if (n_steps == 1 && last_timestep && last_timestep->obs_grid && !wants_stats && !record) {
    // "random policy + step" with the observation written: the HBM-bound shape of the per-step
    // kernel (one env per wave, 16 waves per CU in flight), not the LDS-resident rollout's
    EnvArgs e;
    memset(&e, 0, sizeof(e));
    if (int rc = make_geo(m, N, &e.g)) return rc;  // (the shuffle threshold depends on the lane count)
    e.B = B; e.N = N; e.in = a.in; e.out = a.out; e.p = a.p; e.ts = a.ts; e.mode = kModeStep;
    e.stash = a.stash; e.tick = tick; e.ready_delay = ready_delay;
    e.policy_keys = policy_keys; e.policy_t = t0;
    return launch_env(e, stream, "rbg_connector_rollout");
  }
  const int envs_per_block = kWavesPerBlock * (kWaveSize / lpb);
  const dim3 grid((B + envs_per_block - 1) / envs_per_block), block(kBlock);
  const size_t lds = lds_bytes(a.g, lpb);
  hipStream_t st = (hipStream_t)stream;
  if (record) {
#if defined(RBG_RECORD_TU_SEPARATE)
    return check_hip((hipError_t)rbg_internal_rollout_record(&a, lpb, grid.x, lds, stream),
                     "rbg_connector_rollout_record");
#else
    return check_hip((hipError_t)launch_rollout_record(a, lpb, grid.x, lds, st),
                     "rbg_connector_rollout_record");
#endif
  } else if (lpb == 16) {
    hipLaunchKernelGGL((connector_rollout_kernel<16, false>), grid, block, lds, st, a);
  } else if (lpb == 32) {
    hipLaunchKernelGGL((connector_rollout_kernel<32, false>), grid, block, lds, st, a);
  } else {
    hipLaunchKernelGGL((connector_rollout_kernel<64, false>), grid, block, lds, st, a);
  }
  return check_hip(hipGetLastError(), "rbg_connector_rollout");
}

int rbg_dataset_sample(const uint32_t *keys, int B, int m, int N, int n_boards,
                       const int32_t *heads, const int32_t *targets, const rbg_state *state,
                       int32_t *which, void *stream) {
  if (B < 0) return fail(RBG_E_SHAPE, "rbg_dataset_sample: B < 0");
  if (n_boards < 1) return fail(RBG_E_SHAPE, "rbg_dataset_sample: n_boards < 1");
  DatasetArgs a;
  memset(&a, 0, sizeof(a));
  if (int rc = make_geo(m, N, &a.g)) return rc;
  if (B == 0) return 0;
  if (!keys || !heads || !targets) return fail(RBG_E_NULL, "rbg_dataset_sample: NULL pointer");
  if (!state_complete(state)) return fail(RBG_E_NULL, "rbg_dataset_sample: incomplete state");
  if (int rc = check_align(a.g, state->grid, nullptr, nullptr)) return rc;
  const uint32_t span = (uint32_t)n_boards;
  const uint32_t m16 = 65536u % span;
  a.mult = (uint32_t)(m16 * m16) % span;  // wrapping uint32 arithmetic, as in jax's _randint
  a.keys = keys; a.B = B; a.N = N; a.n_boards = n_boards; a.heads = heads; a.targets = targets;
  a.out = to_ptrs(state); a.which = which;
  hipLaunchKernelGGL(dataset_sample_kernel, dim3((B + kWavesPerBlock - 1) / kWavesPerBlock),
                     dim3(kBlock), lds_bytes(a.g), (hipStream_t)stream, a);
  return check_hip(hipGetLastError(), "rbg_dataset_sample");
}

int rbg_random_actions(const uint32_t *keys, int B, int N, uint32_t t, const uint8_t *action_mask,
                       int32_t *action, void *stream) {
  if (B < 0 || N < 1 || N > kMaxN) return fail(RBG_E_SHAPE, "rbg_random_actions: bad B or N");
  const size_t total = (size_t)B * (size_t)N;
  if (total == 0) return 0;
  if (!keys || !action_mask || !action) return fail(RBG_E_NULL, "rbg_random_actions: NULL pointer");
  hipLaunchKernelGGL(random_actions_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, keys, B, N, t, action_mask, action);
  return check_hip(hipGetLastError(), "rbg_random_actions");
}

int rbg_prw_initialise_agents(const uint32_t *keys, int B, int m, int N, const int32_t *grid_in,
                              int32_t *grid_out, int32_t *start, void *stream) {
  if (B < 0) return fail(RBG_E_SHAPE, "rbg_prw_initialise_agents: B < 0");
  PiecesArgs a = {};
  if (int rc = make_geo(m, N, &a.g)) return rc;
  if (B == 0) return 0;
  if (!keys || !grid_out || !start) return fail(RBG_E_NULL, "rbg_prw_initialise_agents: NULL pointer");
  if (int rc = check_align(a.g, grid_in, grid_out, nullptr)) return rc;
  a.keys = keys; a.B = B; a.N = N; a.grid_in = grid_in; a.grid_out = grid_out; a.position_out = start;
  hipLaunchKernelGGL(prw_initialise_agents_kernel, dim3((B + kWavesPerBlock - 1) / kWavesPerBlock),
                     dim3(kBlock), lds_bytes(a.g), (hipStream_t)stream, a);
  return check_hip(hipGetLastError(), "rbg_prw_initialise_agents");
}

int rbg_prw_step(const uint32_t *keys, int B, int m, int N, const int32_t *grid,
                 const int32_t *target, const int32_t *position, uint32_t *key_out,
                 int32_t *grid_out, int32_t *position_out, int32_t *action_out, void *stream) {
  if (B < 0) return fail(RBG_E_SHAPE, "rbg_prw_step: B < 0");
  PiecesArgs a = {};
  if (int rc = make_geo(m, N, &a.g)) return rc;
  if (B == 0) return 0;
  if (!keys || !grid || !position || !grid_out || !position_out)
    return fail(RBG_E_NULL, "rbg_prw_step: NULL pointer");
  if (int rc = check_align(a.g, grid, grid_out, nullptr)) return rc;
  a.keys = keys; a.B = B; a.N = N; a.grid_in = grid; a.target = target; a.position = position;
  a.key_out = key_out; a.grid_out = grid_out; a.position_out = position_out; a.action_out = action_out;
  hipLaunchKernelGGL(prw_step_kernel, dim3((B + kWavesPerBlock - 1) / kWavesPerBlock),
                     dim3(kBlock), lds_bytes(a.g), (hipStream_t)stream, a);
  return check_hip(hipGetLastError(), "rbg_prw_step");
}

int rbg_prw_select_action(const uint32_t *agent_keys, int B, int m, int N, const int32_t *grid,
                          const int32_t *position, int32_t *action, void *stream) {
  if (B < 0) return fail(RBG_E_SHAPE, "rbg_prw_select_action: B < 0");
  Geo g;
  if (int rc = make_geo(m, N, &g)) return rc;
  const size_t total = (size_t)B * (size_t)N;
  if (total == 0) return 0;
  if (!agent_keys || !grid || !position || !action)
    return fail(RBG_E_NULL, "rbg_prw_select_action: NULL pointer");
  hipLaunchKernelGGL(prw_select_action_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, agent_keys, B, m, N, grid, position, action);
  return check_hip(hipGetLastError(), "rbg_prw_select_action");
}

int rbg_prw_cell_queries(const int32_t *grid, int B, int m, const int32_t *cells, int Q,
                         const int32_t *wire_id, int32_t *adjacent, int32_t *available,
                         uint8_t *is_free, uint8_t *not_doubling_back, void *stream) {
  if (B < 0 || Q < 0) return fail(RBG_E_SHAPE, "rbg_prw_cell_queries: negative size");
  if (m < 2 || m > kMaxM) return fail(RBG_E_SHAPE, "grid size m must be in [2, 40]");
  const size_t total = (size_t)B * (size_t)Q;
  if (total == 0) return 0;
  if (!cells) return fail(RBG_E_NULL, "rbg_prw_cell_queries: cells is NULL");
  if (!grid && (available || is_free || not_doubling_back))
    return fail(RBG_E_NULL, "rbg_prw_cell_queries: these outputs need a grid");
  hipLaunchKernelGGL(prw_cell_query_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, grid, B, m, cells, Q, wire_id, adjacent, available,
                     is_free, not_doubling_back);
  return check_hip(hipGetLastError(), "rbg_prw_cell_queries");
}

int rbg_prw_step_agent(const int32_t *grid, int B, int m, int N, const int32_t *target,
                       const int32_t *position, const int32_t *action, int32_t *new_position,
                       int32_t *new_grids, uint8_t *moved, void *stream) {
  if (B < 0) return fail(RBG_E_SHAPE, "rbg_prw_step_agent: B < 0");
  Geo g;
  if (int rc = make_geo(m, N, &g)) return rc;
  const size_t total = (size_t)B * (size_t)N;
  if (total == 0) return 0;
  if (total > 0x7FFFFFFFull) return fail(RBG_E_SHAPE, "rbg_prw_step_agent: B * N too large");
  if (!grid || !position || !action || !new_position)
    return fail(RBG_E_NULL, "rbg_prw_step_agent: NULL pointer");
  hipLaunchKernelGGL(prw_step_agent_kernel, dim3((unsigned)total), dim3(256), 0,
                     (hipStream_t)stream, grid, B, m, N, target, position, action, new_position,
                     new_grids, moved);
  return check_hip(hipGetLastError(), "rbg_prw_step_agent");
}

}  // extern "C"
#endif  // !RBG_TU_RECORD
