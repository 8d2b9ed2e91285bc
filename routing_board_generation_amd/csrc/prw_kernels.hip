// prw_kernels.hip -- hand-written gfx950 (CDNA4) kernels + the C-ABI of include/rbg_prw.h.
//
// One 64-lane wavefront owns one board / one env for the whole launch:
//   * the board lives in LDS as a zero-copy "padded" image (2 wall rows above/below, 4 wall
//     columns shared between consecutive rows) so that the 4-neighbour and neighbour-of-neighbour
//     stencils need no bounds checks;
//   * lane i < N is agent i (availability stencil, weighted choice, move);
//   * the reference's "N private grid copies -> max-join -> correction mask" collision rule
//     (JAXI/parallel_random_walk.py:119-145) collapses to ONE `ds_max_u32` per mover on the
//     contested cell: the highest head value (= highest agent id) wins, losers never left;
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

namespace {

constexpr int kWaveSize = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kBlock = kWaveSize * kWavesPerBlock;
constexpr uint32_t kWall = 0xFFFFFFFFu;
constexpr int kMaxM = 40;
constexpr int kMaxN = 64;

// ---------------------------------------------------------------------------------------------
// Threefry-2x32-20 ([3P] jax/_src/prng.py::_threefry2x32_lowering)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rotl(uint32_t x, int r) {
  return __builtin_rotateleft32(x, r);  // v_alignbit_b32
}

#define RBG_TF_ROUND(r)   \
  x0 += x1;               \
  x1 = rotl(x1, r) ^ x0;

// The 20 rounds are split in two halves so that the walk loop can place each half in the shadow
// of a different LDS round trip (see prw_walk_impl).
struct TfState {
  uint32_t x0, x1, k0, k1, k2;
};

__device__ __forceinline__ TfState threefry_first_half(uint32_t k0, uint32_t k1, uint32_t x0,
                                                       uint32_t x1) {
  const uint32_t k2 = k0 ^ k1 ^ 0x1BD11BDAu;
  x0 += k0;
  x1 += k1;
  RBG_TF_ROUND(13) RBG_TF_ROUND(15) RBG_TF_ROUND(26) RBG_TF_ROUND(6)
  x0 += k1;
  x1 += k2 + 1u;
  RBG_TF_ROUND(17) RBG_TF_ROUND(29) RBG_TF_ROUND(16) RBG_TF_ROUND(24)
  x0 += k2;
  x1 += k0 + 2u;
  TfState t = {x0, x1, k0, k1, k2};
  return t;
}

__device__ __forceinline__ void threefry_second_half(const TfState &t, uint32_t &y0,
                                                     uint32_t &y1) {
  uint32_t x0 = t.x0, x1 = t.x1;
  const uint32_t k0 = t.k0, k1 = t.k1, k2 = t.k2;
  RBG_TF_ROUND(13) RBG_TF_ROUND(15) RBG_TF_ROUND(26) RBG_TF_ROUND(6)
  x0 += k0;
  x1 += k1 + 3u;
  RBG_TF_ROUND(17) RBG_TF_ROUND(29) RBG_TF_ROUND(16) RBG_TF_ROUND(24)
  x0 += k1;
  x1 += k2 + 4u;
  RBG_TF_ROUND(13) RBG_TF_ROUND(15) RBG_TF_ROUND(26) RBG_TF_ROUND(6)
  x0 += k2;
  x1 += k0 + 5u;
  y0 = x0;
  y1 = x1;
}

__device__ __forceinline__ void threefry(uint32_t k0, uint32_t k1, uint32_t x0, uint32_t x1,
                                         uint32_t &y0, uint32_t &y1) {
  const TfState t = threefry_first_half(k0, k1, x0, x1);
  threefry_second_half(t, y0, y1);
}

// Out-of-line copy for the once-per-board code (key derivation, sort keys, pipeline prologue):
// that code is straight-line and executed once, so its cost is instruction FETCH, not issue;
// sharing one Threefry body keeps it ~4 KB smaller.  The walk loop keeps its inlined halves.
__device__ __noinline__ uint2 threefry_call(uint32_t k0, uint32_t k1, uint32_t x0, uint32_t x1) {
  uint32_t y0, y1;
  threefry(k0, k1, x0, x1, y0, y1);
  return make_uint2(y0, y1);
}

struct Key2 {
  uint32_t a, b;
};

// Word w of random_bits(key, total) with jax's layout: counters iota(total) (+ one 0 pad word if
// total is odd) split in halves; block j hashes (j, half + j); out = y0-half ++ y1-half.
__device__ __forceinline__ uint32_t random_bits_word(Key2 k, uint32_t w, uint32_t total) {
  const uint32_t half = (total + 1u) >> 1;
  const bool hi = w >= half;
  const uint32_t j = hi ? w - half : w;
  const uint32_t c1 = (half + j < total) ? half + j : 0u;
  uint32_t y0, y1;
  threefry(k.a, k.b, j, c1, y0, y1);
  return hi ? y1 : y0;
}

// split(key)[which]: children (o[0],o[1]) and (o[2],o[3]) of random_bits(key, 4):
// blocks (0,2) and (1,3); o = [y0(0), y0(1), y1(0), y1(1)].
__device__ __forceinline__ Key2 split2(Key2 k, int which) {
  uint32_t a0, a1, b0, b1;
  threefry(k.a, k.b, 0u, 2u, a0, a1);
  threefry(k.a, k.b, 1u, 3u, b0, b1);
  Key2 r;
  r.a = which ? a1 : a0;
  r.b = which ? b1 : b0;
  return r;
}

// Child i of split(key, num): words (2i, 2i+1) of random_bits(key, 2*num).
__device__ __forceinline__ Key2 split_child(Key2 k, uint32_t i, uint32_t num) {
  Key2 r;
  r.a = random_bits_word(k, 2u * i, 2u * num);
  r.b = random_bits_word(k, 2u * i + 1u, 2u * num);
  return r;
}

// [3P] jax/_src/random.py::_uniform (float32): mantissa fill, subtract 1.
__device__ __forceinline__ float uniform01(Key2 k) {
  uint32_t y0, y1;
  threefry(k.a, k.b, 0u, 0u, y0, y1);  // random_bits(key, 1): counters (0, pad 0)
  return __fsub_rn(__uint_as_float((y0 >> 9) | 0x3F800000u), 1.0f);
}

// ---------------------------------------------------------------------------------------------
// Wave helpers
// ---------------------------------------------------------------------------------------------
// LDS traffic of ONE wave is processed in issue order, so a later ds_read observes an earlier
// ds_write of another lane of the same wave; this only has to stop the compiler from reordering.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const uint32_t lo = __shfl_xor((uint32_t)v, off);
    const uint32_t hi = __shfl_xor((uint32_t)(v >> 32), off);
    const uint64_t o = ((uint64_t)hi << 32) | lo;
    v = o < v ? o : v;
  }
  return v;
}

__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ---------------------------------------------------------------------------------------------
// Padded board geometry
// ---------------------------------------------------------------------------------------------
struct Geo {
  int m;        // grid side
  int n;        // m * m
  int S;        // padded row stride = m + 4 (4 wall columns on the left of every row; they also
                // serve as the right wall of the previous row)
  int words;    // (m + 4) * S: 2 wall rows above and below
  uint32_t magic;  // ceil(2^20 / m): idx / m == (idx * magic) >> 20 for idx * m < 2^20
  uint32_t thr;    // shuffle_prefix candidate threshold for this launch's count (0 = slow path)
};

__device__ __forceinline__ int pad_index(const Geo &g, int r, int c) {
  return (r + 2) * g.S + 4 + c;
}
__device__ __forceinline__ int div_m(const Geo &g, int idx) {
  return (int)(((uint32_t)idx * g.magic) >> 20);
}

// walls everywhere, zeros inside
__device__ __forceinline__ void board_clear(uint32_t *board, const Geo &g, int lane) {
  for (int w = lane; w < g.words; w += kWaveSize) board[w] = kWall;
  wave_lds_fence();
  for (int idx = lane; idx < g.n; idx += kWaveSize) {
    const int r = div_m(g, idx);
    board[pad_index(g, r, idx - r * g.m)] = 0u;
  }
  wave_lds_fence();
}

// ---------------------------------------------------------------------------------------------
// First `count` entries of jax's one-round _shuffle of arange(n):
//   key, sub = split(key); sort_keys = random_bits(sub, n); stable sort -> ascending
//   (sort_key, index).
// No full sort: the first `count` entries are the `count` smallest (sort_key, index) pairs.
//   fast path  a threshold `thr` (host-computed so that ~3*count keys fall below it) filters the
//              candidates, a ballot/popcount prefix compacts them one per lane, and every lane
//              ranks its candidate against the others through v_readlane -- exact whenever
//              count <= #candidates <= 64;
//   slow path  otherwise (rare, or count > 32): `count` wave-wide min-reductions.
// Lane l receives entry l in sel0 and entry 64 + l in sel1.  `scratch` needs n words, `aux` 64.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void shuffle_prefix(uint32_t *scratch, uint32_t *aux, const Geo &g,
                                               Key2 sub, int count, uint32_t thr, int lane,
                                               int &sel0, int &sel1) {
  const uint32_t n = (uint32_t)g.n;
  const uint32_t half = (n + 1u) >> 1;
  for (uint32_t j = lane; j < half; j += kWaveSize) {
    const bool has_hi = half + j < n;
    const uint2 y = threefry_call(sub.a, sub.b, j, has_hi ? half + j : 0u);
    scratch[j] = y.x;
    if (has_hi) scratch[half + j] = y.y;
  }
  wave_lds_fence();
  sel0 = sel1 = -1;

  if (thr != 0u) {
    // ---- compact the candidates (key < thr) into lanes, in index order ----------------------
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint32_t my_key = 0xFFFFFFFFu, my_idx = 0xFFFFFFFFu;
    int total = 0;
    for (uint32_t base = 0; base < n; base += kWaveSize) {
      const uint32_t j = base + lane;
      const uint32_t key = j < n ? scratch[j] : 0xFFFFFFFFu;
      const bool flag = j < n && key < thr;
      const uint64_t mask = __ballot(flag);
      const int pos = total + __popcll(mask & lt_mask);
      if (flag && pos < kWaveSize) {
        aux[pos] = j;  // the key is re-read below: one word per candidate keeps aux at 64 words
      }
      total += __popcll(mask);
    }
    wave_lds_fence();
    if (total >= count && total <= kWaveSize) {
      if (lane < total) {
        my_idx = aux[lane];
        my_key = scratch[my_idx];
      }
      const uint64_t mine = ((uint64_t)my_key << 32) | my_idx;
      int rank = 0;
      for (int c = 0; c < total; ++c) {
        const uint32_t ck = __builtin_amdgcn_readlane(my_key, c);
        const uint32_t ci = __builtin_amdgcn_readlane(my_idx, c);
        rank += (int)((((uint64_t)ck << 32) | ci) < mine);
      }
      wave_lds_fence();
      if (lane < total) aux[rank] = my_idx;  // ranks of distinct pairs are a permutation
      wave_lds_fence();
      if (lane < count) sel0 = (int)aux[lane];
      wave_lds_fence();
      return;
    }
  }

  // ---- slow path: one wave-wide 64-bit min per entry ------------------------------------------
  uint64_t prev = 0;
  bool have_prev = false;
  for (int round = 0; round < count; ++round) {
    uint64_t best = ~0ull;
    for (uint32_t j = lane; j < n; j += kWaveSize) {
      const uint64_t comp = ((uint64_t)scratch[j] << 32) | j;
      if ((!have_prev || comp > prev) && comp < best) best = comp;
    }
    best = wave_min_u64(best);
    prev = best;
    have_prev = true;
    const int cell = (int)(uint32_t)best;
    if (round == lane) sel0 = cell;
    if (round == lane + kWaveSize) sel1 = cell;
  }
  wave_lds_fence();
}

// split(key) with the two Threefry blocks in lanes 0 and 1 of ONE evaluation:
// children (y0[0], y0[1]) and (y1[0], y1[1]) of blocks (0, 2) and (1, 3).
__device__ __forceinline__ void split2_lanes(Key2 k, int lane, Key2 &child0, Key2 &child1) {
  const uint32_t x = (uint32_t)(lane & 1);
  const uint2 y = threefry_call(k.a, k.b, x, x + 2u);
  const uint32_t y0 = y.x, y1 = y.y;
  child0.a = __builtin_amdgcn_readlane(y0, 0);
  child0.b = __builtin_amdgcn_readlane(y0, 1);
  child1.a = __builtin_amdgcn_readlane(y1, 0);
  child1.b = __builtin_amdgcn_readlane(y1, 1);
}

// ---------------------------------------------------------------------------------------------
// The random walk (JAXI/parallel_random_walk.py:60-145, 192-230, 262-374)
// ---------------------------------------------------------------------------------------------
struct Walker {
  int r0, c0;  // start (row, col)
  int r, c;    // current head (row, col)
  int P;       // padded index of the head
};

constexpr int kPrngDumpSlot = 2 * kMaxN;          // where non-split lanes park their stores
constexpr int kPrngScratchWords = 2 * kMaxN + 4;  // split(key, N) words of one iteration + dump

// One round of the lane-parallel PRNG pipeline.  A walk iteration consumes, per board,
//   keys = split(k_t, N)  (N Threefry blocks),  u_i = uniform(keys[i])  (N blocks),
//   k_{t+1} = split(k_t)[1]  (2 blocks)
// none of which depends on the board.  Instead of 5 dependent blocks per lane the 2N + 2 blocks
// of one round are spread over the lanes of the wave as "virtual slots" and ONE Threefry
// evaluation per 64 slots serves three pipeline stages at once:
//   slot s < N          uniform(keys(t+1)[s])            -> u_s(t+1), lands in agent lane s
//   slot N + j, j < N   block j of split(k_{t+2}, N)     -> W (LDS), read by the next round
//   slot 2N + x, x < 2  block x of split(k_{t+2})        -> k_{t+3} (wave-uniform, via readlane)
// so the randomness of iteration t+1 is finished while iteration t waits on its LDS stencil.
// kSingle: 2N + 2 <= 64, all slots fit one call (no loop).
template <bool kSingle>
__device__ __forceinline__ float prng_round(uint32_t *W, int N, int lane, Key2 &kcur) {
  const int twoN = 2 * N;
  float u = 0.0f;
  uint32_t nk0 = 0u, nk1 = 0u;
  const int last_base = kSingle ? 0 : ((twoN + 1) & ~(kWaveSize - 1));
#pragma unroll 1
  for (int base = 0; base <= last_base; base += kWaveSize) {
    const int s = base + lane;
    const bool r_uni = s < N;
    const bool r_split = !r_uni && s < twoN;
    // child s of split(k, N); the load is unconditional (lane-clamped index) so that the key
    // stays a VALUE select -- a pointer select here turns into a flat load through scratch
    const uint2 kk = *reinterpret_cast<const uint2 *>(&W[r_uni ? 2 * s : 0]);
    const uint32_t ka = r_uni ? kk.x : kcur.a;
    const uint32_t kb = r_uni ? kk.y : kcur.b;
    // counters: uniform (0, pad 0) | split block j = s - N: (j, N + j) | split(k) block x: (x, x+2)
    const uint32_t c0 = r_uni ? 0u : r_split ? (uint32_t)(s - N) : (uint32_t)(s - twoN);
    const uint32_t c1 = r_uni ? 0u : r_split ? (uint32_t)s : (uint32_t)(s - twoN + 2);
    const uint2 y = threefry_call(ka, kb, c0, c1);
    const uint32_t y0 = y.x, y1 = y.y;
    if (r_uni) u = __fsub_rn(__uint_as_float((y0 >> 9) | 0x3F800000u), 1.0f);
    // out = y0-half ++ y1-half; other roles store to a dump slot (branch-free)
    W[r_split ? s - N : kPrngDumpSlot] = y0;
    W[r_split ? s : kPrngDumpSlot + 1] = y1;
    if (base == (twoN & ~(kWaveSize - 1))) {  // wave-uniform: the call that holds slots 2N, 2N+1
      nk0 = __builtin_amdgcn_readlane(y1, twoN & (kWaveSize - 1));
      nk1 = __builtin_amdgcn_readlane(y1, (twoN & (kWaveSize - 1)) + 1);
    }
  }
  wave_lds_fence();
  kcur.a = nk0;
  kcur.b = nk1;
  return u;
}

// The single-call round cut in three pieces for software pipelining inside the walk loop:
//   prng_issue   ds_read of this lane's uniform key (issued BEFORE the stencil reads so that
//                waiting for it does not wait for them)
//   prng_first   counters/keys + Threefry rounds 1-8   (runs under the stencil's LDS latency)
//   prng_second  rounds 9-20, W stores, chain readlanes (runs under the move's ds_max + read-back)
struct PrngLanes {  // per-lane constants of the single-call layout
  bool r_uni, r_split;
  int w_read, w_y0, w_y1;  // LDS word indices
  uint32_t c0, c1;         // counters of the non-uniform roles
  int chain_lane;
};

__device__ __forceinline__ PrngLanes prng_lanes(int N, int lane) {
  PrngLanes p;
  const int twoN = 2 * N, s = lane;
  p.r_uni = s < N;
  p.r_split = !p.r_uni && s < twoN;
  p.w_read = p.r_uni ? 2 * s : 0;
  p.w_y0 = p.r_split ? s - N : kPrngDumpSlot;
  p.w_y1 = p.r_split ? s : kPrngDumpSlot + 1;
  p.c0 = p.r_uni ? 0u : p.r_split ? (uint32_t)(s - N) : (uint32_t)(s - twoN);
  p.c1 = p.r_uni ? 0u : p.r_split ? (uint32_t)s : (uint32_t)(s - twoN + 2);
  p.chain_lane = twoN;
  return p;
}

__device__ __forceinline__ uint2 prng_issue(const uint32_t *W, const PrngLanes &p) {
  return *reinterpret_cast<const uint2 *>(&W[p.w_read]);
}

__device__ __forceinline__ TfState prng_first(const PrngLanes &p, uint2 kk, Key2 kcur) {
  return threefry_first_half(p.r_uni ? kk.x : kcur.a, p.r_uni ? kk.y : kcur.b, p.c0, p.c1);
}

__device__ __forceinline__ float prng_second(uint32_t *W, const PrngLanes &p, const TfState &t,
                                             Key2 &kcur) {
  uint32_t y0, y1;
  threefry_second_half(t, y0, y1);
  W[p.w_y0] = y0;
  W[p.w_y1] = y1;
  kcur.a = __builtin_amdgcn_readlane(y1, p.chain_lane);
  kcur.b = __builtin_amdgcn_readlane(y1, p.chain_lane + 1);
  return __fsub_rn(__uint_as_float((y0 >> 9) | 0x3F800000u), 1.0f);  // meaningful in r_uni lanes
}

// Runs generate_board(key) on `board` (LDS, padded).  On return the board holds the solved grid
// with the head/target encodings of :435-447 applied; lanes i < N hold agent i's start / final
// position.  Returns the while_loop trip count.
//
// During the walk every cell of wire i holds the single value 3i+1: the walk never needs to tell
// a head from a path cell (availability only asks "empty?" and "own wire?"), and the value is
// still monotone in the agent id, which is all the max-join needs.  Heads and targets are
// written once at the end from the positions kept in registers.
template <bool kSingle>
__device__ __forceinline__ int prw_walk_impl(uint32_t *board, uint32_t *W, const Geo &g, int N,
                                             Key2 key, int lane, Walker &w) {
  Key2 kB, kcur, unused, sub;
  split2_lanes(key, lane, kB, kcur);  // key, step_key = split(key)   (:73); kcur = k_0
  const bool is_agent = lane < N;

  // _initialise_agents (:147-186): N distinct cells = first N of the shuffled arange(m*m);
  // _shuffle's round does key, sub = split(key) first
  split2_lanes(kB, lane, unused, sub);
  int sel0, sel1;
  shuffle_prefix(board, W, g, sub, N, g.thr, lane, sel0, sel1);
#if defined(RBG_PROBE_STAGE) && RBG_PROBE_STAGE == 1  // developer probe: stop after the shuffle
  w.r0 = w.c0 = w.r = w.c = sel0; w.P = 0;
  return 0;
#endif
  board_clear(board, g, lane);
  const uint32_t wire = 3u * (uint32_t)lane + 1u;
  w.r0 = w.c0 = w.r = w.c = 0;
  w.P = pad_index(g, 0, 0);
  if (is_agent) {
    w.r0 = div_m(g, sel0);
    w.c0 = sel0 - w.r0 * g.m;
    w.r = w.r0;
    w.c = w.c0;
    w.P = pad_index(g, w.r, w.c);
    board[w.P] = wire;
  }
  wave_lds_fence();

  // pipeline prologue: W <- split(k_0, N), kcur <- k_1; then u <- u(0), W <- split(k_1, N), ...
#if defined(RBG_PROBE_STAGE) && RBG_PROBE_STAGE == 2  // developer probe: stop before the prologue
  return 0;
#endif
  (void)prng_round<kSingle>(W, N, lane, kcur);
  float u = prng_round<kSingle>(W, N, lane, kcur);

  const int S = g.S;
  const PrngLanes pl = prng_lanes(N, lane);
  int iters = 0;
  // Every iteration that runs moves at least the highest-id mover into an empty cell, so the
  // loop ends after at most n - N trips; the explicit bound keeps a faulty board from spinning.
#ifdef RBG_PROBE_MAX_ITERS  // developer probe builds only (tools/probe_walk.py)
  const int max_iters = g.n < RBG_PROBE_MAX_ITERS ? g.n : RBG_PROBE_MAX_ITERS;
#else
  const int max_iters = g.n;
#endif
  for (; iters < max_iters; ++iters) {
    // randomness of the NEXT iteration is independent of the board: its LDS read goes first ...
    uint2 kk = make_uint2(0u, 0u);
    if (kSingle) kk = prng_issue(W, pl);
    // ---- _available_cells for [up, down, left, right] (:293-374) -------------------------
    // candidate q is available iff it is empty and touches no cell of this wire other than the
    // head itself ("doubling back" counts the head once, so the other three must be foreign).
    const int P = w.P;
    const uint32_t vu = board[P - S], vd = board[P + S], vl = board[P - 1], vr = board[P + 1];
    const uint32_t uu = board[P - 2 * S], dd = board[P + 2 * S];
    const uint32_t ll = board[P - 2], rr = board[P + 2];
    const uint32_t ul = board[P - S - 1], ur = board[P - S + 1];
    const uint32_t dl = board[P + S - 1], dr = board[P + S + 1];

    // ... and the first Threefry rounds run while the stencil reads are in flight
    TfState tf = {0u, 0u, 0u, 0u, 0u};
    float u_next = 0.0f;
    __builtin_amdgcn_sched_barrier(0);
    if (kSingle) {
      tf = prng_first(pl, kk, kcur);
      // pin the half-finished state here: without it the optimiser sinks these rounds below
      // the loop's exit test, i.e. out of the stencil's latency shadow
      asm volatile("" : "+v"(tf.x0), "+v"(tf.x1));
    } else {
      u_next = prng_round<false>(W, N, lane, kcur);
    }
    __builtin_amdgcn_sched_barrier(0);

    const bool own_ul = ul == wire, own_ur = ur == wire, own_dl = dl == wire, own_dr = dr == wire;
    const bool a_up = is_agent && vu == 0u && !(uu == wire || own_ul || own_ur);
    const bool a_dn = is_agent && vd == 0u && !(dd == wire || own_dl || own_dr);
    const bool a_lf = is_agent && vl == 0u && !(ll == wire || own_ul || own_dl);
    const bool a_rt = is_agent && vr == 0u && !(rr == wire || own_ur || own_dr);
    const int c1 = (int)a_up, c2 = c1 + (int)a_dn, c3 = c2 + (int)a_lf, c4 = c3 + (int)a_rt;

    // ---- _continue_stepping (:192-203): stop when no agent has an available cell ----------
    if (__ballot(c4 > 0) == 0ull) break;

    // ---- _select_action (:205-230): keys = split(key, N); choice(p = available) -----------
    // choice: cum = cumsum(p); r = cum[-1] * (1 - uniform(key_i)); first index with cum >= r.
    const float rsel = __fmul_rn((float)c4, __fsub_rn(1.0f, u));
    int dir = 3;
    if ((float)c3 >= rsel) dir = 2;
    if ((float)c2 >= rsel) dir = 1;
    if ((float)c1 >= rsel) dir = 0;
    const bool mover = c4 > 0;  // a blocked agent's -1 sentinel maps to NOOP / invalid UP (A.5)
    const int dP = dir == 0 ? -S : dir == 1 ? S : dir == 2 ? -1 : 1;
    // non-movers aim at word 0, a wall: max(wall, wire) leaves it unchanged (branch-free)
    const int Q = mover ? P + dP : 0;

    // ---- _step_agents (:101-145): contested cell -> highest id wins, losers stay ----------
    atomicMax(&board[Q], wire);
    wave_lds_fence();
    const uint32_t after = board[Q];
    __builtin_amdgcn_sched_barrier(0);
    // the remaining Threefry rounds hide the ds_max + read-back round trip
    if (kSingle) u_next = prng_second(W, pl, tf, kcur);
    __builtin_amdgcn_sched_barrier(0);
    const bool won = after == wire;  // never true for the wall
    w.P = won ? Q : P;
    w.r += won ? (dir == 0 ? -1 : dir == 1 ? 1 : 0) : 0;
    w.c += won ? (dir == 2 ? -1 : dir == 3 ? 1 : 0) : 0;
    wave_lds_fence();
    u = u_next;  // _step (:98) carried split(key)[1]: that is kcur's pipeline
  }

  // update_solved_board_with_head_target_encodings (:435-447): heads first, then targets
  if (is_agent) board[pad_index(g, w.r0, w.c0)] = wire + 1u;  // get_position(id)
  wave_lds_fence();
  if (is_agent) board[w.P] = wire + 2u;  // get_target(id); wins when the agent never moved
  wave_lds_fence();
  return iters;
}

__device__ __forceinline__ int prw_walk(uint32_t *board, uint32_t *W, const Geo &g, int N,
                                        Key2 key, int lane, Walker &w) {
  if (2 * N + 2 <= kWaveSize) return prw_walk_impl<true>(board, W, g, N, key, lane, w);
  return prw_walk_impl<false>(board, W, g, N, key, lane, w);
}

// ---------------------------------------------------------------------------------------------
// Dense int32 write-out of a padded board.  kTrain: keep only heads and targets
// (parallel_random_walk_generator.py:63-64): path cells (v % 3 == 1) become 0.
// ---------------------------------------------------------------------------------------------
template <bool kTrain>
__device__ __forceinline__ void store_grid(const uint32_t *board, const Geo &g, int lane,
                                           int32_t *__restrict__ out) {
  if ((g.m & 3) == 0) {
    // 4 consecutive cells never straddle a row; LDS rows are 16-B aligned (S % 4 == 0)
    for (int idx = lane * 4; idx < g.n; idx += kWaveSize * 4) {
      const int r = div_m(g, idx);
      const uint4 v = *reinterpret_cast<const uint4 *>(&board[pad_index(g, r, idx - r * g.m)]);
      int4 o;
      o.x = (kTrain && v.x % 3u == 1u) ? 0 : (int)v.x;
      o.y = (kTrain && v.y % 3u == 1u) ? 0 : (int)v.y;
      o.z = (kTrain && v.z % 3u == 1u) ? 0 : (int)v.z;
      o.w = (kTrain && v.w % 3u == 1u) ? 0 : (int)v.w;
      *reinterpret_cast<int4 *>(&out[idx]) = o;
    }
  } else {
    for (int idx = lane; idx < g.n; idx += kWaveSize) {
      const int r = div_m(g, idx);
      const uint32_t v = board[pad_index(g, r, idx - r * g.m)];
      out[idx] = (kTrain && v % 3u == 1u) ? 0 : (int)v;
    }
  }
}

// HBM int32 grid -> padded LDS board
__device__ __forceinline__ void load_grid(uint32_t *board, const Geo &g, int lane,
                                          const int32_t *__restrict__ in) {
  for (int w = lane; w < g.words; w += kWaveSize) board[w] = kWall;
  wave_lds_fence();
  if ((g.m & 3) == 0) {
    for (int idx = lane * 4; idx < g.n; idx += kWaveSize * 4) {
      const int r = div_m(g, idx);
      const int4 v = *reinterpret_cast<const int4 *>(&in[idx]);
      *reinterpret_cast<uint4 *>(&board[pad_index(g, r, idx - r * g.m)]) =
          make_uint4((uint32_t)v.x, (uint32_t)v.y, (uint32_t)v.z, (uint32_t)v.w);
    }
  } else {
    for (int idx = lane; idx < g.n; idx += kWaveSize) {
      const int r = div_m(g, idx);
      board[pad_index(g, r, idx - r * g.m)] = (uint32_t)in[idx];
    }
  }
  wave_lds_fence();
}

// ---------------------------------------------------------------------------------------------
// Connector observation pieces ([3P] jumanji connector/env.py)
// ---------------------------------------------------------------------------------------------
struct AgentRegs {
  int r0, c0;  // start
  int tr, tc;  // target
  int r, c;    // position
};

// _get_action_mask: bits 0..4 = NOOP, UP, RIGHT, DOWN, LEFT
__device__ __forceinline__ uint32_t action_mask_bits(const uint32_t *board, const Geo &g,
                                                     const AgentRegs &a, int lane) {
  const bool connected = a.r == a.tr && a.c == a.tc;
  const int P = pad_index(g, a.r, a.c);
  const uint32_t tgt = 3u * (uint32_t)lane + 3u;
  const uint32_t vu = board[P - g.S], vr = board[P + 1], vd = board[P + g.S], vl = board[P - 1];
  uint32_t bits = 1u;
  if (!connected) {
    bits |= (uint32_t)(vu == 0u || vu == tgt) << 1;
    bits |= (uint32_t)(vr == 0u || vr == tgt) << 2;
    bits |= (uint32_t)(vd == 0u || vd == tgt) << 3;
    bits |= (uint32_t)(vl == 0u || vl == tgt) << 4;
  }
  return bits;
}

// _obs_from_grid: agent i sees v > 0 as ((v - 1 - 3 i) mod 3N) + 1 (own cells become 1, 2, 3)
__device__ __forceinline__ int obs_value(uint32_t v, int i3, int n3) {
  int t = (int)v - 1 - i3;
  t += t < 0 ? n3 : 0;
  return v == 0u ? 0 : t + 1;
}

__device__ __forceinline__ void store_obs(const uint32_t *board, const Geo &g, int N, int lane,
                                          int32_t *__restrict__ obs) {
  const int n3 = 3 * N;
  if ((g.m & 3) == 0) {
    for (int idx = lane * 4; idx < g.n; idx += kWaveSize * 4) {
      const int r = div_m(g, idx);
      const uint4 v = *reinterpret_cast<const uint4 *>(&board[pad_index(g, r, idx - r * g.m)]);
      for (int i = 0; i < N; ++i) {
        int4 o;
        o.x = obs_value(v.x, 3 * i, n3);
        o.y = obs_value(v.y, 3 * i, n3);
        o.z = obs_value(v.z, 3 * i, n3);
        o.w = obs_value(v.w, 3 * i, n3);
        *reinterpret_cast<int4 *>(&obs[(size_t)i * g.n + idx]) = o;
      }
    }
  } else {
    for (int idx = lane; idx < g.n; idx += kWaveSize) {
      const int r = div_m(g, idx);
      const uint32_t v = board[pad_index(g, r, idx - r * g.m)];
      for (int i = 0; i < N; ++i) obs[(size_t)i * g.n + idx] = obs_value(v, 3 * i, n3);
    }
  }
}

// number of PATH cells (v % 3 == 1), the `total_path_length` extra
__device__ __forceinline__ int count_path_cells(const uint32_t *board, const Geo &g, int lane) {
  int cnt = 0;
  for (int idx = lane; idx < g.n; idx += kWaveSize) {
    const int r = div_m(g, idx);
    cnt += (int)(board[pad_index(g, r, idx - r * g.m)] % 3u == 1u);
  }
  return wave_sum_i32(cnt);
}

struct TimestepPtrs {
  int8_t *step_type;
  float *reward;
  float *discount;
  int32_t *obs_grid;
  uint8_t *action_mask;
  int32_t *obs_step_count;
  int32_t *num_connections;
  float *ratio_connections;
  int32_t *total_path_length;
};
struct StatePtrs {
  int32_t *grid;
  int32_t *step_count;
  int32_t *start;
  int32_t *target;
  int32_t *position;
  uint32_t *key;
};

__device__ __forceinline__ void store_mask(const TimestepPtrs &ts, int b, int N, int lane,
                                           uint32_t bits) {
  if (ts.action_mask && lane < N) {
    uint8_t *p = ts.action_mask + ((size_t)b * N + lane) * 5;
#pragma unroll
    for (int a = 0; a < 5; ++a) p[a] = (uint8_t)((bits >> a) & 1u);
  }
}

__device__ __forceinline__ void store_agents(const StatePtrs &st, int b, int N, int lane,
                                             const AgentRegs &a, bool with_start_target) {
  if (lane < N) {
    const size_t o = ((size_t)b * N + lane) * 2;
    if (with_start_target) {
      *reinterpret_cast<int2 *>(&st.start[o]) = make_int2(a.r0, a.c0);
      *reinterpret_cast<int2 *>(&st.target[o]) = make_int2(a.tr, a.tc);
    }
    *reinterpret_cast<int2 *>(&st.position[o]) = make_int2(a.r, a.c);
  }
}

// Builds a fresh State from a finished walk (parallel_random_walk_generator.py:58-77) and turns
// the LDS board into the training grid (heads + targets only).
__device__ __forceinline__ void board_to_training(uint32_t *board, const Geo &g, int lane) {
  for (int idx = lane; idx < g.n; idx += kWaveSize) {
    const int r = div_m(g, idx);
    const int p = pad_index(g, r, idx - r * g.m);
    const uint32_t v = board[p];
    if (v % 3u == 1u) board[p] = 0u;
  }
  wave_lds_fence();
}

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
};

extern __shared__ __attribute__((aligned(16))) uint32_t g_lds[];

// per-wave LDS slab: padded board (16-B aligned) followed by the PRNG pipeline's W scratch
__device__ __forceinline__ int wave_board_words(const Geo &g) { return (g.words + 3) & ~3; }
__device__ __forceinline__ int wave_slab_words(const Geo &g) {
  return wave_board_words(g) + kPrngScratchWords;
}

__global__ __launch_bounds__(kBlock) void prw_generate_kernel(GenArgs a) {
  const int lane = threadIdx.x & (kWaveSize - 1);
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x * kWavesPerBlock + wib;
  if (b >= a.B) return;  // wave-uniform; waves of a block never synchronise with each other
#if defined(RBG_PROBE_STAGE) && RBG_PROBE_STAGE == 0
  return;
#endif
  uint32_t *board = g_lds + wib * wave_slab_words(a.g);
  uint32_t *W = board + wave_board_words(a.g);
  const Geo g = a.g;

  Key2 key;
  key.a = a.keys[2 * (size_t)b];
  key.b = a.keys[2 * (size_t)b + 1];
  if (a.split_first) {  // key, pos_key = split(key)  (generator :52)
    Key2 unused;
    split2_lanes(key, lane, key, unused);
  }

  Walker w;
  const int iters = prw_walk(board, W, g, a.N, key, lane, w);

  const size_t gofs = (size_t)b * g.n;
  if (a.solved_grid) store_grid<false>(board, g, lane, a.solved_grid + gofs);
  if (a.train_grid) store_grid<true>(board, g, lane, a.train_grid + gofs);
  if (lane < a.N) {
    const size_t o = ((size_t)b * a.N + lane) * 2;
    *reinterpret_cast<int2 *>(&a.start[o]) = make_int2(w.r0, w.c0);
    *reinterpret_cast<int2 *>(&a.target[o]) = make_int2(w.r, w.c);
    if (a.position) *reinterpret_cast<int2 *>(&a.position[o]) = make_int2(w.r0, w.c0);
  }
  if (lane == 0) {
    if (a.key_out) *reinterpret_cast<uint2 *>(&a.key_out[2 * (size_t)b]) = make_uint2(key.a, key.b);
    if (a.step_count) a.step_count[b] = 0;
    if (a.iters) a.iters[b] = iters;
  }
}

struct UniformArgs {
  const uint32_t *keys;
  int B, N;
  Geo g;
  int32_t *train_grid;
  int32_t *start;
  int32_t *target;
  uint32_t *key_out;
};

// UniformRandomGenerator.__call__ (uniform_generator.py:70-109)
__global__ __launch_bounds__(kBlock) void uniform_generate_kernel(UniformArgs a) {
  const int lane = threadIdx.x & (kWaveSize - 1);
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x * kWavesPerBlock + wib;
  if (b >= a.B) return;
  uint32_t *board = g_lds + wib * wave_slab_words(a.g);
  uint32_t *W = board + wave_board_words(a.g);
  const Geo g = a.g;
  Key2 key;
  key.a = a.keys[2 * (size_t)b];
  key.b = a.keys[2 * (size_t)b + 1];
  Key2 kept, pos_key, unused, sub;
  split2_lanes(key, lane, kept, pos_key);
  split2_lanes(pos_key, lane, unused, sub);
  int sel0, sel1;
  shuffle_prefix(board, W, g, sub, 2 * a.N, g.thr, lane, sel0, sel1);
  // entry e of the prefix: e < N -> start of agent e, else target of agent e - N
  const int e_t = lane + a.N;  // prefix entry holding this agent's target
  const int t_lo = __shfl(sel0, e_t & 63), t_hi = __shfl(sel1, e_t & 63);
  const int tcell = e_t < kWaveSize ? t_lo : t_hi;
  board_clear(board, g, lane);
  if (lane < a.N) {
    const int r0 = div_m(g, sel0), c0 = sel0 - r0 * g.m;
    const int tr = div_m(g, tcell), tc = tcell - tr * g.m;
    board[pad_index(g, r0, c0)] = 3u * lane + 2u;
    board[pad_index(g, tr, tc)] = 3u * lane + 3u;  // distinct cells: no ordering issue
    const size_t o = ((size_t)b * a.N + lane) * 2;
    *reinterpret_cast<int2 *>(&a.start[o]) = make_int2(r0, c0);
    *reinterpret_cast<int2 *>(&a.target[o]) = make_int2(tr, tc);
  }
  wave_lds_fence();
  if (a.train_grid) store_grid<false>(board, g, lane, a.train_grid + (size_t)b * g.n);
  if (lane == 0 && a.key_out)
    *reinterpret_cast<uint2 *>(&a.key_out[2 * (size_t)b]) = make_uint2(kept.a, kept.b);
}

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
};

// Observation + extras of the state held in (board, ag); `terminal` keeps reward/discount/
// step_type/extras of the step that just ended when the env was auto-reset.
__device__ __forceinline__ void write_observation(const uint32_t *board, const Geo &g, int N,
                                                  int b, int lane, const AgentRegs &ag,
                                                  int step_count, const TimestepPtrs &ts) {
  const uint32_t bits = lane < N ? action_mask_bits(board, g, ag, lane) : 1u;
  store_mask(ts, b, N, lane, bits);
  if (ts.obs_grid) store_obs(board, g, N, lane, ts.obs_grid + (size_t)b * N * g.n);
  if (ts.obs_step_count && lane == 0) ts.obs_step_count[b] = step_count;
}

__device__ __forceinline__ void write_extras(const uint32_t *board, const Geo &g, int N, int b,
                                             int lane, const AgentRegs &ag,
                                             const TimestepPtrs &ts) {
  const bool connected = lane < N && ag.r == ag.tr && ag.c == ag.tc;
  const int conn = __popcll(__ballot(connected));
  const int plen = ts.total_path_length ? count_path_cells(board, g, lane) : 0;
  if (lane == 0) {
    if (ts.num_connections) ts.num_connections[b] = conn;
    if (ts.ratio_connections) ts.ratio_connections[b] = __fdiv_rn((float)conn, (float)N);
    if (ts.total_path_length) ts.total_path_length[b] = plen;
  }
}

// Generator -> State: runs the walk for `key_in` (the generator's argument), leaves the training
// grid in LDS, fills agent registers, writes the State to HBM.
__device__ __forceinline__ void generate_state(uint32_t *board, uint32_t *W, const Geo &g, int N, int b,
                                               int lane, Key2 key_in, const StatePtrs &out,
                                               AgentRegs &ag) {
  Key2 key, unused;
  split2_lanes(key_in, lane, key, unused);
  Walker w;
  prw_walk(board, W, g, N, key, lane, w);
  board_to_training(board, g, lane);
  ag.r0 = w.r0; ag.c0 = w.c0;
  ag.tr = w.r;  ag.tc = w.c;
  ag.r = w.r0;  ag.c = w.c0;
  store_grid<false>(board, g, lane, out.grid + (size_t)b * g.n);
  store_agents(out, b, N, lane, ag, true);
  if (lane == 0) {
    out.step_count[b] = 0;
    *reinterpret_cast<uint2 *>(&out.key[2 * (size_t)b]) = make_uint2(key.a, key.b);
  }
}

__global__ __launch_bounds__(kBlock) void connector_kernel(EnvArgs a) {
  const int lane = threadIdx.x & (kWaveSize - 1);
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x * kWavesPerBlock + wib;
  if (b >= a.B) return;
  uint32_t *board = g_lds + wib * wave_slab_words(a.g);
  uint32_t *W = board + wave_board_words(a.g);
  const Geo g = a.g;
  const int N = a.N;
  const bool is_agent = lane < N;
  AgentRegs ag = {0, 0, 0, 0, 0, 0};

  if (a.mode == 0) {  // ---- Connector.reset ---------------------------------------------------
    Key2 kin;
    kin.a = a.keys[2 * (size_t)b];
    kin.b = a.keys[2 * (size_t)b + 1];
    generate_state(board, W, g, N, b, lane, kin, a.out, ag);
    write_observation(board, g, N, b, lane, ag, 0, a.ts);
    write_extras(board, g, N, b, lane, ag, a.ts);
    if (is_agent) {
      if (a.ts.reward) a.ts.reward[(size_t)b * N + lane] = 0.0f;
      if (a.ts.discount) a.ts.discount[(size_t)b * N + lane] = 1.0f;
    }
    if (lane == 0 && a.ts.step_type) a.ts.step_type[b] = 0;
    return;
  }

  // ---- load the state ------------------------------------------------------------------------
  load_grid(board, g, lane, a.in.grid + (size_t)b * g.n);
  if (is_agent) {
    const size_t o = ((size_t)b * N + lane) * 2;
    const int2 s = *reinterpret_cast<const int2 *>(&a.in.start[o]);
    const int2 t = *reinterpret_cast<const int2 *>(&a.in.target[o]);
    const int2 p = *reinterpret_cast<const int2 *>(&a.in.position[o]);
    ag.r0 = s.x; ag.c0 = s.y; ag.tr = t.x; ag.tc = t.y; ag.r = p.x; ag.c = p.y;
  }
  int step_count = a.in.step_count[b];

  if (a.mode == 1) {  // ---- observe an existing state (restart timestep) ----------------------
    write_observation(board, g, N, b, lane, ag, step_count, a.ts);
    write_extras(board, g, N, b, lane, ag, a.ts);
    if (is_agent) {
      if (a.ts.reward) a.ts.reward[(size_t)b * N + lane] = 0.0f;
      if (a.ts.discount) a.ts.discount[(size_t)b * N + lane] = 1.0f;
    }
    if (lane == 0 && a.ts.step_type) a.ts.step_type[b] = 0;
    return;
  }

  // ---- Connector.step --------------------------------------------------------------------------
  // _step_agent: move iff the action is not NOOP, the destination is in bounds and empty or the
  // agent's own target, and the agent is not connected yet.
  const uint32_t head = 3u * (uint32_t)lane + 2u;
  const bool was_connected = is_agent && ag.r == ag.tr && ag.c == ag.tc;
  int act = is_agent ? a.action[(size_t)b * N + lane] : 0;
  act = act < 0 ? 0 : act > 4 ? 4 : act;  // lax.switch clamps its index (move_position)
  const int dr = act == 1 ? -1 : act == 3 ? 1 : 0;
  const int dc = act == 2 ? 1 : act == 4 ? -1 : 0;
  const int P = pad_index(g, ag.r, ag.c);
  const int Q = P + dr * g.S + dc;
  const uint32_t vq = board[Q];
  const bool to_empty = vq == 0u;
  const bool to_target = vq == head + 1u;
  const bool mover = is_agent && act != 0 && !was_connected && (to_empty || to_target);
  // Only an EMPTY cell can be contested (a target cell admits its owner alone): max-join.
  if (mover && to_empty) atomicMax(&board[Q], head);
  wave_lds_fence();
  if (mover && (to_target || board[Q] == head)) {
    board[Q] = head;  // overwrites the own target on arrival (move_agent)
    board[P] = head - 1u;
    ag.r += dr;
    ag.c += dc;
  }
  wave_lds_fence();
  step_count += 1;

  const bool connected = is_agent && ag.r == ag.tr && ag.c == ag.tc;
  // DenseRewardFn
  const float rew = __fadd_rn(a.p.connected_reward * (float)(!was_connected && connected),
                              a.p.timestep_reward * (float)(!was_connected));
  const uint32_t bits = is_agent ? action_mask_bits(board, g, ag, lane) : 1u;
  const bool done = connected || (bits >> 1) == 0u;  // connected_or_blocked
  const bool all_done = __ballot(is_agent && !done) == 0ull;
  const bool last = all_done || step_count >= a.p.time_limit;

  if (is_agent) {
    if (a.ts.reward) a.ts.reward[(size_t)b * N + lane] = rew;
    if (a.ts.discount) a.ts.discount[(size_t)b * N + lane] = (last || done) ? 0.0f : 1.0f;
  }
  if (lane == 0 && a.ts.step_type) a.ts.step_type[b] = last ? 2 : 1;
  write_extras(board, g, N, b, lane, ag, a.ts);

  if (last && a.p.auto_reset) {
    // VmapAutoResetWrapper._auto_reset: key, _ = split(state.key); state, ts = reset(key);
    // only the observation of the terminal timestep is replaced.
    Key2 sk;
    sk.a = a.in.key[2 * (size_t)b];
    sk.b = a.in.key[2 * (size_t)b + 1];
    Key2 rk, unused;
    split2_lanes(sk, lane, rk, unused);
    generate_state(board, W, g, N, b, lane, rk, a.out, ag);
    write_observation(board, g, N, b, lane, ag, 0, a.ts);
    return;
  }

  // ---- store the stepped state and its observation -------------------------------------------
  store_grid<false>(board, g, lane, a.out.grid + (size_t)b * g.n);
  store_agents(a.out, b, N, lane, ag, true);
  if (lane == 0) {
    a.out.step_count[b] = step_count;
    if (a.out.key != a.in.key) {
      a.out.key[2 * (size_t)b] = a.in.key[2 * (size_t)b];
      a.out.key[2 * (size_t)b + 1] = a.in.key[2 * (size_t)b + 1];
    }
  }
  store_mask(a.ts, b, N, lane, bits);
  if (a.ts.obs_grid) store_obs(board, g, N, lane, a.ts.obs_grid + (size_t)b * N * g.n);
  if (a.ts.obs_step_count && lane == 0) a.ts.obs_step_count[b] = step_count;
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
  Key2 kt;
  threefry(keys[2 * b], keys[2 * b + 1], 0u, t, kt.a, kt.b);  // fold_in(key, t)
  const uint32_t bits = random_bits_word(kt, i, (uint32_t)N);
  const uint8_t *mk = mask + idx * 5;
  int cnt = 0;
#pragma unroll
  for (int a = 0; a < 5; ++a) cnt += mk[a] != 0;
  const int pick = (int)(bits % (uint32_t)(cnt > 0 ? cnt : 1));
  int chosen = 0, seen = 0;
#pragma unroll
  for (int a = 0; a < 5; ++a) {
    if (mk[a]) {
      if (seen == pick) chosen = a;
      ++seen;
    }
  }
  action[idx] = chosen;
}

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

// Candidate threshold of shuffle_prefix: about 3*count (24..44) of the n uniform 32-bit sort keys
// are expected below it; 0 disables the fast path (count > 32, or the threshold would not fit).
uint32_t prefix_threshold(int n, int count) {
  if (count > 32) return 0u;
  int expect = 3 * count;
  expect = expect < 24 ? 24 : expect > 44 ? 44 : expect;
  if (expect >= n) return 0u;
  return (uint32_t)((((uint64_t)expect) << 32) / (uint64_t)n);
}

int make_geo(int m, int N, Geo *g, int prefix_count = -1) {
  if (m < 2 || m > kMaxM) return fail(RBG_E_SHAPE, "grid size m must be in [2, 40]");
  if (N < 1 || N > kMaxN) return fail(RBG_E_SHAPE, "num_agents N must be in [1, 64]");
  if (N > m * m) return fail(RBG_E_SHAPE, "num_agents N exceeds m*m cells");
  g->m = m;
  g->n = m * m;
  g->S = m + 4;
  g->words = (m + 4) * (m + 4);
  g->magic = ((1u << 20) + (uint32_t)m - 1u) / (uint32_t)m;
  g->thr = prefix_threshold(m * m, prefix_count < 0 ? N : prefix_count);
  return 0;
}

// The m % 4 == 0 fast path moves grids with 16-byte loads/stores.
int check_align(const Geo &g, const void *p0, const void *p1, const void *p2) {
  if ((g.m & 3) != 0) return 0;
  const uintptr_t bits = (uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2;
  if (bits & 15u) return fail(RBG_E_PARAM, "grid buffers must be 16-byte aligned when m % 4 == 0");
  return 0;
}

size_t lds_bytes(const Geo &g) {
  return (size_t)kWavesPerBlock * (size_t)(((g.words + 3) & ~3) + kPrngScratchWords) *
         sizeof(uint32_t);
}

StatePtrs to_ptrs(const rbg_state *s) {
  StatePtrs p = {s->grid, s->step_count, s->start, s->target, s->position, s->key};
  return p;
}
TimestepPtrs to_ptrs(const rbg_timestep *t) {
  TimestepPtrs p = {t->step_type,      t->reward,          t->discount,
                    t->obs_grid,       t->action_mask,     t->obs_step_count,
                    t->num_connections, t->ratio_connections, t->total_path_length};
  return p;
}
bool state_complete(const rbg_state *s) {
  return s && s->grid && s->step_count && s->start && s->target && s->position && s->key;
}

}  // namespace

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
  if (B == 0) return 0;  // an empty batch may come with NULL (empty-tensor) pointers
  if (!keys || !start || !target) return fail(RBG_E_NULL, "rbg_prw_generate: NULL pointer");
  if (int rc = check_align(a.g, solved_grid, train_grid, nullptr)) return rc;
  a.keys = keys; a.B = B; a.N = N; a.split_first = split_first != 0;
  a.solved_grid = solved_grid; a.train_grid = train_grid; a.start = start; a.target = target;
  a.position = position; a.step_count = step_count; a.key_out = key_out; a.iters = iters;
  hipLaunchKernelGGL(prw_generate_kernel, dim3((B + kWavesPerBlock - 1) / kWavesPerBlock),
                     dim3(kBlock), lds_bytes(a.g), (hipStream_t)stream, a);
  return check_hip(hipGetLastError(), "rbg_prw_generate");
}

int rbg_uniform_generate(const uint32_t *keys, int B, int m, int N, int32_t *train_grid,
                         int32_t *start, int32_t *target, uint32_t *key_out, void *stream) {
  if (B < 0) return fail(RBG_E_SHAPE, "rbg_uniform_generate: B < 0");
  UniformArgs a;
  if (int rc = make_geo(m, N, &a.g, 2 * N)) return rc;
  if (2 * N > m * m) return fail(RBG_E_SHAPE, "rbg_uniform_generate: 2N exceeds m*m cells");
  if (B == 0) return 0;
  if (!keys || !start || !target) return fail(RBG_E_NULL, "rbg_uniform_generate: NULL pointer");
  if (int rc = check_align(a.g, train_grid, nullptr, nullptr)) return rc;
  a.keys = keys; a.B = B; a.N = N;
  a.train_grid = train_grid; a.start = start; a.target = target; a.key_out = key_out;
  hipLaunchKernelGGL(uniform_generate_kernel, dim3((B + kWavesPerBlock - 1) / kWavesPerBlock),
                     dim3(kBlock), lds_bytes(a.g), (hipStream_t)stream, a);
  return check_hip(hipGetLastError(), "rbg_uniform_generate");
}

static int launch_env(EnvArgs &a, void *stream, const char *what) {
  hipLaunchKernelGGL(connector_kernel, dim3((a.B + kWavesPerBlock - 1) / kWavesPerBlock),
                     dim3(kBlock), lds_bytes(a.g), (hipStream_t)stream, a);
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

}  // extern "C"
