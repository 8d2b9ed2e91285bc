// rbg_walk.h -- the ParallelRandomWalk itself, one board per sub-wave (see rbg_device.h).
//
// Reference: JAXI/parallel_random_walk.py:60-145, 192-230, 262-374, 435-447
// (JAXI = routing_board_generation/board_generation_methods/jax_implementation/board_generation).
#pragma once
#include "rbg_device.h"

namespace rbg {

// ---------------------------------------------------------------------------------------------
// First `count` entries of jax's one-round _shuffle of arange(n):
//   key, sub = split(key); sort_keys = random_bits(sub, n); stable sort -> ascending
//   (sort_key, index).
// No full sort: the first `count` entries are the `count` smallest (sort_key, index) pairs.
//   fast path  a threshold `thr` (host-computed so that a few times `count` keys fall below it)
//              filters the candidates, a ballot/popcount prefix compacts them into LDS, and
//              every lane ranks its candidate against the others (LDS broadcast reads) -- exact
//              whenever count <= #candidates <= LPB.  A board with too few (too many)
//              candidates moves its threshold (x1.5 / x0.625, then bisection) and tries again,
//              three times at most: a miss costs one more compaction, not the slow path (0.5 %
//              of the boards miss once at 16x16 / 10 agents / 32 lanes).  Up to kKeyRounds * LPB Threefry blocks the sort
//              keys never leave the registers.
//   slow path  otherwise (thr == 0, or four misses): `count` sub-wave-wide 64-bit
//              min-reductions over the keys in `scratch`.
// Sub-lane l receives entry l in sel0 and entry LPB + l in sel1.
// `scratch` needs n words (the board area is used), `aux` 2*LPB words.
// ---------------------------------------------------------------------------------------------
constexpr int kKeyRounds = 4;  // register-resident sort keys: 2 * kKeyRounds words per lane

template <int LPB>
__device__ __forceinline__ void shuffle_prefix(const SubWave<LPB> &sw, uint32_t *scratch,
                                               uint32_t *aux, const Geo &g, Key2 sub, int count,
                                               uint32_t thr, int &sel0, int &sel1,
                                               bool keys_ready = false) {
  // keys_ready: `scratch` already holds the n sort keys (another stream than jax's made them)
  const uint32_t n = (uint32_t)g.n;
  const uint32_t half = (n + 1u) >> 1;
  const int l = sw.l;
  sel0 = sel1 = -1;
  // Threefry block j yields the keys of cells j (y0) and half + j (y1)
  const bool in_regs = !keys_ready && thr != 0u && half <= (uint32_t)(kKeyRounds * LPB);  // wave-uniform
  uint32_t klo[kKeyRounds], khi[kKeyRounds];
  if (in_regs) {
#pragma unroll
    for (int k = 0; k < kKeyRounds; ++k) {
      klo[k] = khi[k] = 0u;
      if ((uint32_t)(k * LPB) < half) {
        uint32_t j = (uint32_t)(l + k * LPB);
        // computed here, not hoisted above the calls: their operands would otherwise all be live
        // across every call (the only spills of connector_rollout_kernel<64> were these)
        asm volatile("" : "+v"(j));
        const uint2 y = threefry_call(sub.a, sub.b, j < half ? j : 0u, half + j < n ? half + j : 0u);
        klo[k] = y.x;
        khi[k] = y.y;
      }
    }
  } else if (!keys_ready) {
    for (uint32_t j = l; j < half; j += LPB) {
      const bool has_hi = half + j < n;
      const uint2 y = threefry_call(sub.a, sub.b, j, has_hi ? half + j : 0u);
      scratch[j] = y.x;
      if (has_hi) scratch[half + j] = y.y;
    }
    wave_lds_fence();
  }

  bool ok = false;
  if (thr != 0u) {
    const uint64_t lt_mask = (1ull << l) - 1ull;
    uint32_t thr_cur = thr;                   // per board
    uint32_t thr_few = 0u, thr_many = 0u;     // thresholds known to give too few / too many
#pragma unroll 1
    for (int attempt = 0; attempt < 4; ++attempt) {
      // ---- compact the candidates (key < thr_cur) as (key, index) pairs ----------------------
      int total = 0;
      auto emit = [&](bool flag, uint32_t key, uint32_t idx) {
        const uint64_t mask = sw.ballot(flag);
        const int pos = total + __popcll(mask & lt_mask);
        if (flag && pos < LPB) *reinterpret_cast<uint2 *>(&aux[2 * pos]) = make_uint2(key, idx);
        total += __popcll(mask);
      };
      if (in_regs) {
#pragma unroll
        for (int k = 0; k < kKeyRounds; ++k) {
          if ((uint32_t)(k * LPB) < half) {
            const uint32_t j = (uint32_t)(l + k * LPB);
            emit(j < half && klo[k] < thr_cur, klo[k], j);
          }
        }
#pragma unroll
        for (int k = 0; k < kKeyRounds; ++k) {
          if ((uint32_t)(k * LPB) < half) {
            const uint32_t j = (uint32_t)(l + k * LPB);
            emit(j < half && half + j < n && khi[k] < thr_cur, khi[k], half + j);
          }
        }
      } else {
        for (uint32_t base = 0; base < n; base += LPB) {
          const uint32_t j = base + l;
          const uint32_t key = j < n ? scratch[j] : 0xFFFFFFFFu;
          emit(j < n && key < thr_cur, key, j);
        }
      }
      // unused slots compare greater than every candidate: the rank loop needs no bounds test
      if (l >= total) *reinterpret_cast<uint2 *>(&aux[2 * l]) = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
      wave_lds_fence();
      const bool fits = total >= count && total <= LPB;
      // ---- rank every candidate among the candidates of its board ----------------------------
      // (all LPB slots hold a pair or the filler; two broadcast reads in flight: more costs the rollout kernel its 4th wave)
      const uint2 mine_kv = *reinterpret_cast<const uint2 *>(&aux[2 * l]);
      const uint64_t mine = ((uint64_t)mine_kv.x << 32) | mine_kv.y;
      int rank = 0;
#pragma unroll 2
      for (int c = 0; c < LPB; c += 2) {  // two pairs per read
        const uint4 o = *reinterpret_cast<const uint4 *>(&aux[2 * c]);
        rank += (int)((((uint64_t)o.x << 32) | o.y) < mine);
        rank += (int)((((uint64_t)o.z << 32) | o.w) < mine);
      }
      wave_lds_fence();
      const bool take = fits && !ok;
      if (take && l < total) aux[rank] = mine_kv.y;  // ranks of distinct pairs form a permutation
      wave_lds_fence();
      if (take && l < count) sel0 = (int)aux[l];
      wave_lds_fence();
      ok = ok || fits;
      if (__ballot(!ok) == 0ull) return;
      if (!ok) {  // x1.5 / x0.625 towards the expected count, bisection once both sides are known
        if (total < count) {
          thr_few = thr_cur;
          const uint64_t up = (uint64_t)thr_cur + (thr_cur >> 1);
          thr_cur = thr_many ? thr_few + ((thr_many - thr_few) >> 1)
                             : (up > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)up);
        } else {
          thr_many = thr_cur;
          thr_cur = thr_few ? thr_few + ((thr_many - thr_few) >> 1)
                            : (thr_cur >> 1) + (thr_cur >> 3);
        }
      }
    }
  }

  // ---- slow path (boards with ok == false): one 64-bit min-reduction per entry ---------------
  if (in_regs) {
#pragma unroll
    for (int k = 0; k < kKeyRounds; ++k) {
      const uint32_t j = (uint32_t)(l + k * LPB);
      if (j < half) {
        scratch[j] = klo[k];
        if (half + j < n) scratch[half + j] = khi[k];
      }
    }
    wave_lds_fence();
  }
  uint64_t prev = 0;
  bool have_prev = false;
  for (int round = 0; round < count; ++round) {
    uint64_t best = ~0ull;
    for (uint32_t j = l; j < n; j += LPB) {
      const uint64_t comp = ((uint64_t)scratch[j] << 32) | j;
      if ((!have_prev || comp > prev) && comp < best) best = comp;
    }
    best = sw.min_u64(best);
    prev = best;
    have_prev = true;
    const int cell = (int)(uint32_t)best;
    if (!ok && round == l) sel0 = cell;
    if (!ok && round == l + LPB) sel1 = cell;
  }
  wave_lds_fence();
}

// ---------------------------------------------------------------------------------------------
// Lane-parallel PRNG pipeline.  A walk iteration consumes, per board,
//   keys = split(k_t, N)  (N Threefry blocks),  u_i = uniform(keys[i])  (N blocks),
//   k_{t+1} = split(k_t)[1]  (2 blocks)
// none of which depends on the board.  The 2N + 2 blocks of one round are spread over the lanes
// of the board's sub-wave as "virtual slots" and ONE Threefry evaluation serves three pipeline
// stages at once:
//   slot s < N          uniform(keys(t+1)[s])            -> u_s(t+1), lands in agent lane s
//   slot N + j, j < N   block j of split(k_{t+2}, N)     -> W (LDS), read by the next round
//   slot 2N + x, x < 2  block x of split(k_{t+2})        -> k_{t+3} (uniform per board)
// so the randomness of iteration t+1 is finished while iteration t waits on its LDS stencil.
// ---------------------------------------------------------------------------------------------
template <int LPB>
struct PrngLanes {  // per-lane constants of the single-evaluation layout (2N + 2 <= LPB)
  bool r_uni;
  int w_read, w_y0, w_y1;  // LDS word indices into W
  uint32_t c0, c1;         // counters of this lane's role
};

// W: words [0, 2N) = split(k, N) as y0-half ++ y1-half (key i = words 2i, 2i+1); words 2N, 2N+1 =
// the chain key k that the split and chain slots of the NEXT evaluation start from.  Every lane
// therefore finds the key of its role with one ds_read_b64 and nothing is broadcast through
// readlanes.
// Evaluation e of a round covers the virtual slots [e * LPB, (e + 1) * LPB): slot s = l + e * LPB.
// One evaluation when 2N + 2 <= LPB; two when N <= LPB and 2N + 2 <= 2 * LPB (then every uniform
// slot lies in evaluation 0, i.e. in its agent's own lane, and both chain slots in evaluation 1).
template <int LPB>
__device__ __forceinline__ PrngLanes<LPB> prng_lanes(int N, int l, int e = 0) {
  PrngLanes<LPB> p;
  const int twoN = 2 * N, s = l + e * LPB, dump = 2 * LPB + 2 * l;  // this lane's own dump words
  p.r_uni = s < N;
  const bool r_split = !p.r_uni && s < twoN;
  const bool r_chain = s >= twoN && s < twoN + 2;
  p.w_read = p.r_uni ? 2 * s : twoN;
  // the other roles park their stores in a dump slot (branch-free); the chain slot x = s - 2N
  // keeps y1 of block x, i.e. word x of split(k)[1], at W[2N + x] = W[s]
  p.w_y0 = r_split ? s - N : dump;
  p.w_y1 = (r_split || r_chain) ? s : dump + 1;
  // counters: uniform (0, pad 0) | split block j = s - N: (j, N + j) | split(k) block x: (x, x+2)
  p.c0 = p.r_uni ? 0u : r_split ? (uint32_t)(s - N) : (uint32_t)(s - twoN);
  p.c1 = p.r_uni ? 0u : r_split ? (uint32_t)s : (uint32_t)(s - twoN + 2);
  return p;
}

// The round cut in three pieces for software pipelining inside the walk loop:
//   prng_issue   ds_read of this lane's key (issued BEFORE the stencil reads so that waiting for
//                it does not wait for them)
//   prng_first   Threefry rounds 1-8   (runs under the stencil's LDS latency)
//   prng_second  rounds 9-20 and the W stores (run under the move's ds_max + read-back)
template <int LPB>
__device__ __forceinline__ uint2 prng_issue(const uint32_t *W, const PrngLanes<LPB> &p) {
  return *reinterpret_cast<const uint2 *>(&W[p.w_read]);
}

template <int LPB>
__device__ __forceinline__ TfState prng_first(const PrngLanes<LPB> &p, uint2 kk) {
  return threefry_first_half(kk.x, kk.y, p.c0, p.c1);
}

template <int LPB>
__device__ __forceinline__ float prng_second(uint32_t *W, const PrngLanes<LPB> &p,
                                             const TfState &t) {
  uint32_t y0, y1;
  threefry_second_half(t, y0, y1);
  W[p.w_y0] = y0;
  W[p.w_y1] = y1;
  // 1 - uniform, converted here: in the shadow of the move's LDS round trip the chain of Threefry
  // rounds leaves issue slots free, the availability code that consumes it does not
  return bits_to_one_minus_uniform(y0);  // meaningful in the r_uni lanes
}

// Start of the pipeline: the chain key goes to its W slot.
__device__ __forceinline__ void prng_seed(uint32_t *W, int N, int l, Key2 k) {
  if (l == 0) *reinterpret_cast<uint2 *>(&W[2 * N]) = make_uint2(k.a, k.b);
  wave_lds_fence();
}

// One whole round, out of the loop (pipeline prologue) -- single evaluation.
template <int LPB>
__device__ __forceinline__ float prng_round_single(uint32_t *W, const PrngLanes<LPB> &p) {
  const uint2 kk = prng_issue<LPB>(W, p);
  const uint2 y = threefry_call(kk.x, kk.y, p.c0, p.c1);
  W[p.w_y0] = y.x;
  W[p.w_y1] = y.y;
  wave_lds_fence();
  return bits_to_one_minus_uniform(y.x);
}

// One whole round when 2N + 2 > 64 (full-wave boards with N >= 32): two or three evaluations.
__device__ __forceinline__ float prng_round_multi(uint32_t *W, int N, int lane, Key2 &kcur) {
  const int twoN = 2 * N, dump = 2 * kWaveSize + 2 * lane;
  float u = 0.0f;
  uint32_t nk0 = 0u, nk1 = 0u;
#pragma unroll 1
  for (int base = 0; base < twoN + 2; base += kWaveSize) {
    const int s = base + lane;
    const bool r_uni = s < N;
    const bool r_split = !r_uni && s < twoN;
    const uint2 kk = *reinterpret_cast<const uint2 *>(&W[r_uni ? 2 * s : 0]);
    const uint32_t c0 = r_uni ? 0u : r_split ? (uint32_t)(s - N) : (uint32_t)(s - twoN);
    const uint32_t c1 = r_uni ? 0u : r_split ? (uint32_t)s : (uint32_t)(s - twoN + 2);
    const uint2 y = threefry_call(r_uni ? kk.x : kcur.a, r_uni ? kk.y : kcur.b, c0, c1);
    if (r_uni) u = bits_to_one_minus_uniform(y.x);
    W[r_split ? s - N : dump] = y.x;
    W[r_split ? s : dump + 1] = y.y;
    if (base == (twoN & ~(kWaveSize - 1))) {  // wave-uniform: the call that holds slots 2N, 2N+1
      nk0 = __builtin_amdgcn_readlane(y.y, twoN & (kWaveSize - 1));
      nk1 = __builtin_amdgcn_readlane(y.y, (twoN & (kWaveSize - 1)) + 1);
    }
  }
  wave_lds_fence();
  kcur.a = nk0;
  kcur.b = nk1;
  return u;
}

// ---------------------------------------------------------------------------------------------
// The walk
// ---------------------------------------------------------------------------------------------
struct Walker {
  int r0, c0;  // start (row, col)
  int r, c;    // current head (row, col)
  int P;       // padded index of the head
  int iters;   // while_loop trip count of this lane's board
#if defined(RBG_PROBE_TIMES)
  int32_t *stamp;  // developer probe (tools/probe_walk.py --times): 8 s_memtime stamps per wave
#endif
};

#if defined(RBG_PROBE_TIMES)
#define RBG_STAMP(w, i)                                                     \
  do {                                                                      \
    if ((threadIdx.x & 63) == 0) (w).stamp[i] = (int32_t)__builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define RBG_STAMP(w, i) \
  do {                  \
  } while (0)
#endif

// Runs generate_board(key) for the board of this lane's sub-wave on `board` (LDS, padded) with
// scratch `W`.  On return the board holds the solved grid with the head/target encodings of
// :435-447 applied; sub-lanes i < N hold agent i's start / final position.
//
// During the walk every cell of wire i holds the single value 3i+1: the walk never needs to tell
// a head from a path cell (availability only asks "empty?" and "own wire?"), and the value is
// still monotone in the agent id, which is all the max-join needs.  Heads and targets are
// written once at the end from the positions kept in registers.
//
// kEvals: Threefry evaluations per iteration.  1: 2N + 2 <= LPB, software-pipelined; 2: N <= LPB and
// 2N + 2 <= 2 * LPB, the first one pipelined like the single one and the second issued behind it;
// 0: the generic loop of prng_round_multi (full-wave boards with N >= 32).
template <int LPB, int kEvals>
__device__ __forceinline__ void prw_walk_impl(const SubWave<LPB> &sw, uint32_t *board,
                                              uint32_t *W, const Geo &g, int N, Key2 key,
                                              Walker &w) {
  const int l = sw.l;
  Key2 kB, kcur, unused, sub;
  split2_lanes(sw, key, kB, kcur);  // key, step_key = split(key)   (:73); kcur = k_0
  const bool is_agent = l < N;

  // _initialise_agents (:147-186): N distinct cells = first N of the shuffled arange(m*m);
  // _shuffle's round does key, sub = split(key) first
  split2_lanes(sw, kB, unused, sub);
  RBG_STAMP(w, 2);
  int sel0, sel1;
  shuffle_prefix(sw, board, W, g, sub, N, g.thr, sel0, sel1);
  RBG_STAMP(w, 3);
  w.iters = 0;
#if defined(RBG_PROBE_STAGE) && RBG_PROBE_STAGE == 1  // developer probe: stop after the shuffle
  w.r0 = w.c0 = w.r = w.c = sel0;
  w.P = 0;
  return;
#endif
  board_clear<LPB>(board, g, l);
  const uint32_t wire = 3u * (uint32_t)l + 1u;
  w.r0 = w.c0 = w.r = w.c = 0;
  // Lanes without an agent run the same branch-free loop from a word of the wall strip between
  // two rows (second or third of its four words, rows 0 .. m-1): its four neighbours are walls,
  // so nothing is ever "available" there, the stencil reads stay inside the board, and
  // max(wall, wire) leaves the wall alone.  Distinct words as far as the strip allows (lanes
  // sharing an LDS address serialise their atomics).
  {
    const int q = div_m(g, l);
    w.P = pad_index(g, l - q * g.m, -3 + (q & 1));
  }
  if (is_agent) {
    w.r0 = div_m(g, sel0);
    w.c0 = sel0 - w.r0 * g.m;
    w.r = w.r0;
    w.c = w.c0;
    w.P = pad_index(g, w.r, w.c);
    board[w.P] = wire;
  }
  wave_lds_fence();
  RBG_STAMP(w, 4);
#if defined(RBG_PROBE_STAGE) && RBG_PROBE_STAGE == 2  // developer probe: stop before the prologue
  return;
#endif

  // pipeline prologue: W <- split(k_0, N), kcur <- k_1; then u <- u(0), W <- split(k_1, N), ...
  const PrngLanes<LPB> pl = prng_lanes<LPB>(N, l);
  float u;  // 1 - uniform of this lane's agent for the coming iteration
  const PrngLanes<LPB> pl1 = prng_lanes<LPB>(N, l, 1);  // second evaluation (kEvals == 2)
  if (kEvals >= 1) {
    prng_seed(W, N, l, kcur);
    (void)prng_round_single(W, pl);
    if (kEvals == 2) (void)prng_round_single(W, pl1);
    u = prng_round_single(W, pl);
    if (kEvals == 2) (void)prng_round_single(W, pl1);
  } else {
    (void)prng_round_multi(W, N, sw.lane, kcur);
    u = prng_round_multi(W, N, sw.lane, kcur);
  }

  RBG_STAMP(w, 5);
  const int S = g.S;
  int last_live = 0;  // last iteration (+1) in which this lane's agent could move
  // Every iteration that runs moves at least the highest-id mover of a live board into an empty
  // cell, so a board is finished after at most n - N trips; the explicit bound keeps a faulty
  // board from spinning.
#ifdef RBG_PROBE_MAX_ITERS  // developer probe builds only (tools/probe_walk.py)
  const int max_iters = g.n < RBG_PROBE_MAX_ITERS ? g.n : RBG_PROBE_MAX_ITERS;
#else
  const int max_iters = g.n;
#endif
#if defined(RBG_WALK_PRIO_LOOP)
  __builtin_amdgcn_s_setprio(RBG_WALK_PRIO_LOOP);
#endif
  for (int t = 0; t < max_iters; ++t) {
#if defined(RBG_WALK_PRIO_AT)
    // A wave still walking after RBG_WALK_PRIO_AT iterations holds a long board: from here on
    // it outranks the younger/older wave it shares the SIMD with in VALU arbitration (the launch
    // lasts as long as its slowest wave).
    if (t == RBG_WALK_PRIO_AT) __builtin_amdgcn_s_setprio(RBG_WALK_PRIO);
#endif
    // randomness of the NEXT iteration is independent of the board: its LDS read goes first ...
    uint2 kk = make_uint2(0u, 0u);
    uint2 kk1 = make_uint2(0u, 0u);
    if (kEvals >= 1) kk = prng_issue<LPB>(W, pl);
    if (kEvals == 2) kk1 = prng_issue<LPB>(W, pl1);
    // ---- _available_cells for [up, down, left, right] (:293-374) -------------------------
    // candidate q is available iff it is empty and touches no cell of this wire other than the
    // head itself ("doubling back" counts the head once, so the other three must be foreign).
    const int P = w.P;
    // five row pointers one stride apart; everything else is an immediate ds offset
    const uint32_t *mid = board + P, *up = mid - S, *dn = mid + S, *up2 = up - S, *dn2 = dn + S;
    const uint32_t vu = up[0], vd = dn[0], vl = mid[-1], vr = mid[1];
    const uint32_t uu = up2[0], dd = dn2[0];
    const uint32_t ll = mid[-2], rr = mid[2];
    const uint32_t ul = up[-1], ur = up[1];
    const uint32_t dl = dn[-1], dr = dn[1];

    // ... and the first Threefry rounds run while the stencil reads are in flight
    TfState tf = {0u, 0u, 0u, 0u, 0u};
    float u_next = 0.0f;
    __builtin_amdgcn_sched_barrier(0);
    if (kEvals >= 1) {
      tf = prng_first<LPB>(pl, kk);
      // pin the half-finished state here: without it the optimiser sinks these rounds below
      // the loop's exit test, i.e. out of the stencil's latency shadow
      asm volatile("" : "+v"(tf.x0), "+v"(tf.x1));
    } else {
      u_next = prng_round_multi(W, N, sw.lane, kcur);
    }
    __builtin_amdgcn_sched_barrier(0);

    // (lanes without an agent sit between walls: no `is_agent` term needed)
    const bool own_ul = ul == wire, own_ur = ur == wire, own_dl = dl == wire, own_dr = dr == wire;
    const bool a_up = vu == 0u && !(uu == wire || own_ul || own_ur);
    const bool a_dn = vd == 0u && !(dd == wire || own_dl || own_dr);
    const bool a_lf = vl == 0u && !(ll == wire || own_ul || own_dl);
    const bool a_rt = vr == 0u && !(rr == wire || own_ur || own_dr);
    // cumsum(p) kept in float32 (small integers are exact): it is compared with a float below
    const float c1 = a_up ? 1.0f : 0.0f, c2 = c1 + (a_dn ? 1.0f : 0.0f);
    const float c3 = c2 + (a_lf ? 1.0f : 0.0f), c4 = c3 + (a_rt ? 1.0f : 0.0f);

    // ---- _continue_stepping (:192-203): a board stops when no agent has an available cell;
    // a finished board stays finished (its grid no longer changes), so the other boards of the
    // wave simply keep going until the last one is done.
    const bool mover = c4 > 0.0f;  // a blocked agent's -1 sentinel maps to NOOP / invalid UP (A.5)
    last_live = mover ? t + 1 : last_live;
    if (__ballot(mover) == 0ull) break;

    // ---- _select_action (:205-230): keys = split(key, N); choice(p = available) -----------
    // choice: cum = cumsum(p); r = cum[-1] * (1 - uniform(key_i)); first index with cum >= r.
    // (u = 1 - uniform, rbg_device.h::bits_to_one_minus_uniform)
    const float rsel = __fmul_rn(c4, u);
    int dir = 3;
    if (c3 >= rsel) dir = 2;
    if (c2 >= rsel) dir = 1;
    if (c1 >= rsel) dir = 0;
    // step = signed byte `dir` of [-S, +S, -1, +1]: one shift + one v_bfe_i32
    const int dP = __builtin_amdgcn_sbfe((int)g.step_table, (uint32_t)dir * 8u, 8u);
    // branch-free: a lane that does not move "claims" the word it stands on -- its own head cell
    // (already = wire) or its wall word (max(wall, wire) = wall)
    const int Q = mover ? P + dP : P;

    // ---- _step_agents (:101-145): contested cell -> highest id wins, losers stay ----------
    atomicMax(&board[Q], wire);
    wave_lds_fence();
    const uint32_t after = board[Q];
    __builtin_amdgcn_sched_barrier(0);
    // the remaining Threefry rounds hide the ds_max + read-back round trip
    if (kEvals >= 1) u_next = prng_second<LPB>(W, pl, tf);
    if (kEvals == 2) {  // the slots past LPB: rest of split(k, N) and the chain
      const TfState tf1 = prng_first<LPB>(pl1, kk1);
      (void)prng_second<LPB>(W, pl1, tf1);
    }
    __builtin_amdgcn_sched_barrier(0);
    const bool won = after == wire;  // never true for a wall; a no-op for a non-moving agent
    w.P = won ? Q : P;
    wave_lds_fence();
    u = u_next;  // _step (:98) carried split(key)[1]: that is kcur's pipeline
  }
#if defined(RBG_WALK_PRIO_LOOP) && defined(RBG_WALK_PRIO_RESET)
  __builtin_amdgcn_s_setprio(0);
#endif
  RBG_STAMP(w, 6);
  w.iters = sw.max_i32(last_live);  // while_loop trip count of this lane's board

  unpad_index(g, w.P, w.r, w.c);  // the loop tracks the padded index only
  // update_solved_board_with_head_target_encodings (:435-447): heads first, then targets
  if (is_agent) board[pad_index(g, w.r0, w.c0)] = wire + 1u;  // get_position(id)
  wave_lds_fence();
  if (is_agent) board[w.P] = wire + 2u;  // get_target(id); wins when the agent never moved
  wave_lds_fence();
}

// The walk with the Threefry pipeline that fits (N, LPB) -- a wave-uniform choice.  The host picks
// the smallest LPB with 2N + 2 <= LPB; only full-wave boards need more than one evaluation.
// (Two evaluations on narrower sub-waves -- four 16x16/10 boards per wave, two 32x32/20 -- are
// bit-exact too but measured slower at every batch size: 51.6 vs 36 us and 230 vs 244 boards/us
// for 16x16/10, 136 vs 98 us for 32x32/20; fewer, longer waves lose more to the slowest board of
// the wave than they save in instructions.)
template <int LPB>
__device__ __forceinline__ void prw_walk(const SubWave<LPB> &sw, uint32_t *board, uint32_t *W,
                                         const Geo &g, int N, Key2 key, Walker &w) {
  if (2 * N + 2 <= LPB) {
    prw_walk_impl<LPB, 1>(sw, board, W, g, N, key, w);
  } else if constexpr (LPB == kWaveSize) {
    if (2 * N + 2 <= 2 * LPB) {
      prw_walk_impl<LPB, 2>(sw, board, W, g, N, key, w);  // 32 <= N <= 63
    } else {
      prw_walk_impl<LPB, 0>(sw, board, W, g, N, key, w);  // N = 64: three evaluations
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Dense int32 write-out of a padded board by its sub-wave.  kTrain: keep only heads and targets
// (parallel_random_walk_generator.py:63-64): path cells (v % 3 == 1) become 0.
// ---------------------------------------------------------------------------------------------
template <int LPB, bool kTrain>
__device__ __forceinline__ void store_grid(const uint32_t *board, const Geo &g, int l,
                                           int32_t *__restrict__ out) {
  if ((g.m & 3) == 0) {
    // 4 consecutive cells never straddle a row; LDS rows are 16-B aligned (S % 4 == 0)
    for (int idx = l * 4; idx < g.n; idx += LPB * 4) {
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
    for (int idx = l; idx < g.n; idx += LPB) {
      const int r = div_m(g, idx);
      const uint32_t v = board[pad_index(g, r, idx - r * g.m)];
      out[idx] = (kTrain && v % 3u == 1u) ? 0 : (int)v;
    }
  }
}

}  // namespace rbg
