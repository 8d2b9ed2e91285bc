// rbg_kernels_env.h -- Connector reset / observe / step (one env per wavefront, one body, three
// kernels) and the producer of the next-episode stash.
#pragma once
#include "rbg_kernels_stash.h"
#include "rbg_regen.h"

namespace {

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
  GenSpec gen;  // the env's generator (rbg_regen.h); all zero = ParallelRandomWalk
};

#ifndef RBG_ENV_WPB
#define RBG_ENV_WPB 4
#endif
constexpr int kEnvWavesPerBlock = RBG_ENV_WPB;  // waves (envs) per workgroup of the env kernels

// One body, three kernels (reset / observe / step) so that every launch population has its own
// name in a rocprofv3 kernel trace.  KIND = the generator that (re)starts episodes
// (rbg_regen.h): the ParallelRandomWalk kernels keep their round-1 names, the other generators
// of setup_train.py:112-152 get connector_{reset,step}_gen_kernel<KIND>.
enum { kModeReset = 0, kModeObserve = 1, kModeStep = 2 };

// DEEP (step only): the stash has several slots per env or in-flight claims (rolling refills);
// false = no stash or a plain one, and the step carries nothing of the deep form -- its
// probes, its request rule, its wave priority cost the plain form 0.5-1 us per launch in round 3.
// `in`, `B`, `N`: a.in / a.B / a.N, handed over separately so that a kernel whose signature leads
// with them gets them PRELOADED into SGPRs at wave launch (-amdgpu-kernarg-preload-count: the
// first loads of a wave then do not wait for an s_load of the kernarg segment).
// POL (step only): where the action comes from -- the caller (a.action) or the masked-uniform
// random policy inside the launch (a.policy_keys).  For the plain form a template parameter, not
// a test of the pointer: the kernel of the caller's own actions then carries no early read of
// that kernarg word and none of the policy's code (as a run-time branch at the head of one
// kernel it cost the path without a fused policy 0.3 us per launch).
// The deep form keeps the run-time test (kPolicyAtRunTime): beside rolling refills neither variant
// of the compile-time form did better than it (SeedExtension 18.4-18.9 against 17.3-17.4 us per
// step, SequentialRandomWalk 16.8-17.1 against 16.6 -- that generator's producers run at their
// throughput limit, and any shift in the step's instruction schedule moves the balance).
enum { kActionsFromCaller = 0, kPolicyInLaunch = 1, kPolicyAtRunTime = 2 };
template <int MODE, int KIND, bool DEEP = true, int POL = kPolicyAtRunTime>
__device__ __forceinline__ void connector_body(const EnvArgs &a, const StatePtrs &in, const int B,
                                               const int N_) {
#if defined(RBG_PROBE_STEP_TIMES)  // developer probe (tools/probe_step.py --times)
  const uint32_t probe_t0 = (uint32_t)__builtin_amdgcn_s_memrealtime();  // 100 MHz, chip-wide
#endif
  const int lane = threadIdx.x & (kWaveSize - 1);
  const SubWave<64> sw(lane);
  int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if constexpr (MODE == kModeStep && DEEP) {
#if defined(RBG_PROBE_SETPRIO_FIRST)  // developer probe: round 3's form (see below)
    if (a.stash.inflight) __builtin_amdgcn_s_setprio(1);
#elif !defined(RBG_PROBE_SETPRIO_LATE)
    // Rolling refills (refills run beside the step launches): the step's waves outrank the
    // refill's in issue arbitration, from their first instruction on.  A refill wave is the
    // oldest wave on its SIMD for a millisecond and wins every cycle it is ready in; at equal
    // priority a SeedExtension refill makes the step launches it overlaps 3-7x longer (13 ->
    // 30-105 us), at priority 1 they take 17-18 us and the refill, which has slack, takes the
    // rest.
    // Written as an asm WITHOUT side effects whose (zero) result feeds the env index: to the
    // compiler the s_setprio builtin is a call with unmodelled side effects, and a wave-uniform
    // global load that follows one is no longer provably unclobbered -- it becomes a vector load
    // + s_waitcnt + v_readfirstlane instead of an s_load.  Round 3 had the builtin first thing,
    // and State.step_count (needed for a branch) then cost every wave of the launch an exposed
    // memory round trip before its grid load was even issued (12.1 -> 12.3-13.3
    // us per launch).  The data dependence keeps the instruction ahead of everything indexed by
    // the env; harmless for a deep stash that is refilled in line.
    int zero;
    asm("s_setprio 1\n\ts_mov_b32 %0, 0" : "=s"(zero));
    wib += zero;
#endif
  }
  // (the other generators' kernels may run fewer waves per workgroup: SeedExtension's LDS area
  // at large grids -- launch_env)
  const int wpb = KIND == kGenPrw ? kEnvWavesPerBlock : (int)(blockDim.x >> 6);
  const int b = blockIdx.x * wpb + wib;
  if (b >= B) return;
  uint32_t *board = g_lds + wib * env_slab_words<64>(a.g, a.gen);
  uint32_t *W = board + board_words(a.g);
  const Geo g = a.g;
  const int N = N_;
  AgentRegs ag = {0, 0, 0, 0, 0, 0};

  if constexpr (MODE == kModeReset) {  // ---- Connector.reset ----------------------------------
    Key2 kin;
    kin.a = a.keys[2 * (size_t)b];
    kin.b = a.keys[2 * (size_t)b + 1];
    const Key2 key = regenerate_kind<64, KIND>(sw, board, W, g, N, kin, ag, a.gen);
    store_state(sw, board, g, N, b, ag, 0, key, a.out);
    request_refill<KIND>(sw, a.stash, B, b, key, a.tick + a.ready_delay);
    write_observation(sw, board, g, N, b, ag, 0, a.ts,
                      lane < N ? action_mask_bits(board, g, ag, lane) : 1u);
    write_extras(sw, board, g, N, b, ag, a.ts);
    write_restart(N, b, lane, a.ts);
    return;
  }

  // Policy fused into the step launch: the random word of (step, agent) does not depend on the
  // state, so its key is fetched first and its Threefry runs while the grid is still in flight
  // (only the pick among the legal actions needs the mask of the state just loaded).
  // (Plain stash form only: beside rolling refills a step wave that spends 70 instructions
  // before its loads go out gives the refill's waves that much more of a head start -- 17.1 /
  // 18.6 against 16.6 / 17.4 us per step for the two rolling generators; the deep form keeps the
  // evaluation behind the state load.)
  constexpr bool kPolicyFirst = MODE == kModeStep && POL == kPolicyInLaunch;
  uint32_t policy_y0 = 0u;
  int b_state = b;  // (the env index the state is loaded with)
  if constexpr (kPolicyFirst) {
    uint32_t y1;
    threefry(a.policy_keys[2 * (size_t)b], a.policy_keys[2 * (size_t)b + 1], a.policy_t,
             (uint32_t)lane, policy_y0, y1);
#if !defined(RBG_PROBE_POLICY_UNPINNED)
    // The whole evaluation stays AHEAD of the state's loads: left to itself the ILP scheduler
    // weaves the 70 Threefry instructions into the address arithmetic of those loads and the
    // grid load goes out later (14.40 against 14.05 us per step, same box).  An empty asm
    // without side effects that takes both values in and out is only a data dependence -- the
    // loads are indexed by b_state --, so it clobbers nothing (see the deep form's priority).
    asm("" : "+s"(b_state), "+v"(policy_y0));
#endif
  }
  int step_count;
  Key2 key;
  load_state(sw, board, g, N, b_state, in, ag, step_count, key);

  if constexpr (MODE == kModeObserve) {  // ---- observe an existing state (restart timestep) ---
    write_observation(sw, board, g, N, b, ag, step_count, a.ts,
                      lane < N ? action_mask_bits(board, g, ag, lane) : 1u);
    write_extras(sw, board, g, N, b, ag, a.ts);
    write_restart(N, b, lane, a.ts);
    return;
  }

  // ---- Connector.step --------------------------------------------------------------------------
  int act = 0;
  if (POL == kPolicyInLaunch || (POL == kPolicyAtRunTime && a.policy_keys)) {
    // the mask rbg_random_actions would read from the previous timestep is the mask of the state
    // just loaded
    if constexpr (!kPolicyFirst) {
      uint32_t y1;
      threefry(a.policy_keys[2 * (size_t)b], a.policy_keys[2 * (size_t)b + 1], a.policy_t,
               (uint32_t)lane, policy_y0, y1);
    }
    if (lane < N) act = policy_pick(policy_y0, action_mask_bits(board, g, ag, lane));
  } else if (lane < N) {
    act = a.action[(size_t)b * N + lane];
  }
  const StashProbe probe = stash_probe_issue<DEEP>(a.stash, B, b, lane);  // in flight during the step
  RowPrefetch pf;
  pf.row = -1;
  if (a.p.auto_reset && a.stash.tag && step_count + 1 >= a.p.time_limit && (g.m & 3) == 0 &&
      g.n <= 4 * kWaveSize)  // wave-uniform
    pf = stash_row_issue(a.stash.next, g, N,
                         stash_probe_row<DEEP>(a.stash, probe, a.B, b, lane, key, a.tick), lane);
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
                                : stash_probe_row<DEEP>(a.stash, probe, a.B, b, lane, key, a.tick);
    if (pf.row >= 0) {
      stash_row_commit(sw, board, g, pf, ag, key);
      if (a.stash.hits && lane == 0) a.stash.hits[b] += 1;
    } else if (row >= 0) {
      load_state(sw, board, g, N, row, a.stash.next, ag, step_count, key);
      if (a.stash.hits && lane == 0) a.stash.hits[b] += 1;
    } else {
      if (a.stash.misses && lane == 0) a.stash.misses[b] += 1;
      Key2 rk, unused;
      split2_lanes(sw, key, rk, unused);
      key = regenerate_kind<64, KIND>(sw, board, W, g, N, rk, ag, a.gen);
    }
    step_count = 0;
    request_refill<KIND, DEEP ? kMaxStashDepth : 1>(sw, a.stash, B, b, key,
                                                    a.tick + a.ready_delay, row, true, probe.far);
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

__global__ __launch_bounds__(kWaveSize * kEnvWavesPerBlock) void connector_reset_kernel(EnvArgs a) {
  connector_body<kModeReset, kGenPrw>(a, a.in, a.B, a.N);
}
__global__ __launch_bounds__(kWaveSize * kEnvWavesPerBlock) void connector_observe_kernel(EnvArgs a) {
  connector_body<kModeObserve, kGenPrw>(a, a.in, a.B, a.N);
}
#if defined(RBG_PROBE_PRELOAD)  // developer probe: the State pointers as leading (preloaded) arguments
__global__ __launch_bounds__(kWaveSize * kEnvWavesPerBlock) void connector_step_kernel(
    int32_t *grid, int32_t *step_count, int32_t *start, int32_t *target, int32_t *position,
    uint32_t *key, int B, int N, EnvArgs a) {
  const StatePtrs in = {grid, step_count, start, target, position, key};
  connector_body<kModeStep, kGenPrw, false, kActionsFromCaller>(a, in, B, N);
}
#else
__global__ __launch_bounds__(kWaveSize * kEnvWavesPerBlock) void connector_step_kernel(EnvArgs a) {
  connector_body<kModeStep, kGenPrw, false, kActionsFromCaller>(a, a.in, a.B, a.N);
}
#endif
__global__ __launch_bounds__(kWaveSize * kEnvWavesPerBlock) void connector_step_deep_kernel(
    EnvArgs a) {
  connector_body<kModeStep, kGenPrw, true>(a, a.in, a.B, a.N);
}
// ... with the random policy inside the launch (rbg_connector_step_gen with policy_keys; plain
// stash form -- the deep form decides at run time, connector_body)
__global__ __launch_bounds__(kWaveSize * kEnvWavesPerBlock) void connector_step_policy_kernel(
    EnvArgs a) {
  connector_body<kModeStep, kGenPrw, false, kPolicyInLaunch>(a, a.in, a.B, a.N);
}

// Connector(generator = Uniform / SequentialRandomWalk / SeedExtension / offline dataset)
template <int KIND>
__global__ __launch_bounds__(kWaveSize * kEnvWavesPerBlock) void connector_reset_gen_kernel(
    EnvArgs a) {
  connector_body<kModeReset, KIND>(a, a.in, a.B, a.N);
}
template <int KIND, bool DEEP, int POL>
__global__ __launch_bounds__(kWaveSize * kEnvWavesPerBlock) void connector_step_gen_kernel(
    EnvArgs a) {
  connector_body<kModeStep, KIND, DEEP, POL>(a, a.in, a.B, a.N);
}


// Producer of the stash: for every env whose entry is not the one its running episode will need
// (tag != refill_key), generate G(split(refill_key)[0]) and publish it.  One env per sub-wave
// (two per wave at 16x16/10); sub-waves with nothing to do retire at once.
struct RefillArgs {
  int B, N;
  Geo g;
  StashPtrs stash;
  GenSpec gen;
  int listed;  // the rows to generate are listed in stash.scratch (stash_list_kernel ran)
  // stash_list_kernel with in-flight claims (rolling refills; include/rbg_prw.h::rbg_stash_list)
  int claim_ready;  // ready_at of the requests this list serves (>= 1): the value a claim carries
  int skip_ready;   // ready_at of the requests the NEXT list will serve
  int envs;         // B (a.B counts rows = depth * B here)
  // rbg_stash_roll's miss snapshot, taken by the list launch itself (all nullable together)
  unsigned long long *miss_acc;   // work word, zero between launches
  unsigned long long *miss_word;  // device copy of the snapshot
  unsigned long long *miss_host;  // the same word in mapped host memory (NULL: the caller copies)
  uint32_t stamp;
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
  // (no fence between the row and its tag: whoever consumes the row is ordered behind this whole
  // launch by the host -- ready_at -- never by the tag alone; an agent-scope fence here is an L2
  // write-back per wavefront and took the refill of 4096 rows from 34 to 86 us)
  if (sw.l == 0) *reinterpret_cast<uint2 *>(&a.stash.tag[2 * (size_t)b]) = want;
}

// The same producer for the other generators.  Their sub-waves run in lock-step (SeedExtension's
// chain phase, one instruction stream for the boards of a wave), so a sub-wave whose row needs
// nothing does not retire: it redoes the wave's first requested board and stores nothing; a wave
// none of whose rows is requested retires at once.  blockDim.x / 64 waves per workgroup.
// Rows with a pending request (tag != refill_key), listed for a dense packing of the producer's
// wavefronts: scratch[0] = their number (zeroed by the host before), scratch[1..] = the rows, in
// no particular order (which rows share a wavefront does not matter to any result).
// With in-flight claims (stash.inflight: refills overlap each other and the step launches):
//   * the claim is read FIRST, the tag after it -- the producer publishes tag -> fence ->
//     inflight = 0 (release), so a row seen released is seen with the tag its producer wrote, and
//     nothing is listed (generated) twice;
//   * a claim carries the ready tick of the requests its list served (claim_ready).  A row that is
//     skipped because an earlier refill is still generating it, but whose request is NEWER than
//     that claim (ready_at > claim: the env re-assigned the row while its producer ran -- after a
//     miss, request_refill's general rule may do that), will only be listed by the next list, so
//     its ready_at moves on to that list's tick (skip_ready): ready_at stays truthful -- a launch
//     that passes `tick >= ready_at` is ordered, by the host's events, behind the producer that
//     wrote the row it finds (include/rbg_prw.h::rbg_stash_list).
__global__ void stash_list_kernel(RefillArgs a) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  bool need = false;
  if (r < a.B) {
    int claim = 0;
    uint2 have;
    if (a.stash.inflight) {
#if defined(RBG_PROBE_LIST_ACQUIRE)  // developer probe: the textbook form (an L2 invalidate per wave)
      claim = __hip_atomic_load(&a.stash.inflight[r], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
      have = *reinterpret_cast<const uint2 *>(&a.stash.tag[2 * (size_t)r]);
#else
      // claim, then tag, both read at agent scope (past any line this XCD's L2 may hold) with the
      // first complete before the second is issued: a released row is seen with the tag its
      // producer published before the release -- without the cache-wide invalidate an
      // agent-scope acquire costs every wave of this launch while step launches run beside it
      claim = __hip_atomic_load(&a.stash.inflight[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const unsigned long long t = __hip_atomic_load(
          reinterpret_cast<const unsigned long long *>(&a.stash.tag[2 * (size_t)r]),
          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      have = make_uint2((uint32_t)t, (uint32_t)(t >> 32));
#endif
    } else {
      have = *reinterpret_cast<const uint2 *>(&a.stash.tag[2 * (size_t)r]);
    }
    const uint2 want = *reinterpret_cast<const uint2 *>(&a.stash.refill_key[2 * (size_t)r]);
    need = want.x != have.x || want.y != have.y;
    if (need && a.stash.inflight) {
      if (claim != 0) {  // an earlier, still running refill is generating the row: leave it alone
        need = false;
        const int ready = a.stash.ready_at[r];
        if (ready > claim && ready < a.skip_ready) a.stash.ready_at[r] = a.skip_ready;
      } else {
        a.stash.inflight[r] = a.claim_ready > 0 ? a.claim_ready : 1;
      }
    }
  }
  if (a.miss_acc) {
    // rbg_stash_roll's snapshot of the stash's miss counter, on the way: every wave adds
    // (its sum << 20) + 1 to ONE word with ONE atomic -- sum and ticket in the same word, so
    // the wave that draws the last ticket holds the complete sum without any fence -- and that
    // wave publishes (sum << 20 | stamp) for the host and clears the word for the next launch.
    // (A launch of its own for this, or a device-to-host copy, sat on the generator stream in
    // front of every producer: +0.5 / +0.8 us per step for SeedExtension.)
    unsigned long long v = r < a.envs && a.stash.misses ? (unsigned long long)a.stash.misses[r] : 0ull;
    for (int off = kWaveSize / 2; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & (kWaveSize - 1)) == 0) {
      const unsigned long long waves = (unsigned long long)gridDim.x * (blockDim.x / kWaveSize);
      const unsigned long long add = (v << 20) + 1ull;
      const unsigned long long old = atomicAdd(a.miss_acc, add);
      if ((old & 0xFFFFFull) + 1ull == waves) {
        const unsigned long long w = (((old + add) >> 20) << 20) | (unsigned long long)(a.stamp & 0xFFFFFu);
        *a.miss_word = w;
        if (a.miss_host) __hip_atomic_store(a.miss_host, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        atomicExch(a.miss_acc, 0ull);
      }
    }
  }
  const uint64_t m = __ballot(need);
  if (m == 0ull) return;
  const int lane = threadIdx.x & (kWaveSize - 1);
  int base = 0;
  if (lane == 0) base = atomicAdd(&a.stash.scratch[0], __popcll(m));
  base = __builtin_amdgcn_readfirstlane(base);
  if (need) a.stash.scratch[1 + base + __popcll(m & ((1ull << lane) - 1ull))] = r;
}

template <int LPB, int KIND>
__global__ __launch_bounds__(kBlock) void stash_refill_gen_kernel(RefillArgs a) {
  constexpr int kSubs = SubWave<LPB>::kSubs;
  const SubWave<LPB> sw(threadIdx.x & (kWaveSize - 1));
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wpb = (int)(blockDim.x >> 6);
  const Geo g = a.g;
  uint32_t *board = g_lds + (wib * kSubs + sw.h) * env_slab_words<LPB>(g, a.gen);
  uint32_t *W = board + board_words(g);
  // A listed launch comes with a grid smaller than the row count (the host does not know how many
  // rows are listed, and thousands of workgroups that start only to find nothing to do cost the
  // dispatcher ~30 ns each -- 0.1-0.2 ms per refill at 8-12 slots per env, during which the step
  // launches of the other queue starve): its workgroups stride over the list.
  for (int blk = blockIdx.x;; blk += gridDim.x) {
    const int slot = (blk * wpb + wib) * kSubs + sw.h;
    uint2 want = make_uint2(0u, 0u);
    bool need = false;
    int b = slot;
    if (a.listed) {  // slot i generates the i-th listed row
      const int count = a.stash.scratch[0];
      if (blk * wpb * kSubs >= count) return;  // (uniform in the workgroup)
      need = slot < count;
      b = need ? a.stash.scratch[1 + slot] : 0;
      if (need) want = *reinterpret_cast<const uint2 *>(&a.stash.refill_key[2 * (size_t)b]);
    } else if (b < a.B) {
      want = *reinterpret_cast<const uint2 *>(&a.stash.refill_key[2 * (size_t)b]);
      const uint2 have = *reinterpret_cast<const uint2 *>(&a.stash.tag[2 * (size_t)b]);
      need = want.x != have.x || want.y != have.y;
    }
    const uint64_t busy = __ballot(need);
    if (busy != 0ull) {  // wave-uniform
      const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)busy) - 1);
      Key2 key;
      key.a = need ? want.x : (uint32_t)__builtin_amdgcn_readlane((int)want.x, src);
      key.b = need ? want.y : (uint32_t)__builtin_amdgcn_readlane((int)want.y, src);
      AgentRegs ag = {0, 0, 0, 0, 0, 0};
      Key2 rk, unused;
      split2_lanes(sw, key, rk, unused);  // VmapAutoResetWrapper: key, _ = split(state.key)
      const Key2 next_key = regenerate_kind<LPB, KIND>(sw, board, W, g, a.N, rk, ag, a.gen);
      if (need) {
        store_state(sw, board, g, a.N, b, ag, 0, next_key, a.stash.next);
#if defined(RBG_PROBE_FENCE_BEFORE_TAG)  // developer probe: what a fence between row and tag costs
        __threadfence();
#endif
        if (sw.l == 0) {
          *reinterpret_cast<uint2 *>(&a.stash.tag[2 * (size_t)b]) = want;
          if (a.listed && a.stash.inflight)  // release: whoever sees the row released sees its tag
            __hip_atomic_store(&a.stash.inflight[b], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if (!a.listed) return;  // a scan: one slot per sub-wave
    wave_lds_fence();
  }
}

}  // namespace
