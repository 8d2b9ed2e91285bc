// rbg_kernels_rollout.h -- the fused random-policy rollout (env resident in LDS across steps),
// plain and recording.
#pragma once
#include "rbg_kernels_stash.h"
#include "rbg_regen.h"

namespace {

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
  GenSpec gen;           // the env's generator (rbg_regen.h); all zero = ParallelRandomWalk
  int block_threads;     // host side only: workgroup size of the launch (other generators' kernels)
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
// slots per env a rollout launch handles (two is what a many-step launch needs; the request rule
// is unrolled over them, and the full-wave rollout kernel has no registers to spare)
#ifndef RBG_ROLLOUT_STASH_DEPTH
#define RBG_ROLLOUT_STASH_DEPTH 2
#endif
constexpr int kRolloutStashDepth = RBG_ROLLOUT_STASH_DEPTH;

template <int LPB, bool kRecord, int KIND>
__device__ __forceinline__ void rollout_body(const RolloutArgs &a) {
  constexpr int kSubs = SubWave<LPB>::kSubs;
  const SubWave<LPB> sw(threadIdx.x & (kWaveSize - 1));
  const int l = sw.l;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wpb = KIND == kGenPrw ? kWavesPerBlock : (int)(blockDim.x >> 6);
  const int b = (blockIdx.x * wpb + wib) * kSubs + sw.h;
  if (b >= a.B) return;  // per sub-wave: the rest of the wave carries on
  const Geo g = a.g;
  uint32_t *board = g_lds + (wib * kSubs + sw.h) * env_slab_words<LPB>(g, a.gen);
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
          if (a.stash.misses && l == 0) a.stash.misses[b] += 1;
          Key2 rk, unused;
          split2_lanes(sw, key, rk, unused);
          key = regenerate_kind<LPB, KIND>(sw, board, W, g, N, rk, ag, a.gen);
        }
        step_count = 0;
        key = uni<LPB>(key);
        request_refill<KIND, kRolloutStashDepth>(sw, a.stash, a.B, b, key, a.tick + a.ready_delay,
                                                 row);
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

template <int LPB, bool kRecord>
__global__ __launch_bounds__(kBlock, (LPB == kWaveSize ? (kRecord ? 3 : RBG_ROLLOUT64_WAVES) : 2)) void connector_rollout_kernel(
    RolloutArgs a) {
  rollout_body<LPB, kRecord, kGenPrw>(a);
}

// The rollout for the other generators of setup_train.py:112-152: one env per wavefront (the
// SequentialRandomWalk needs the whole wave; the others keep one code shape), two waves per SIMD.
template <int KIND, bool kRecord>
__global__ __launch_bounds__(kBlock, 2) void connector_rollout_gen_kernel(RolloutArgs a) {
  rollout_body<kWaveSize, kRecord, KIND>(a);
}

// The recording rollout's launch.
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

}  // namespace
