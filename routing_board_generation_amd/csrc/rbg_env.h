// rbg_env.h -- Connector environment pieces on an LDS-resident env, one env per sub-wave of LPB
// lanes (rbg_device.h::SubWave; sub-lane i < N = agent i): state load/store, action mask,
// observation, extras, one step, regeneration.  Everything that crosses lanes stays inside the
// sub-wave, so the envs of a wave may diverge (one finishing an episode while the other steps on).
// Semantics: [3P] jumanji==0.2.2 jumanji/environments/routing/connector/{env,utils,reward}.py and
// jumanji/wrappers.py (see DESIGN.md section 2 for what is pinned and what is project-defined).
#pragma once
#include "../../include/rbg_prw.h"
#include "rbg_device.h"
#include "rbg_walk.h"

namespace rbg {

// HBM int32 grid -> padded LDS board.  The first pass of global loads is issued before the wall
// fill, so that their latency runs under it.
template <int LPB>
__device__ __forceinline__ void load_grid(const SubWave<LPB> &sw, uint32_t *board, const Geo &g,
                                          const int32_t *__restrict__ in) {
  const int lane = sw.l;
  if ((g.m & 3) == 0) {
    const int idx0 = lane * 4;
    int4 v0 = make_int4(0, 0, 0, 0);
    if (idx0 < g.n) v0 = *reinterpret_cast<const int4 *>(&in[idx0]);
    for (int w = lane; w < g.words; w += LPB) board[w] = kWall;
    wave_lds_fence();
    if (idx0 < g.n) {
      const int r = div_m(g, idx0);
      *reinterpret_cast<uint4 *>(&board[pad_index(g, r, idx0 - r * g.m)]) =
          make_uint4((uint32_t)v0.x, (uint32_t)v0.y, (uint32_t)v0.z, (uint32_t)v0.w);
    }
    for (int idx = idx0 + LPB * 4; idx < g.n; idx += LPB * 4) {
      const int r = div_m(g, idx);
      const int4 v = *reinterpret_cast<const int4 *>(&in[idx]);
      *reinterpret_cast<uint4 *>(&board[pad_index(g, r, idx - r * g.m)]) =
          make_uint4((uint32_t)v.x, (uint32_t)v.y, (uint32_t)v.z, (uint32_t)v.w);
    }
  } else {
    for (int w = lane; w < g.words; w += LPB) board[w] = kWall;
    wave_lds_fence();
    for (int idx = lane; idx < g.n; idx += LPB) {
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

template <int LPB>
__device__ __forceinline__ void store_obs(const SubWave<LPB> &sw, const uint32_t *board,
                                          const Geo &g, int N, int32_t *__restrict__ obs) {
  const int n3 = 3 * N, lane = sw.l;
  if ((g.m & 3) == 0) {
    // Four cells per lane, all N planes from registers.  Values fit 16 bits (<= 3N <= 192), so the
    // arithmetic runs on two packed pairs: u = (v - 1 - 3i) mod 3N is carried from plane to plane
    // (u - 3, wrapped by an unsigned min against u - 3 + 3N), the view is (u + 1) where the cell is
    // not empty.  14 VALU instructions per plane and lane instead of 22.
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    typedef int v4i __attribute__((ext_vector_type(4)));
    const u16x2 one = {1, 1}, three = {3, 3};
    const u16x2 wrap = {(unsigned short)n3, (unsigned short)n3};
    for (int idx = lane * 4; idx < g.n; idx += LPB * 4) {
      const int r = div_m(g, idx);
      const uint4 v = *reinterpret_cast<const uint4 *>(&board[pad_index(g, r, idx - r * g.m)]);
      const u16x2 p01 = __builtin_bit_cast(u16x2, v.x | (v.y << 16));
      const u16x2 p23 = __builtin_bit_cast(u16x2, v.z | (v.w << 16));
      // 32-bit masks of the cells that hold a wire (an empty cell stays 0 in every view)
      const uint32_t mx = v.x ? 0x0000FFFFu : 0u, my = v.y ? 0xFFFF0000u : 0u;
      const uint32_t mz = v.z ? 0x0000FFFFu : 0u, mw = v.w ? 0xFFFF0000u : 0u;
      u16x2 u01 = p01 - one, u23 = p23 - one;
      int32_t *out = obs + idx;
      for (int i = 0; i < N; ++i) {
        const uint32_t s01 = __builtin_bit_cast(uint32_t, (u16x2)(u01 + one));
        const uint32_t s23 = __builtin_bit_cast(uint32_t, (u16x2)(u23 + one));
        // write-once stream: non-temporal keeps it from evicting the states in L2/Infinity Cache
        const v4i ov = {(int)(s01 & mx), (int)((s01 & my) >> 16), (int)(s23 & mz),
                        (int)((s23 & mw) >> 16)};
#if defined(RBG_PROBE_PLAIN_OBS_STORES)  // developer probe (bench.py --mode stream on a variant build)
        *reinterpret_cast<v4i *>(out) = ov;
#else
        __builtin_nontemporal_store(ov, reinterpret_cast<v4i *>(out));
#endif
        out += g.n;
        const u16x2 t01 = u01 - three, t23 = u23 - three;
        u01 = __builtin_elementwise_min(t01, (u16x2)(t01 + wrap));
        u23 = __builtin_elementwise_min(t23, (u16x2)(t23 + wrap));
      }
    }
  } else {
    for (int idx = lane; idx < g.n; idx += LPB) {
      const int r = div_m(g, idx);
      const uint32_t v = board[pad_index(g, r, idx - r * g.m)];
      for (int i = 0; i < N; ++i) obs[(size_t)i * g.n + idx] = obs_value(v, 3 * i, n3);
    }
  }
}

// number of PATH cells (v % 3 == 1), the `total_path_length` extra
template <int LPB>
__device__ __forceinline__ int count_path_cells(const SubWave<LPB> &sw, const uint32_t *board,
                                                const Geo &g) {
  // one sub-wave ballot + popcount per pass instead of a lane-to-lane reduction (six dependent
  // ds_bpermute round trips); the trip count is uniform, lanes past the end contribute 0
  int cnt = 0;
  for (int base = 0; base < g.n; base += LPB) {
    const int idx = base + sw.l;
    const int r = div_m(g, idx);
    const bool path = idx < g.n && board[pad_index(g, r, idx - r * g.m)] % 3u == 1u;
    cnt += __popcll(sw.ballot(path));
  }
  return cnt;
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
  float *reward_total;
  float *discount_max;
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
template <int LPB>
__device__ __forceinline__ void board_to_training(const SubWave<LPB> &sw, uint32_t *board,
                                                  const Geo &g) {
  for (int idx = sw.l; idx < g.n; idx += LPB) {
    const int r = div_m(g, idx);
    const int p = pad_index(g, r, idx - r * g.m);
    const uint32_t v = board[p];
    if (v % 3u == 1u) board[p] = 0u;
  }
  wave_lds_fence();
}

// Observation + extras of the state held in (board, ag); `terminal` keeps reward/discount/
// step_type/extras of the step that just ended when the env was auto-reset.
template <int LPB>
__device__ __forceinline__ void write_observation(const SubWave<LPB> &sw, const uint32_t *board,
                                                  const Geo &g, int N, int b, const AgentRegs &ag,
                                                  int step_count, const TimestepPtrs &ts,
                                                  uint32_t bits) {  // action mask of (board, ag)
  const int lane = sw.l;
  store_mask(ts, b, N, lane, bits);
  if (ts.obs_grid) store_obs(sw, board, g, N, ts.obs_grid + (size_t)b * N * g.n);
  if (ts.obs_step_count && lane == 0) ts.obs_step_count[b] = step_count;
}

template <int LPB>
__device__ __forceinline__ void write_extras(const SubWave<LPB> &sw, const uint32_t *board,
                                             const Geo &g, int N, int b, const AgentRegs &ag,
                                             const TimestepPtrs &ts) {
  const int lane = sw.l;
  const bool connected = lane < N && ag.r == ag.tr && ag.c == ag.tc;
  const int conn = __popcll(sw.ballot(connected));
  const int plen = ts.total_path_length ? count_path_cells(sw, board, g) : 0;
  if (lane == 0) {
    if (ts.num_connections) ts.num_connections[b] = conn;
    if (ts.ratio_connections) ts.ratio_connections[b] = __fdiv_rn((float)conn, (float)N);
    if (ts.total_path_length) ts.total_path_length[b] = plen;
  }
}

// ---------------------------------------------------------------------------------------------
// One Connector.step on the LDS-resident env ([3P] env.py::step / _step_agents / _step_agent).
// ---------------------------------------------------------------------------------------------
struct StepResult {
  bool was_connected, connected, done;  // per agent lane
  bool last;                            // per env (uniform in the sub-wave): all done or time limit
  uint32_t bits;                        // action mask of the stepped state
};

// _step_agent: move iff the action is not NOOP, the destination is in bounds and empty or the
// agent's own target, and the agent is not connected yet.  Only an EMPTY cell can be contested
// (a target cell admits its owner alone): max-join through one ds_max_u32, losers stay.
// `fill` is independent register-only work of the caller (the rollout's Threefry for the NEXT
// step) that is issued into the two LDS round trips of the move: fill.first() after the read of
// the destination cell, fill.second() after the ds_max + read-back.  A lone env step is a chain of
// dependent instructions and LDS latencies; with two waves per SIMD little else hides them.
struct NoFill {
  __device__ __forceinline__ void first() {}
  __device__ __forceinline__ void second() {}
};

template <int LPB, class Fill>
__device__ __forceinline__ StepResult env_step_core(const SubWave<LPB> &sw, uint32_t *board,
                                                    const Geo &g, int N, AgentRegs &ag,
                                                    int &step_count, int act, int time_limit,
                                                    Fill &fill) {
  const int lane = sw.l;
  const bool is_agent = lane < N;
  const uint32_t head = 3u * (uint32_t)lane + 2u;
  StepResult res;
  res.was_connected = is_agent && ag.r == ag.tr && ag.c == ag.tc;
  act = act < 0 ? 0 : act > 4 ? 4 : act;  // lax.switch clamps its index (move_position)
  const int dr = act == 1 ? -1 : act == 3 ? 1 : 0;
  const int dc = act == 2 ? 1 : act == 4 ? -1 : 0;
  const int P = pad_index(g, ag.r, ag.c);
  const int Q = P + dr * g.S + dc;
  const uint32_t vq = board[Q];
  __builtin_amdgcn_sched_barrier(0);
  fill.first();
  __builtin_amdgcn_sched_barrier(0);
  const bool to_empty = vq == 0u;
  const bool to_target = vq == head + 1u;
  const bool mover = is_agent && act != 0 && !res.was_connected && (to_empty || to_target);
  if (mover && to_empty) atomicMax(&board[Q], head);
  wave_lds_fence();
  const uint32_t back = board[Q];  // (Q is a valid padded index for every lane)
  __builtin_amdgcn_sched_barrier(0);
  fill.second();
  __builtin_amdgcn_sched_barrier(0);
  if (mover && (to_target || back == head)) {
    board[Q] = head;  // overwrites the own target on arrival (move_agent)
    board[P] = head - 1u;
    ag.r += dr;
    ag.c += dc;
  }
  wave_lds_fence();
  step_count += 1;
  res.connected = is_agent && ag.r == ag.tr && ag.c == ag.tc;
  res.bits = is_agent ? action_mask_bits(board, g, ag, lane) : 1u;
  res.done = res.connected || (res.bits >> 1) == 0u;  // connected_or_blocked
  const bool all_done = sw.ballot(is_agent && !res.done) == 0ull;
  res.last = all_done || step_count >= time_limit;
  return res;
}

template <int LPB>
__device__ __forceinline__ StepResult env_step_core(const SubWave<LPB> &sw, uint32_t *board,
                                                    const Geo &g, int N, AgentRegs &ag,
                                                    int &step_count, int act, int time_limit) {
  NoFill none;
  return env_step_core(sw, board, g, N, ag, step_count, act, time_limit, none);
}

// DenseRewardFn
__device__ __forceinline__ float dense_reward(const StepResult &r, const rbg_env_params &p) {
  return __fadd_rn(p.connected_reward * (float)(!r.was_connected && r.connected),
                   p.timestep_reward * (float)(!r.was_connected));
}

// Generator -> State in LDS/registers only: runs the walk for `key_in` (the generator's
// argument), leaves the training grid in LDS, fills the agent registers, returns State.key.
template <int LPB>
__device__ __forceinline__ Key2 regenerate(const SubWave<LPB> &sw, uint32_t *board, uint32_t *W,
                                           const Geo &g, int N, Key2 key_in, AgentRegs &ag) {
  Key2 key, unused;
  split2_lanes(sw, key_in, key, unused);
  Walker w;
  prw_walk(sw, board, W, g, N, key, w);
  board_to_training(sw, board, g);
  ag.r0 = w.r0; ag.c0 = w.c0;
  ag.tr = w.r;  ag.tc = w.c;
  ag.r = w.r0;  ag.c = w.c0;
  return key;
}

// State (LDS board + registers) -> HBM
template <int LPB>
__device__ __forceinline__ void store_state(const SubWave<LPB> &sw, const uint32_t *board,
                                            const Geo &g, int N, int b, const AgentRegs &ag,
                                            int step_count, Key2 key, const StatePtrs &out) {
  const int lane = sw.l;
  store_grid<LPB, false>(board, g, lane, out.grid + (size_t)b * g.n);
  store_agents(out, b, N, lane, ag, true);
  if (lane == 0) {
    out.step_count[b] = step_count;
    *reinterpret_cast<uint2 *>(&out.key[2 * (size_t)b]) = make_uint2(key.a, key.b);
  }
}

// HBM -> State (LDS board + registers)
template <int LPB>
__device__ __forceinline__ void load_state(const SubWave<LPB> &sw, uint32_t *board, const Geo &g,
                                           int N, int b, const StatePtrs &in, AgentRegs &ag,
                                           int &step_count, Key2 &key) {
  const int lane = sw.l;
  // the small loads first: they are in flight while the grid is fetched and the walls are filled
  if (lane < N) {
    const size_t o = ((size_t)b * N + lane) * 2;
    const int2 s = *reinterpret_cast<const int2 *>(&in.start[o]);
    const int2 t = *reinterpret_cast<const int2 *>(&in.target[o]);
    const int2 p = *reinterpret_cast<const int2 *>(&in.position[o]);
    ag.r0 = s.x; ag.c0 = s.y; ag.tr = t.x; ag.tc = t.y; ag.r = p.x; ag.c = p.y;
  }
  step_count = in.step_count[b];
  const uint2 k = *reinterpret_cast<const uint2 *>(&in.key[2 * (size_t)b]);
  key.a = k.x;
  key.b = k.y;
  load_grid(sw, board, g, in.grid + (size_t)b * g.n);
}

// reward/discount/step_type of a restart timestep
__device__ __forceinline__ void write_restart(int N, int b, int lane, const TimestepPtrs &ts) {
  if (lane < N) {
    if (ts.reward) ts.reward[(size_t)b * N + lane] = 0.0f;
    if (ts.discount) ts.discount[(size_t)b * N + lane] = 1.0f;
  }
  if (lane == 0) {
    if (ts.step_type) ts.step_type[b] = 0;
    if (ts.reward_total) ts.reward_total[b] = 0.0f;
    if (ts.discount_max) ts.discount_max[b] = 1.0f;
  }
}

// reward/discount/step_type of a stepped timestep
template <int LPB>
__device__ __forceinline__ void write_step_outcome(const SubWave<LPB> &sw, const StepResult &res,
                                                   const rbg_env_params &p, int N, int b,
                                                   const TimestepPtrs &ts) {
  const int lane = sw.l;
  const float reward = dense_reward(res, p);
  const bool alive = !(res.last || res.done);
  if (lane < N) {
    if (ts.reward) ts.reward[(size_t)b * N + lane] = reward;
    if (ts.discount) ts.discount[(size_t)b * N + lane] = alive ? 1.0f : 0.0f;
  }
  if (ts.reward_total) {  // MultiToSingleWrapper: float32 sum in agent order (uniform branch)
    float total = 0.0f;
    for (int i = 0; i < N; ++i)
      total = __fadd_rn(total, __uint_as_float(sw.bcast(__float_as_uint(reward), i)));
    if (lane == 0) ts.reward_total[b] = total;
  }
  if (ts.discount_max) {
    const bool any_alive = sw.ballot(lane < N && alive) != 0ull;
    if (lane == 0) ts.discount_max[b] = any_alive ? 1.0f : 0.0f;
  }
  if (lane == 0 && ts.step_type) ts.step_type[b] = res.last ? 2 : 1;
}

// masked-uniform random policy (defined by this project; include/rbg_prw.h): agent i of an env
// with policy key k picks, at global step t, legal action number
//   floor(threefry2x32(k, (t, i)).y0 * #legal / 2^32)   (legal actions in ascending order).
__device__ __forceinline__ int policy_pick(uint32_t y0, uint32_t mask_bits) {
  const int cnt = __popc(mask_bits & 31u);
  const int pick = (int)__umulhi(y0, (uint32_t)cnt);
  int chosen = 0, seen = 0;
#pragma unroll
  for (int a = 0; a < 5; ++a) {
    if ((mask_bits >> a) & 1u) {
      if (seen == pick) chosen = a;
      ++seen;
    }
  }
  return chosen;
}

__device__ __forceinline__ int random_policy_action(Key2 k, uint32_t t, uint32_t i,
                                                    uint32_t mask_bits) {
  uint32_t y0, y1;
  threefry(k.a, k.b, t, i, y0, y1);
  return policy_pick(y0, mask_bits);
}

// The policy's random word of one (step, agent), computed in the shadow of the previous step.
struct PolicyFill {
  Key2 k;
  uint32_t t, i;  // counter of the word being produced
  TfState st;
  uint32_t y0;
  __device__ __forceinline__ void first() {
    st = threefry_first_half(k.a, k.b, t, i);
    asm volatile("" : "+v"(st.x0), "+v"(st.x1));  // keep the rounds where they are issued
  }
  __device__ __forceinline__ void second() {
    uint32_t y1;
    threefry_second_half(st, y0, y1);
  }
};

}  // namespace rbg
