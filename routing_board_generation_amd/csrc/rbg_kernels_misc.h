// rbg_kernels_misc.h -- small kernels: offline-dataset sampling, jax.random.split, the masked-
// uniform random policy, and the walk taken apart (rbg_pieces.h).
#pragma once
#include "rbg_kernels_common.h"
#include "rbg_pieces.h"

namespace {

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

}  // namespace
