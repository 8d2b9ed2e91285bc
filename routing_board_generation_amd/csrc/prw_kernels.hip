// prw_kernels.hip -- hand-written gfx950 (CDNA4) kernels + the C-ABI of include/rbg_prw.h.
//
// Device code lives in the headers next to this file (kernels in rbg_kernels_*.h):
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
#include "rbg_kernels_common.h"
#include "rbg_kernels_generators.h"
#include "rbg_kernels_stash.h"
#include "rbg_kernels_env.h"
#include "rbg_kernels_rollout.h"
#include "rbg_kernels_misc.h"

namespace {

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

// rbg_generator -> the device-side description of the env's generator (rbg_regen.h) + the count of
// its start-cell shuffle prefix (make_geo's threshold).  NULL = ParallelRandomWalk.
int fill_gen(const rbg_generator *gen, int m, int N, GenSpec *gs, int *prefix_count,
             const char *who) {
  memset(gs, 0, sizeof(*gs));
  *prefix_count = N;
  if (!gen || gen->kind == RBG_GEN_PARALLEL_RW) return 0;
  gs->kind = gen->kind;
  const int words = (m + 4) * (m + 4);
  switch (gen->kind) {
    case RBG_GEN_UNIFORM:
      if (2 * N > m * m) return fail(RBG_E_SHAPE, "uniform generator: 2N exceeds m*m cells");
      *prefix_count = 2 * N;
      return 0;
    case RBG_GEN_SEQUENTIAL_RW:
      if (m < 3) return fail(RBG_E_SHAPE, "sequential random walk: m must be >= 3");
      return 0;
    case RBG_GEN_PARALLEL_RW_PHILOX:
      return 0;
    case RBG_GEN_SEED_EXTENSION: {
      if (N > (m / 2) * (m / 2))
        return fail(RBG_E_SHAPE, "seed extension: N exceeds the (m/2)^2 seed cells");
      if (gen->extension_iterations < 0 || gen->max_extension_steps < 0)
        return fail(RBG_E_PARAM, "seed extension: bad iteration / step count");
      SeParams &p = gs->se;
      p.N = N;
      p.iterations = gen->extension_iterations;
      p.seed_only = 0;
      p.max_steps = gen->max_extension_steps;
      p.two_sided = gen->two_sided != 0;
      p.randomness = gen->randomness;
      p.rnd_mode = !(gen->randomness > 0.0f) ? 0 : (gen->randomness >= 1.0f ? 1 : 2);
      p.bfs_last = 0;  // a State never shows the last re-routing
      p.half = m / 2;
      gs->extra_words = se_extra_words(words, m * m, /*bfs_possible=*/p.iterations > 1);
      p.extra_words = gs->extra_words;
      return 0;
    }
    case RBG_GEN_DATASET: {
      if (gen->n_boards < 1) return fail(RBG_E_SHAPE, "dataset generator: n_boards < 1");
      if (!gen->heads || !gen->targets) return fail(RBG_E_NULL, "dataset generator: NULL boards");
      gs->ds_heads = gen->heads;
      gs->ds_targets = gen->targets;
      gs->ds_span = (uint32_t)gen->n_boards;
      const uint32_t m16 = 65536u % gs->ds_span;
      gs->ds_mult = (uint32_t)(m16 * m16) % gs->ds_span;  // wrapping uint32, as jax's _randint
      return 0;
    }
    default:
      snprintf(g_err, sizeof(g_err), "%s: unknown generator kind %d", who, gen->kind);
      return RBG_E_PARAM;
  }
}

// Which instantiation of the step a stash needs: several slots per env or in-flight claims
// (refills beside the steps) take the deep form; no stash or a plain one -- one slot per env,
// refilled in line: far keys are never read at depth 1 -- takes the form that carries none of it.
bool deep_stash(const StashPtrs &s) {
#if defined(RBG_PROBE_DEEP_ALWAYS)  // developer probe (tools/ab_step.py --build): what the deep form costs a plain stash
  return true;
#endif
  return s.tag && (s.depth > 1 || s.inflight);
}

template <int KIND>
void launch_env_gen(const EnvArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
  if (a.mode == kModeReset) {
    hipLaunchKernelGGL(connector_reset_gen_kernel<KIND>, grid, block, lds, st, a);
  } else if (deep_stash(a.stash)) {
    hipLaunchKernelGGL((connector_step_gen_kernel<KIND, true, kPolicyAtRunTime>), grid, block, lds,
                       st, a);
  } else if (a.policy_keys) {
    hipLaunchKernelGGL((connector_step_gen_kernel<KIND, false, kPolicyInLaunch>), grid, block, lds,
                       st, a);
  } else {
    hipLaunchKernelGGL((connector_step_gen_kernel<KIND, false, kActionsFromCaller>), grid, block,
                       lds, st, a);
  }
}

template <int KIND>
void launch_rollout_gen(const RolloutArgs &a, bool record, dim3 grid, size_t lds,
                               hipStream_t st) {
  const dim3 block(a.block_threads);
  if (record) {
    hipLaunchKernelGGL((connector_rollout_gen_kernel<KIND, true>), grid, block, lds, st, a);
  } else {
    hipLaunchKernelGGL((connector_rollout_gen_kernel<KIND, false>), grid, block, lds, st, a);
  }
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
  p.extra_words = se_extra_words(a.g.words, a.g.n, p.bfs_last || p.iterations > 1);
  // lanes per board: lane w owns wire w; fewer, fuller waves as long as the workgroup's LDS fits
  int lpb = 16;
  while (lpb < N) lpb *= 2;
  int wpb = kWavesPerBlock;
  size_t lds = 0;
  for (;;) {
    const size_t per_board =
        (size_t)((a.g.words + 3) & ~3) + (size_t)(4 * lpb) + (size_t)p.extra_words;
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
  // waves (envs) per workgroup: 4, fewer when a generator's own LDS area (SeedExtension at large
  // grids) would push the workgroup beyond 64 KiB
  const size_t per_env = lds_bytes(a.g) / kWavesPerBlock + (size_t)a.gen.extra_words * sizeof(uint32_t);
  int wpb = kEnvWavesPerBlock;
  while (wpb > 1 && a.gen.kind != kGenPrw && (size_t)wpb * per_env > 64u * 1024u) wpb /= 2;
  if ((size_t)wpb * per_env > 64u * 1024u) return fail(RBG_E_SHAPE, "env launch: board too large");
  const dim3 grid((a.B + wpb - 1) / wpb), block(kWaveSize * wpb);
  const size_t lds = (size_t)wpb * per_env;
  hipStream_t st = (hipStream_t)stream;
  if (a.mode == kModeObserve) {
    hipLaunchKernelGGL(connector_observe_kernel, grid, block, lds, st, a);
  } else if (a.gen.kind == kGenUniform) {
    launch_env_gen<kGenUniform>(a, grid, block, lds, st);
  } else if (a.gen.kind == kGenSeqRw) {
    launch_env_gen<kGenSeqRw>(a, grid, block, lds, st);
  } else if (a.gen.kind == kGenSeedExt) {
    launch_env_gen<kGenSeedExt>(a, grid, block, lds, st);
  } else if (a.gen.kind == kGenDataset) {
    launch_env_gen<kGenDataset>(a, grid, block, lds, st);
  } else if (a.gen.kind == kGenPhilox) {
    launch_env_gen<kGenPhilox>(a, grid, block, lds, st);
  } else if (a.mode == kModeReset) {
    hipLaunchKernelGGL(connector_reset_kernel, grid, block, lds, st, a);
  } else if (deep_stash(a.stash)) {
    hipLaunchKernelGGL(connector_step_deep_kernel, grid, block, lds, st, a);
  } else if (a.policy_keys) {
    hipLaunchKernelGGL(connector_step_policy_kernel, grid, block, lds, st, a);
  } else {
#if defined(RBG_PROBE_PRELOAD)
    hipLaunchKernelGGL(connector_step_kernel, grid, block, lds, st, a.in.grid, a.in.step_count,
                       a.in.start, a.in.target, a.in.position, a.in.key, a.B, a.N, a);
#else
    hipLaunchKernelGGL(connector_step_kernel, grid, block, lds, st, a);
#endif
  }
  return check_hip(hipGetLastError(), what);
}

int rbg_connector_reset(const uint32_t *keys, int B, int m, int N, const rbg_state *state,
                        const rbg_timestep *timestep, void *stream) {
  return rbg_connector_reset_gen(nullptr, keys, B, m, N, state, timestep, nullptr, 0, 0, stream);
}

// Launch ticks are int32 in the ABI and ready_at = tick + ready_delay must not wrap: a host whose
// session outlives 2^31 launches rebases its ticks (and the stash's ready_at) long before
// (include/rbg_prw.h::rbg_stash; session.EpochStash does at 2^30).
static int check_tick(int tick, int ready_delay, const char *who) {
  if (tick < 0 || ready_delay < 0 || tick > INT32_MAX - ready_delay) {
    snprintf(g_err, sizeof(g_err), "%s: tick / ready_delay out of range (rebase the ticks)", who);
    return RBG_E_PARAM;
  }
  return 0;
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
  out->misses = stash->misses;
  out->scratch = stash->scratch;
  out->inflight = stash->inflight;
  out->far_key = stash->far_key;
  out->depth = stash->depth;
  if (stash->depth < 1 || stash->depth > kMaxStashDepth) {
    snprintf(g_err, sizeof(g_err), "%s: stash depth must be in [1, %d]", who, kMaxStashDepth);
    return RBG_E_PARAM;
  }
  return 0;
}

int rbg_connector_reset_prefetch(const uint32_t *keys, int B, int m, int N, const rbg_state *state,
                                 const rbg_timestep *timestep, const rbg_stash *stash, int tick,
                                 int ready_delay, void *stream) {
  return rbg_connector_reset_gen(nullptr, keys, B, m, N, state, timestep, stash, tick, ready_delay,
                                 stream);
}

int rbg_connector_reset_gen(const rbg_generator *generator, const uint32_t *keys, int B, int m,
                            int N, const rbg_state *state, const rbg_timestep *timestep,
                            const rbg_stash *stash, int tick, int ready_delay, void *stream) {
  if (B < 0) return fail(RBG_E_SHAPE, "rbg_connector_reset: B < 0");
  EnvArgs a;
  memset(&a, 0, sizeof(a));
  if (int rc = make_geo(m, N, &a.g)) return rc;
  int prefix = N;
  if (int rc = fill_gen(generator, m, N, &a.gen, &prefix, "rbg_connector_reset")) return rc;
  if (int rc = make_geo(m, N, &a.g, prefix)) return rc;
  if (B == 0) return 0;
  if (!keys || !timestep) return fail(RBG_E_NULL, "rbg_connector_reset: NULL pointer");
  if (!state_complete(state)) return fail(RBG_E_NULL, "rbg_connector_reset: incomplete state");
  if (int rc = check_align(a.g, state->grid, timestep->obs_grid, stash ? stash->next.grid : nullptr))
    return rc;
  if (int rc = fill_stash(stash, &a.stash, "rbg_connector_reset")) return rc;
  if (stash)
    if (int rc = check_tick(tick, ready_delay, "rbg_connector_reset")) return rc;
  a.keys = keys; a.B = B; a.N = N; a.out = to_ptrs(state); a.in = a.out;
  a.ts = to_ptrs(timestep); a.mode = kModeReset; a.tick = tick; a.ready_delay = ready_delay;
  return launch_env(a, stream, "rbg_connector_reset");
}

int rbg_connector_step_prefetch(int B, int m, int N, const rbg_state *state_in,
                                const rbg_state *state_out, const int32_t *action,
                                const rbg_env_params *params, const rbg_timestep *timestep,
                                const rbg_stash *stash, int tick, int ready_delay, void *stream) {
  return rbg_connector_step_gen(nullptr, B, m, N, state_in, state_out, action, nullptr, 0u, params,
                                timestep, stash, tick, ready_delay, stream);
}

int rbg_connector_step_call(const rbg_step_call *c) {
  if (!c) return fail(RBG_E_NULL, "rbg_connector_step_call: NULL call");
  return rbg_connector_step_gen(c->generator, c->B, c->m, c->N, &c->state_in, &c->state_out,
                                c->action, c->policy_keys, c->policy_t, &c->params, &c->timestep,
                                c->stash, c->tick, c->ready_delay, c->stream);
}

int rbg_connector_step_gen(const rbg_generator *generator, int B, int m, int N,
                           const rbg_state *state_in, const rbg_state *state_out,
                           const int32_t *action, const uint32_t *policy_keys, uint32_t policy_t,
                           const rbg_env_params *params, const rbg_timestep *timestep,
                           const rbg_stash *stash, int tick, int ready_delay, void *stream) {
  if (B < 0) return fail(RBG_E_SHAPE, "rbg_connector_step: B < 0");
  if (!params) return fail(RBG_E_NULL, "rbg_connector_step: NULL params");
  if (params->time_limit < 1) return fail(RBG_E_PARAM, "rbg_connector_step: time_limit < 1");
  EnvArgs a;
  memset(&a, 0, sizeof(a));
  if (int rc = make_geo(m, N, &a.g)) return rc;
  int prefix = N;
  if (int rc = fill_gen(generator, m, N, &a.gen, &prefix, "rbg_connector_step")) return rc;
  if (int rc = make_geo(m, N, &a.g, prefix)) return rc;
  if (B == 0) return 0;
  if ((!action && !policy_keys) || !timestep)
    return fail(RBG_E_NULL, "rbg_connector_step: NULL pointer");
  if (!state_complete(state_in) || !state_complete(state_out))
    return fail(RBG_E_NULL, "rbg_connector_step: incomplete state");
  if (int rc = check_align(a.g, state_in->grid, state_out->grid, timestep->obs_grid)) return rc;
  if (stash)
    if (int rc = check_align(a.g, stash->next.grid, nullptr, nullptr)) return rc;
  if (int rc = fill_stash(stash, &a.stash, "rbg_connector_step")) return rc;
  if (stash)
    if (int rc = check_tick(tick, ready_delay, "rbg_connector_step")) return rc;
  a.B = B; a.N = N; a.in = to_ptrs(state_in); a.out = to_ptrs(state_out); a.action = action;
  a.policy_keys = policy_keys; a.policy_t = policy_t;
  a.p = *params; a.ts = to_ptrs(timestep); a.mode = kModeStep; a.tick = tick;
  a.ready_delay = ready_delay;
  return launch_env(a, stream, "rbg_connector_step");
}

int rbg_stash_refill(int B, int m, int N, const rbg_stash *stash, void *stream) {
  return rbg_stash_refill_gen(nullptr, B, m, N, stash, stream);
}

// mode 0: scan every row (listing SeedExtension's first when a scratch area is given: dense
// packing); 1: list only; 2: generate the listed rows
static int stash_refill_impl(const rbg_generator *generator, int B, int m, int N,
                             const rbg_stash *stash, void *stream, int mode, int claim_ready = 1,
                             int skip_ready = 0, const rbg_stash_roll_call *snap = nullptr,
                             void *snap_host = nullptr) {
  if (B < 0) return fail(RBG_E_SHAPE, "rbg_stash_refill: B < 0");
  RefillArgs a;
  memset(&a, 0, sizeof(a));
  if (int rc = make_geo(m, N, &a.g)) return rc;
  int prefix = N;
  if (int rc = fill_gen(generator, m, N, &a.gen, &prefix, "rbg_stash_refill")) return rc;
  const int kind = a.gen.kind;
  // lanes per board: the walk's pipeline width; one lane per wire for SeedExtension; a whole wave
  // for the others
  int lpb = kWaveSize, wpb = kWavesPerBlock;
  if (kind == kGenPrw) {
    lpb = pick_lpb(m, N);
  } else if (kind == kGenPhilox) {
    lpb = N <= 32 ? 32 : kWaveSize;  // lane i = agent i; two boards per wavefront when they fit
    Geo probe;
    probe.words = (m + 4) * (m + 4);
    if (lds_bytes(probe, lpb) > 64u * 1024u) lpb = kWaveSize;
  } else if (kind == kGenSeedExt) {
#if defined(RBG_SE_REFILL_LPB)  // developer probe (tools/probe_envgen.py --flags)
    lpb = RBG_SE_REFILL_LPB;
#else
    lpb = 16;
#endif
    while (lpb < N) lpb *= 2;
  }
  if (int rc = make_geo(m, N, &a.g, prefix, lpb)) return rc;
  size_t lds = lds_bytes(a.g, lpb);
  if (kind == kGenSeedExt) {  // as rbg_seed_extension_generate: fewer, fuller waves while LDS fits
    for (;;) {
      const size_t per_board = (size_t)((a.g.words + 3) & ~3) + (size_t)(4 * lpb) +
                               (size_t)a.gen.extra_words;
      lds = (size_t)wpb * (size_t)(kWaveSize / lpb) * per_board * sizeof(uint32_t);
      if (lds <= 64u * 1024u) break;
      if (lpb < kWaveSize) lpb *= 2; else if (wpb > 1) wpb /= 2; else break;
    }
    if (lds > 64u * 1024u) return fail(RBG_E_SHAPE, "rbg_stash_refill: board too large");
    if (int rc = make_geo(m, N, &a.g, prefix, lpb)) return rc;
  }
  if (B == 0) return 0;
  if (!stash) return fail(RBG_E_NULL, "rbg_stash_refill: NULL stash");
  if (int rc = check_align(a.g, stash->next.grid, nullptr, nullptr)) return rc;
  if (int rc = fill_stash(stash, &a.stash, "rbg_stash_refill")) return rc;
  if (mode != 0 && !a.stash.scratch)
    return fail(RBG_E_NULL, "rbg_stash_list / rbg_stash_refill_listed: the stash has no scratch area");
  if (mode == 1 && a.stash.inflight && (claim_ready < 1 || skip_ready < claim_ready))
    return fail(RBG_E_PARAM, "rbg_stash_list: need 1 <= claim_ready <= skip_ready");
  a.B = B * stash->depth;  // the kernel works on rows
  a.N = N;
  a.envs = B;
  a.claim_ready = claim_ready;
  a.skip_ready = skip_ready;
  // (the list launch's miss snapshot counts its waves in 20 bits of the work word)
  if (mode == 1 && snap && ((long long)a.B + 255) / 256 * 4 < (1ll << 20)) {
    a.miss_acc = (unsigned long long *)snap->miss_acc;
    a.miss_word = (unsigned long long *)snap->miss_word;
    a.miss_host = (unsigned long long *)snap_host;
    a.stamp = snap->stamp;
  }
  const int envs_per_block = wpb * (kWaveSize / lpb);
  dim3 grid((a.B + envs_per_block - 1) / envs_per_block), block(kWaveSize * wpb);
  hipStream_t st = (hipStream_t)stream;
  const bool list_here = mode == 1 || (mode == 0 && kind == kGenSeedExt && a.stash.scratch &&
                                       lpb < kWaveSize);
  if (list_here) {
    // several boards per wavefront: list the requested rows first, then pack them densely
    if (int rc = check_hip(hipMemsetAsync(a.stash.scratch, 0, sizeof(int32_t), st),
                           "rbg_stash_refill")) return rc;
    if (mode == 0) a.stash.inflight = nullptr;  // one stream, nothing overlaps: no claims
    hipLaunchKernelGGL(stash_list_kernel, dim3((a.B + 255) / 256), dim3(256), 0, st, a);
    if (mode == 1) return check_hip(hipGetLastError(), "rbg_stash_list");
  }
  a.listed = (list_here || mode == 2) ? 1 : 0;
  // a listed launch strides over its list: four workgroups per CU at most (see the kernel)
  if (a.listed && grid.x > 4u * 256u) grid.x = 4u * 256u;
  if (kind == kGenUniform) {
    hipLaunchKernelGGL((stash_refill_gen_kernel<64, kGenUniform>), grid, block, lds, st, a);
  } else if (kind == kGenSeqRw) {
    hipLaunchKernelGGL((stash_refill_gen_kernel<64, kGenSeqRw>), grid, block, lds, st, a);
  } else if (kind == kGenDataset) {
    hipLaunchKernelGGL((stash_refill_gen_kernel<64, kGenDataset>), grid, block, lds, st, a);
  } else if (kind == kGenPhilox) {
    if (lpb == 32) {
      hipLaunchKernelGGL((stash_refill_gen_kernel<32, kGenPhilox>), grid, block, lds, st, a);
    } else {
      hipLaunchKernelGGL((stash_refill_gen_kernel<64, kGenPhilox>), grid, block, lds, st, a);
    }
  } else if (kind == kGenSeedExt) {
    if (lpb == 16) {
      hipLaunchKernelGGL((stash_refill_gen_kernel<16, kGenSeedExt>), grid, block, lds, st, a);
    } else if (lpb == 32) {
      hipLaunchKernelGGL((stash_refill_gen_kernel<32, kGenSeedExt>), grid, block, lds, st, a);
    } else {
      hipLaunchKernelGGL((stash_refill_gen_kernel<64, kGenSeedExt>), grid, block, lds, st, a);
    }
  } else if (a.listed) {  // the walk from a list (rolling refills)
    if (lpb == 16) {
      hipLaunchKernelGGL((stash_refill_gen_kernel<16, kGenPrw>), grid, block, lds, st, a);
    } else if (lpb == 32) {
      hipLaunchKernelGGL((stash_refill_gen_kernel<32, kGenPrw>), grid, block, lds, st, a);
    } else {
      hipLaunchKernelGGL((stash_refill_gen_kernel<64, kGenPrw>), grid, block, lds, st, a);
    }
  } else if (lpb == 16) {
    hipLaunchKernelGGL(stash_refill_kernel<16>, grid, block, lds, st, a);
  } else if (lpb == 32) {
    hipLaunchKernelGGL(stash_refill_kernel<32>, grid, block, lds, st, a);
  } else {
    hipLaunchKernelGGL(stash_refill_kernel<64>, grid, block, lds, st, a);
  }
  return check_hip(hipGetLastError(), "rbg_stash_refill");
}

int rbg_stash_refill_gen(const rbg_generator *generator, int B, int m, int N,
                         const rbg_stash *stash, void *stream) {
  return stash_refill_impl(generator, B, m, N, stash, stream, 0);
}

int rbg_stash_list(int B, int m, int N, const rbg_stash *stash, int claim_ready, int skip_ready,
                   void *stream) {
  return stash_refill_impl(nullptr, B, m, N, stash, stream, 1, claim_ready, skip_ready);
}

// One epoch boundary of the rolling form (include/rbg_prw.h::rbg_stash_roll_call): what
// session.EpochStash queued with five event operations, two library calls and three torch kernels
// from Python.
int rbg_stash_roll(const rbg_stash_roll_call *c) {
  if (!c) return fail(RBG_E_NULL, "rbg_stash_roll: NULL call");
  if (!c->stash || !c->launched || !c->listed || !c->done)
    return fail(RBG_E_NULL, "rbg_stash_roll: NULL stash / event");
  if (!c->stash->inflight || !c->stash->scratch)
    return fail(RBG_E_NULL, "rbg_stash_roll: the stash has no in-flight claims / scratch area");
  if (c->miss_word && (!c->miss_acc || !c->miss_host))
    return fail(RBG_E_NULL, "rbg_stash_roll: incomplete miss snapshot (acc, word, host)");
  hipStream_t step = (hipStream_t)c->step_stream, gen = (hipStream_t)c->gen_stream;
  const char *who = "rbg_stash_roll";
  if (int rc = check_hip(hipEventRecord((hipEvent_t)c->launched, step), who)) return rc;
  if (c->await_done)
    if (int rc = check_hip(hipStreamWaitEvent(step, (hipEvent_t)c->await_done, 0), who)) return rc;
  if (int rc = check_hip(hipStreamWaitEvent(gen, (hipEvent_t)c->launched, 0), who)) return rc;
  if (c->after_listed)
    if (int rc = check_hip(hipStreamWaitEvent(gen, (hipEvent_t)c->after_listed, 0), who)) return rc;
  // The snapshot goes straight into the host's pinned word when the device can address it
  // (mapped host memory: hipHostMalloc / hipHostRegister, what torch's pin_memory() gives);
  // otherwise 8 bytes are copied behind the list launch.
  void *host_dev = nullptr;
  if (c->miss_word) {
    static thread_local const void *seen_host = nullptr;
    static thread_local void *seen_dev = nullptr;
    if (seen_host != c->miss_host) {
      void *d = nullptr;
      if (hipHostGetDevicePointer(&d, c->miss_host, 0) != hipSuccess) {
        (void)hipGetLastError();  // not mapped: the copy below
        d = nullptr;
      }
      seen_host = c->miss_host;
      seen_dev = d;
    }
    host_dev = seen_dev;
  }
  if (int rc = stash_refill_impl(nullptr, c->B, c->m, c->N, c->stash, gen, 1, c->claim_ready,
                                 c->skip_ready, c->miss_word ? c : nullptr, host_dev))
    return rc;
  if (c->miss_word && !host_dev && c->B > 0)
    if (int rc = check_hip(hipMemcpyAsync(c->miss_host, c->miss_word, sizeof(int64_t),
                                          hipMemcpyDeviceToHost, gen), who)) return rc;
  if (int rc = check_hip(hipEventRecord((hipEvent_t)c->listed, gen), who)) return rc;
  if (int rc = stash_refill_impl(c->generator, c->B, c->m, c->N, c->stash, gen, 2)) return rc;
  return check_hip(hipEventRecord((hipEvent_t)c->done, gen), who);
}

int rbg_stash_refill_listed(const rbg_generator *generator, int B, int m, int N,
                            const rbg_stash *stash, void *stream) {
  return stash_refill_impl(generator, B, m, N, stash, stream, 2);
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
  return rbg_connector_step_gen(nullptr, B, m, N, state_in, state_out, action, nullptr, 0u, params,
                                timestep, nullptr, 0, 0, stream);
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
  return rbg_connector_rollout_gen(nullptr, B, m, N, state_in, state_out, policy_keys, t0, n_steps,
                                   params, stats, last_timestep, trajectory, trajectory_action,
                                   stash, tick, ready_delay, stream);
}

int rbg_connector_rollout_gen(const rbg_generator *generator, int B, int m, int N,
                              const rbg_state *state_in, const rbg_state *state_out,
                              const uint32_t *policy_keys, uint32_t t0, int n_steps,
                              const rbg_env_params *params, const rbg_rollout_stats *stats,
                              const rbg_timestep *last_timestep, const rbg_timestep *trajectory,
                              int32_t *trajectory_action, const rbg_stash *stash, int tick,
                              int ready_delay, void *stream) {
  if (B < 0) return fail(RBG_E_SHAPE, "rbg_connector_rollout: B < 0");
  if (n_steps < 1) return fail(RBG_E_PARAM, "rbg_connector_rollout: n_steps < 1");
  if (!params) return fail(RBG_E_NULL, "rbg_connector_rollout: NULL params");
  if (params->time_limit < 1) return fail(RBG_E_PARAM, "rbg_connector_rollout: time_limit < 1");
  RolloutArgs a;
  memset(&a, 0, sizeof(a));
  if (int rc = make_geo(m, N, &a.g)) return rc;
  int prefix = N;
  if (int rc = fill_gen(generator, m, N, &a.gen, &prefix, "rbg_connector_rollout")) return rc;
  const int kind = a.gen.kind;
  const int lpb = kind == kGenPrw ? pick_lpb(m, N) : kWaveSize;
  if (int rc = make_geo(m, N, &a.g, prefix, lpb)) return rc;
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
  if (stash && stash->depth > kRolloutStashDepth)
    return fail(RBG_E_PARAM, "rbg_connector_rollout: at most 2 stash slots per env in a rollout launch");
  if (stash)
    if (int rc = check_tick(tick, ready_delay, "rbg_connector_rollout")) return rc;
  a.tick = tick; a.ready_delay = ready_delay;
  const bool wants_stats = stats && (stats->episodes || stats->connections_made ||
                                     stats->unconnected_agent_steps || stats->final_connections ||
                                     stats->final_path_length);
  if (n_steps == 1 && last_timestep && last_timestep->obs_grid && !wants_stats && !record) {
    // "random policy + step" with the observation written: the HBM-bound shape of the per-step
    // kernel (one env per wave, 16 waves per CU in flight), not the LDS-resident rollout's
    EnvArgs e;
    memset(&e, 0, sizeof(e));
    if (int rc = make_geo(m, N, &e.g, prefix)) return rc;  // (the shuffle threshold depends on the lane count)
    e.B = B; e.N = N; e.in = a.in; e.out = a.out; e.p = a.p; e.ts = a.ts; e.mode = kModeStep;
    e.stash = a.stash; e.tick = tick; e.ready_delay = ready_delay;
    e.policy_keys = policy_keys; e.policy_t = t0; e.gen = a.gen;
    return launch_env(e, stream, "rbg_connector_rollout");
  }
  int envs_per_block = kWavesPerBlock * (kWaveSize / lpb);
  size_t lds = lds_bytes(a.g, lpb) + (size_t)envs_per_block * (size_t)a.gen.extra_words *
                                         sizeof(uint32_t);
  while (kind != kGenPrw && envs_per_block > 1 && lds > 64u * 1024u) {
    envs_per_block /= 2;  // (one env per wave for these kinds: fewer waves per workgroup)
    lds /= 2;
  }
  if (lds > 64u * 1024u) return fail(RBG_E_SHAPE, "rbg_connector_rollout: board too large");
  const dim3 grid((B + envs_per_block - 1) / envs_per_block), block(kBlock);
  hipStream_t st = (hipStream_t)stream;
  if (kind != kGenPrw) {
    a.block_threads = kWaveSize * envs_per_block;
    if (kind == kGenUniform) launch_rollout_gen<kGenUniform>(a, record, grid, lds, st);
    else if (kind == kGenSeqRw) launch_rollout_gen<kGenSeqRw>(a, record, grid, lds, st);
    else if (kind == kGenSeedExt) launch_rollout_gen<kGenSeedExt>(a, record, grid, lds, st);
    else if (kind == kGenPhilox) launch_rollout_gen<kGenPhilox>(a, record, grid, lds, st);
    else launch_rollout_gen<kGenDataset>(a, record, grid, lds, st);
    return check_hip(hipGetLastError(), "rbg_connector_rollout");
  }
  if (record) {
    return check_hip((hipError_t)launch_rollout_record(a, lpb, grid.x, lds, st),
                     "rbg_connector_rollout_record");
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
