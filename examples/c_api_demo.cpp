// c_api_demo.cpp -- the drop-in boundary without Python or torch: a plain C++ host program that
// links librbg_prw.so (include/rbg_prw.h) and the HIP runtime, generates the BASELINE configs[1]
// batch (16x16 grid, 10 agents, 4096 boards from split(PRNGKey(0), 4096)), resets a Connector
// env on it and runs a 50-step fused random rollout; then steps the same env coupled to another
// generator (SequentialRandomWalk) through rbg_generator / rbg_step_call / rbg_stash_refill_gen,
// and once more with that generator's refills rolling beside the stepping stream (rbg_stash_roll).
//
//   hipcc --offload-arch=gfx950 -O2 -I include examples/c_api_demo.cpp \
//         -L routing_board_generation_amd/lib -lrbg_prw -Wl,-rpath,'$ORIGIN' -o c_api_demo
//
// Prints an FNV-1a checksum of the solved grids (tests/test_gpu_parity.py compares it with the
// oracle's) and the rollout counters.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "rbg_prw.h"

#define HIP_OK(x)                                                        \
  do {                                                                   \
    hipError_t e_ = (x);                                                 \
    if (e_ != hipSuccess) {                                              \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));            \
      return 2;                                                          \
    }                                                                    \
  } while (0)
#define RBG_OK(x)                                                        \
  do {                                                                   \
    int rc_ = (x);                                                       \
    if (rc_ != 0) {                                                      \
      fprintf(stderr, "%s -> %d: %s\n", #x, rc_, rbg_last_error());      \
      return 3;                                                          \
    }                                                                    \
  } while (0)

template <typename T>
static T *dalloc(size_t n) {
  void *p = nullptr;
  if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) exit(2);
  return (T *)p;
}

int main(int argc, char **argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4096, m = 16, N = 10, n_steps = 50;
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  printf("%s, ABI %d\n", rbg_version(), rbg_abi_version());

  uint32_t *keys = dalloc<uint32_t>((size_t)B * 2);
  RBG_OK(rbg_split(0u, 0u, B, keys, stream));  // jax.random.split(PRNGKey(0), B)

  // vmap(ParallelRandomWalkGenerator(16, 10).__call__)(keys), solved boards included
  int32_t *solved = dalloc<int32_t>((size_t)B * m * m), *train = dalloc<int32_t>((size_t)B * m * m);
  int32_t *start = dalloc<int32_t>((size_t)B * N * 2), *target = dalloc<int32_t>((size_t)B * N * 2);
  int32_t *iters = dalloc<int32_t>(B);
  uint32_t *key_out = dalloc<uint32_t>((size_t)B * 2);
  RBG_OK(rbg_prw_generate(keys, B, m, N, 1, solved, train, start, target, nullptr, nullptr, key_out,
                          iters, stream));

  // env.reset + 50 random-policy steps with fused auto-reset
  rbg_state st = {dalloc<int32_t>((size_t)B * m * m), dalloc<int32_t>(B),
                  dalloc<int32_t>((size_t)B * N * 2), dalloc<int32_t>((size_t)B * N * 2),
                  dalloc<int32_t>((size_t)B * N * 2), dalloc<uint32_t>((size_t)B * 2)};
  rbg_timestep ts = {};  // no timestep outputs wanted: every member NULL
  RBG_OK(rbg_connector_reset(keys, B, m, N, &st, &ts, stream));
  rbg_rollout_stats stats = {dalloc<int32_t>(B), dalloc<int32_t>(B), dalloc<int32_t>(B),
                             dalloc<int32_t>(B), dalloc<int32_t>(B)};
  rbg_env_params params = {50, -0.03f, 0.1f, 1};
  RBG_OK(rbg_connector_rollout(B, m, N, &st, &st, keys, 0u, n_steps, &params, &stats, nullptr,
                               stream));
  HIP_OK(hipStreamSynchronize(stream));

  std::vector<int32_t> h_solved((size_t)B * m * m), h_iters(B), h_eps(B), h_conn(B), h_pen(B);
  HIP_OK(hipMemcpy(h_solved.data(), solved, h_solved.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(h_iters.data(), iters, (size_t)B * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(h_eps.data(), stats.episodes, (size_t)B * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(h_conn.data(), stats.connections_made, (size_t)B * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(h_pen.data(), stats.unconnected_agent_steps, (size_t)B * 4, hipMemcpyDeviceToHost));
  uint64_t fnv = 1469598103934665603ull;
  for (int32_t v : h_solved) {
    fnv ^= (uint32_t)v;
    fnv *= 1099511628211ull;
  }
  long long it_sum = 0, eps = 0, conn = 0, pen = 0;
  for (int b = 0; b < B; ++b) {
    it_sum += h_iters[b];
    eps += h_eps[b];
    conn += h_conn[b];
    pen += h_pen[b];
  }
  printf("boards %d  walk_iters_sum %lld  solved_fnv1a %016llx\n", B, it_sum,
         (unsigned long long)fnv);
  printf("rollout %d steps: episodes %lld  connections %lld  unconnected_agent_steps %lld  "
         "return %.3f\n", n_steps, eps, conn, pen, 0.1 * (double)conn - 0.03 * (double)pen);

  // The same env hot path for another generator of setup_train.py:112-152 -- online_seq_rw:
  // VmapAutoResetWrapper(Connector(SequentialRandomWalkGenerator(16, 10))), 20 NOOP steps with a
  // time limit of 6, next episodes from the generator's own stash (refilled every 4 steps), the
  // step described by one rbg_step_call.
  rbg_generator gen = {};
  gen.kind = RBG_GEN_SEQUENTIAL_RW;
  const int D = 2;  // stash slots per env
  rbg_stash stash = {};
  stash.next = {dalloc<int32_t>((size_t)D * B * m * m), dalloc<int32_t>((size_t)D * B),
                dalloc<int32_t>((size_t)D * B * N * 2), dalloc<int32_t>((size_t)D * B * N * 2),
                dalloc<int32_t>((size_t)D * B * N * 2), dalloc<uint32_t>((size_t)D * B * 2)};
  stash.tag = dalloc<uint32_t>((size_t)D * B * 2);
  stash.ready_at = dalloc<int32_t>((size_t)D * B);
  stash.refill_key = dalloc<uint32_t>((size_t)D * B * 2);
  stash.hits = dalloc<int32_t>(B);
  stash.misses = dalloc<int32_t>(B);
  stash.depth = D;
  HIP_OK(hipMemsetAsync(stash.tag, 0, (size_t)D * B * 8, stream));
  HIP_OK(hipMemsetAsync(stash.ready_at, 0, (size_t)D * B * 4, stream));
  HIP_OK(hipMemsetAsync(stash.refill_key, 0, (size_t)D * B * 8, stream));
  HIP_OK(hipMemsetAsync(stash.hits, 0, (size_t)B * 4, stream));
  HIP_OK(hipMemsetAsync(stash.misses, 0, (size_t)B * 4, stream));
  int32_t *action = dalloc<int32_t>((size_t)B * N);
  HIP_OK(hipMemsetAsync(action, 0, (size_t)B * N * 4, stream));  // NOOP
  const int K = 4;  // refill period: requests of launches 0..3 are served from launch 4 on
  int tick = 0;
  RBG_OK(rbg_connector_reset_gen(&gen, keys, B, m, N, &st, &ts, &stash, tick, K - tick, stream));
  ++tick;
  rbg_step_call call = {};
  call.generator = &gen;
  call.B = B; call.m = m; call.N = N;
  call.state_in = call.state_out = st;  // in place
  call.timestep = ts;
  call.action = action;
  call.params = {6, -0.03f, 0.1f, 1};
  call.stash = &stash;
  call.stream = stream;
  for (int s = 0; s < 20; ++s, ++tick) {
    call.tick = tick;
    call.ready_delay = K * (tick / K + 1) - tick;
    RBG_OK(rbg_connector_step_call(&call));
    if (tick % K == K - 1) RBG_OK(rbg_stash_refill_gen(&gen, B, m, N, &stash, stream));
  }
  HIP_OK(hipStreamSynchronize(stream));
  std::vector<int32_t> h_grid((size_t)B * m * m), h_hits(B), h_miss(B);
  HIP_OK(hipMemcpy(h_grid.data(), st.grid, h_grid.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(h_hits.data(), stash.hits, (size_t)B * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(h_miss.data(), stash.misses, (size_t)B * 4, hipMemcpyDeviceToHost));
  fnv = 1469598103934665603ull;
  for (int32_t v : h_grid) {
    fnv ^= (uint32_t)v;
    fnv *= 1099511628211ull;
  }
  long long hits = 0, misses = 0;
  for (int b = 0; b < B; ++b) {
    hits += h_hits[b];
    misses += h_miss[b];
  }
  printf("seq_rw env after 20 steps: grid_fnv1a %016llx  stash hits %lld  misses %lld\n",
         (unsigned long long)fnv, hits, misses);

  // The same env once more with its refills ROLLING beside the stepping stream (what
  // VmapAutoResetWrapper does by default for this generator): four slots per env, a refill
  // behind every epoch of K = 4 launches on one of two generator streams, usable `lead` = 2
  // epochs later; ONE rbg_stash_roll call per epoch boundary (events of this program), 40 steps.
  {
    const int D4 = 4, Kr = 4, lead = 2, ring = lead + 1, R = D4 * B;
    hipStream_t gstream[2];
    HIP_OK(hipStreamCreate(&gstream[0]));
    HIP_OK(hipStreamCreate(&gstream[1]));
    hipEvent_t launched[3], listed[3], done[3];
    for (int i = 0; i < ring; ++i) {
      HIP_OK(hipEventCreateWithFlags(&launched[i], hipEventDisableTiming));
      HIP_OK(hipEventCreateWithFlags(&listed[i], hipEventDisableTiming));
      HIP_OK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
    }
    rbg_stash rs = {};
    rs.next = {dalloc<int32_t>((size_t)R * m * m), dalloc<int32_t>(R),
               dalloc<int32_t>((size_t)R * N * 2), dalloc<int32_t>((size_t)R * N * 2),
               dalloc<int32_t>((size_t)R * N * 2), dalloc<uint32_t>((size_t)R * 2)};
    rs.tag = dalloc<uint32_t>((size_t)R * 2);
    rs.ready_at = dalloc<int32_t>(R);
    rs.refill_key = dalloc<uint32_t>((size_t)R * 2);
    rs.hits = dalloc<int32_t>(B);
    rs.misses = dalloc<int32_t>(B);
    rs.inflight = dalloc<int32_t>(R);              // claims of the refills in flight
    rs.far_key = dalloc<uint32_t>((size_t)B * 2);  // one new request per restart that hit
    rs.depth = D4;
    HIP_OK(hipMemsetAsync(rs.tag, 0, (size_t)R * 8, stream));
    HIP_OK(hipMemsetAsync(rs.ready_at, 0, (size_t)R * 4, stream));
    HIP_OK(hipMemsetAsync(rs.refill_key, 0, (size_t)R * 8, stream));
    HIP_OK(hipMemsetAsync(rs.hits, 0, (size_t)B * 4, stream));
    HIP_OK(hipMemsetAsync(rs.misses, 0, (size_t)B * 4, stream));
    HIP_OK(hipMemsetAsync(rs.inflight, 0, (size_t)R * 4, stream));
    HIP_OK(hipMemsetAsync(rs.far_key, 0, (size_t)B * 8, stream));
    rbg_stash per_refill[3];  // every refill in flight lists its rows into a scratch area of its own
    for (int i = 0; i < ring; ++i) {
      per_refill[i] = rs;
      per_refill[i].scratch = dalloc<int32_t>((size_t)R + 1);
    }
    rs.scratch = per_refill[0].scratch;
    // tick 0: the reset; its requests are served by one refill on the stepping stream (tick 1 on)
    RBG_OK(rbg_connector_reset_gen(&gen, keys, B, m, N, &st, &ts, &rs, 0, 1, stream));
    RBG_OK(rbg_stash_refill_gen(&gen, B, m, N, &rs, stream));
    call.stash = &rs;
    call.state_in = call.state_out = st;
    for (int t = 1; t <= 40; ++t) {  // epoch e covers the ticks 1 + e Kr .. (e + 1) Kr
      const int e = (t - 1) / Kr;
      call.tick = t;
      call.ready_delay = 1 + Kr * (e + lead) - t;  // usable from the first tick of epoch e + lead
      RBG_OK(rbg_connector_step_call(&call));
      if ((t - 1) % Kr == Kr - 1) {
        rbg_stash_roll_call roll = {};
        roll.generator = &gen;
        roll.B = B; roll.m = m; roll.N = N;
        roll.claim_ready = 1 + Kr * (e + lead);
        roll.skip_ready = roll.claim_ready + Kr;
        roll.stash = &per_refill[e % ring];
        roll.step_stream = stream;
        roll.gen_stream = gstream[e % 2];
        roll.launched = launched[e % ring];
        roll.listed = listed[e % ring];
        roll.done = done[e % ring];
        roll.await_done = e + 1 >= lead ? done[(e + 1 - lead) % ring] : nullptr;
        roll.after_listed = e > 0 ? listed[(e - 1) % ring] : nullptr;
        RBG_OK(rbg_stash_roll(&roll));
      }
    }
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(h_grid.data(), st.grid, h_grid.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h_hits.data(), rs.hits, (size_t)B * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h_miss.data(), rs.misses, (size_t)B * 4, hipMemcpyDeviceToHost));
    fnv = 1469598103934665603ull;
    for (int32_t v : h_grid) {
      fnv ^= (uint32_t)v;
      fnv *= 1099511628211ull;
    }
    hits = misses = 0;
    for (int b = 0; b < B; ++b) {
      hits += h_hits[b];
      misses += h_miss[b];
    }
    printf("seq_rw env, rolling refills, after 40 steps: rolling_fnv1a %016llx  rolling hits %lld  "
           "rolling misses %lld\n", (unsigned long long)fnv, hits, misses);
  }
  return 0;
}
