// step_groups.cpp -- developer probe: env.step over one batch as G independent groups of envs on
// G HIP streams, launched from C (no Python between launches).  A group's step t + 1 depends only
// on its own step t, so the launch floor of one group can hide behind the store stream of the
// others.  Uses the fused "random policy + step" launch (rbg_connector_rollout, n_steps = 1, full
// timestep written), auto-reset off so that no walk runs.
//   hipcc --offload-arch=gfx950 -O2 -I include tools/microbench/step_groups.cpp \
//         -L routing_board_generation_amd/lib -lrbg_prw -Wl,-rpath,$PWD/routing_board_generation_amd/lib -o step_groups
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <vector>

#include "rbg_prw.h"

template <typename T>
static T *dalloc(size_t n) {
  void *p = nullptr;
  if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) exit(2);
  return (T *)p;
}

int main() {
  const int B = 4096, m = 16, N = 10, steps = 600;
  uint32_t *keys = dalloc<uint32_t>((size_t)B * 2);
  if (rbg_split(0u, 0u, B, keys, nullptr)) return 3;
  for (int G : {1, 2, 4}) {
    const int per = B / G;
    std::vector<hipStream_t> streams(G);
    std::vector<rbg_state> st(G);
    std::vector<rbg_timestep> ts(G);
    for (int g = 0; g < G; ++g) {
      if (hipStreamCreate(&streams[g]) != hipSuccess) return 2;
      st[g] = {dalloc<int32_t>((size_t)per * m * m), dalloc<int32_t>(per),
               dalloc<int32_t>((size_t)per * N * 2), dalloc<int32_t>((size_t)per * N * 2),
               dalloc<int32_t>((size_t)per * N * 2), dalloc<uint32_t>((size_t)per * 2)};
      ts[g] = {dalloc<int8_t>(per), dalloc<float>((size_t)per * N), dalloc<float>((size_t)per * N),
               dalloc<int32_t>((size_t)per * N * m * m), dalloc<uint8_t>((size_t)per * N * 5),
               dalloc<int32_t>(per), dalloc<int32_t>(per), dalloc<float>(per), dalloc<int32_t>(per),
               nullptr, nullptr};
      if (rbg_connector_reset(keys + (size_t)g * per * 2, per, m, N, &st[g], &ts[g], streams[g])) return 3;
    }
    rbg_env_params params = {1 << 30, -0.03f, 0.1f, 0};
    auto run = [&](int t0, int n) {
      for (int t = t0; t < t0 + n; ++t)
        for (int g = 0; g < G; ++g)
          if (rbg_connector_rollout(per, m, N, &st[g], &st[g], keys + (size_t)g * per * 2, (uint32_t)t, 1,
                                    &params, nullptr, &ts[g], streams[g]))
            exit(3);
    };
    run(0, 50);
    hipDeviceSynchronize();
    const auto a = std::chrono::steady_clock::now();
    run(50, steps);
    const auto h = std::chrono::steady_clock::now();
    hipDeviceSynchronize();
    const auto b = std::chrono::steady_clock::now();
    const double us = std::chrono::duration<double, std::micro>(b - a).count() / steps;
    const double host = std::chrono::duration<double, std::micro>(h - a).count() / steps;
    printf("groups=%d: %6.2f us per step of %d envs (host issue %5.2f us per step) = %6.1f M env-steps/s\n",
           G, us, B, host, B / us);
    fflush(stdout);
  }
  return 0;
}
