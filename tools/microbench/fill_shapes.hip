// fill_shapes.hip -- developer probe: write-only fill rate vs. the size of the contiguous chunk a
// wave owns (power-of-two chunks put every wave on the same HBM channels at the same time).
//   hipcc --offload-arch=gfx950 -O3 -o fill_shapes fill_shapes.hip && ./fill_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float v4f __attribute__((ext_vector_type(4)));

template <bool kNT>
__global__ __launch_bounds__(256) void fill_rows(v4f *__restrict__ dst, size_t chunks, int rows) {
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const v4f x = {1.f, 2.f, 3.f, 4.f};
  for (size_t c = wave; c < chunks; c += (size_t)gridDim.x * 4) {
    v4f *p = dst + c * (size_t)rows * 64 + lane;
    for (int u = 0; u < rows; ++u) {
      if (kNT) __builtin_nontemporal_store(x, &p[u * 64]);
      else p[u * 64] = x;
    }
  }
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  v4f *buf;
  if (hipMalloc(&buf, bytes + (1 << 20)) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int rows_list[] = {4, 8, 10, 11, 12, 15, 16, 17, 20, 32, 33};
  for (int nt = 0; nt < 2; ++nt)
    for (int rows : rows_list) {
      const size_t chunks = bytes / ((size_t)rows * 1024);
      float best = 0.f;
      int best_blocks = 0;
      for (int blocks : {2048, 4096, 16384, (int)((chunks + 3) / 4)}) {
        for (int rep = 0; rep < 2; ++rep) {
          if (nt) hipLaunchKernelGGL(fill_rows<true>, dim3(blocks), dim3(256), 0, 0, buf, chunks, rows);
          else hipLaunchKernelGGL(fill_rows<false>, dim3(blocks), dim3(256), 0, 0, buf, chunks, rows);
        }
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int rep = 0; rep < 10; ++rep) {
          if (nt) hipLaunchKernelGGL(fill_rows<true>, dim3(blocks), dim3(256), 0, 0, buf, chunks, rows);
          else hipLaunchKernelGGL(fill_rows<false>, dim3(blocks), dim3(256), 0, 0, buf, chunks, rows);
        }
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const float gbs = (float)(chunks * rows * 1024.0 * 10 / (ms * 1e-3) / 1e9);
        if (gbs > best) { best = gbs; best_blocks = blocks; }
      }
      printf("nt=%d rows=%2d (%2d KiB per wave): %7.1f GB/s (best grid %d)\n", nt, rows, rows, best, best_blocks);
      fflush(stdout);
    }
  return 0;
}
