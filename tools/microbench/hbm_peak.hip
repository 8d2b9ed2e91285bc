// hbm_peak.hip -- measurement yardsticks for bench.py (SURVEY.md 8d asks for the measured HBM
// peak beside the 8 TB/s spec figure): a 16-B-per-lane streaming copy and a write-only fill,
// each with plain and non-temporal accesses.  Not part of the product library.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libhbm_peak.so hbm_peak.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float v4f __attribute__((ext_vector_type(4)));

// Each wave owns contiguous chunks of kChunk float4 rows (64 lanes x 16 B = 1 KiB per row): the
// access shape of the env kernels, whose waves write N consecutive 1 KiB observation planes.
constexpr int kRows = 16;  // 16 KiB per wave and chunk

template <bool kNT>
__global__ __launch_bounds__(256) void copy_kernel(v4f *__restrict__ dst,
                                                   const v4f *__restrict__ src, size_t n) {
  const size_t waves = (size_t)gridDim.x * 4, wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  for (size_t c = wave; (c + 1) * kRows * 64 <= n; c += waves) {
    const size_t base = c * kRows * 64 + lane;
    v4f r[kRows];
#pragma unroll
    for (int u = 0; u < kRows; ++u)
      r[u] = kNT ? __builtin_nontemporal_load(&src[base + u * 64]) : src[base + u * 64];
#pragma unroll
    for (int u = 0; u < kRows; ++u) {
      if (kNT) {
        __builtin_nontemporal_store(r[u], &dst[base + u * 64]);
      } else {
        dst[base + u * 64] = r[u];
      }
    }
  }
}

template <bool kNT>
__global__ __launch_bounds__(256) void fill_kernel(v4f *__restrict__ dst, size_t n, float v) {
  const size_t waves = (size_t)gridDim.x * 4, wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const v4f x = {v, v, v, v};
  for (size_t c = wave; (c + 1) * kRows * 64 <= n; c += waves) {
    const size_t base = c * kRows * 64 + lane;
#pragma unroll
    for (int u = 0; u < kRows; ++u) {
      if (kNT) {
        __builtin_nontemporal_store(x, &dst[base + u * 64]);
      } else {
        dst[base + u * 64] = x;
      }
    }
  }
}

extern "C" {

// bytes must be a multiple of 16 KiB; `blocks` workgroups of 256 threads (0 = 256 CUs x 8)
int hbm_copy(void *dst, const void *src, size_t bytes, int nt, int blocks, void *stream) {
  const size_t n = bytes / 16;
  const dim3 grid(blocks > 0 ? blocks : 2048), block(256);
  if (nt) {
    hipLaunchKernelGGL(copy_kernel<true>, grid, block, 0, (hipStream_t)stream, (v4f *)dst,
                       (const v4f *)src, n);
  } else {
    hipLaunchKernelGGL(copy_kernel<false>, grid, block, 0, (hipStream_t)stream, (v4f *)dst,
                       (const v4f *)src, n);
  }
  return (int)hipGetLastError();
}

int hbm_fill(void *dst, size_t bytes, int nt, int blocks, void *stream) {
  const size_t n = bytes / 16;
  const dim3 grid(blocks > 0 ? blocks : 2048), block(256);
  if (nt) {
    hipLaunchKernelGGL(fill_kernel<true>, grid, block, 0, (hipStream_t)stream, (v4f *)dst, n, 1.0f);
  } else {
    hipLaunchKernelGGL(fill_kernel<false>, grid, block, 0, (hipStream_t)stream, (v4f *)dst, n, 1.0f);
  }
  return (int)hipGetLastError();
}

}  // extern "C"
