// Developer microbenchmark: sustained issue rate of individual VALU instructions on gfx950
// (the Threefry mix of rbg_device.h is add / rotate / xor).
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/valu_rate tools/microbench/valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../routing_board_generation_amd/csrc/rbg_device.h"

#define REP16(x) x x x x x x x x x x x x x x x x
#define OP_KERNEL(NAME, ASM)                                                      \
  __global__ void NAME(uint32_t *out, int reps) {                                 \
    uint32_t a = threadIdx.x, b = blockIdx.x | 1u, c = a ^ 0x1234u, d = b + 77u;  \
    for (int i = 0; i < reps; ++i) {                                              \
      REP16(asm volatile(ASM : "+v"(a), "+v"(c) : "v"(b), "v"(d));)              \
    }                                                                             \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + c;                           \
  }

// two independent chains (a, c) per statement = 2 instructions
OP_KERNEL(k_add, "v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %3")
OP_KERNEL(k_xor, "v_xor_b32 %0, %0, %2\n v_xor_b32 %1, %1, %3")
OP_KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %0, 19\n v_alignbit_b32 %1, %1, %1, 17")
OP_KERNEL(k_add3, "v_add3_u32 %0, %0, %2, %3\n v_add3_u32 %1, %1, %3, %2")
OP_KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 13, %2\n v_lshl_add_u32 %1, %1, 15, %3")
OP_KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 13, %2\n v_lshl_or_b32 %1, %1, 15, %3")
OP_KERNEL(k_lshr, "v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 5, %1")
OP_KERNEL(k_xad, "v_xad_u32 %0, %0, %2, %3\n v_xad_u32 %1, %1, %3, %2")
OP_KERNEL(k_perm, "v_perm_b32 %0, %0, %0, %2\n v_perm_b32 %1, %1, %1, %3")
OP_KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %3, vcc")
OP_KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %2\n v_mul_lo_u32 %1, %1, %3")
OP_KERNEL(k_mul_hi, "v_mul_hi_u32 %0, %0, %2\n v_mul_hi_u32 %1, %1, %3")

// Philox-4x32-10 (Salmon et al., SC'11), the generator BASELINE.json's north_star names: two
// 32x32 -> 64 multiplies per round.  Timed next to Threefry-2x32-20 to price a random word on
// this chip (Philox yields four words per evaluation, Threefry two).
__global__ void k_philox(uint32_t *out, int reps) {
  uint32_t c0 = threadIdx.x, c1 = blockIdx.x, c2 = c0 ^ 0x1234u, c3 = c1 + 77u;
  for (int i = 0; i < reps; ++i) {
    uint32_t k0 = 0xA4093822u, k1 = 0x299F31D0u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
      const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
      c0 = hi1 ^ c1 ^ k0;
      c1 = lo1;
      c2 = hi0 ^ c3 ^ k1;
      c3 = lo0;
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3;
}

__global__ void k_threefry(uint32_t *out, int reps) {
  uint32_t a = threadIdx.x, b = blockIdx.x, c = a ^ 0x1234u, d = b + 77u;
  for (int i = 0; i < reps; ++i) {
    uint32_t y0, y1;
    rbg::threefry(a, b, c, d, y0, y1);
    c = y0;
    d = y1;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c + d;
}

typedef void (*kern_t)(uint32_t *, int);

static void run(const char *name, kern_t k, int blocks, int reps, double instr_per_rep) {
  uint32_t *out;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  k<<<blocks, 256>>>(out, reps);
  (void)hipEventRecord(e0);
  k<<<blocks, 256>>>(out, reps);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = blocks * 4.0 * reps * instr_per_rep / 1024.0;
  printf("%-10s waves/SIMD=%d  %.3f ms  -> %.2f cycles/instr/SIMD @2.4GHz\n", name, blocks * 4 / 1024,
         ms, (ms * 1e-3 * 2.4e9) / instr_per_simd);
  (void)hipFree(out);
}

int main() {
  for (int blocks : {256, 512, 2048}) {
    run("add", k_add, blocks, 4000, 32);
    run("xor", k_xor, blocks, 4000, 32);
    run("alignbit", k_alignbit, blocks, 4000, 32);
    run("add3", k_add3, blocks, 4000, 32);
    run("lshl_add", k_lshl_add, blocks, 4000, 32);
    run("lshl_or", k_lshl_or, blocks, 4000, 32);
    run("lshrrev", k_lshr, blocks, 4000, 32);
    run("xad", k_xad, blocks, 4000, 32);
    run("perm", k_perm, blocks, 4000, 32);
    run("cndmask", k_cndmask, blocks, 4000, 32);
    run("mul_lo", k_mul_lo, blocks, 4000, 32);
    run("mul_hi", k_mul_hi, blocks, 4000, 32);
    run("threefry", k_threefry, blocks, 2000, 82);  // ~82 instructions per evaluation
    run("3fry/eval", k_threefry, blocks, 2000, 1);  // cycles per evaluation (2 words)
    run("philox/ev", k_philox, blocks, 2000, 1);    // cycles per evaluation (4 words)
  }
  return 0;
}
