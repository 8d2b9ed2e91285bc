// dep_latency.hip -- developer microbenchmark: issue-to-issue latency of DEPENDENT vector ALU
// instructions for one wave alone on its SIMD (what bounds the serial Threefry chains of the
// SeedExtension sweep and of the walk's stragglers).
//   hipcc --offload-arch=gfx950 -O3 -o dep_latency dep_latency.hip && ./dep_latency
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

#define CHAIN(name, body)                                                            \
  __global__ void name(uint32_t *out, uint64_t *cycles) {                            \
    uint32_t x = threadIdx.x, y = threadIdx.x * 7 + 1, z = 0x9E3779B9u, w = threadIdx.x + 5;              \
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();                            \
    const uint64_t t0 = __builtin_amdgcn_s_memtime();                                \
    for (int i = 0; i < 16; ++i) {                                                   \
      REP64(body)                                                                    \
    }                                                                                \
    const uint64_t t1 = __builtin_amdgcn_s_memtime();                                \
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();                            \
    out[threadIdx.x] = x + y + w + z;                                                        \
    if (threadIdx.x == 0) {                                                          \
      cycles[0] = t1 - t0;                                                           \
      cycles[1] = r1 - r0; /* 100 MHz ticks */                                       \
    }                                                                                \
  }

CHAIN(k_add, asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(z));)
CHAIN(k_xor, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(z));)
CHAIN(k_alignbit, asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(x));)
CHAIN(k_add3, asm volatile("v_add3_u32 %0, %0, %1, 1" : "+v"(x) : "v"(z));)
CHAIN(k_xad, asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(x) : "v"(z));)
// one Threefry round as the compiler emits it: add ; alignbit ; xor
CHAIN(k_round, asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 13\n v_xor_b32 %1, %1, %0"
                            : "+v"(x), "+v"(y));)
// the same round with the independent rotate issued first
CHAIN(k_round_rot_first, asm volatile("v_alignbit_b32 %2, %1, %1, 13\n v_add_u32 %0, %0, %1\n v_xor_b32 %1, %2, %0"
                                      : "+v"(x), "+v"(y), "+v"(z));)
// two independent chains interleaved (what a second wave, or a second board, would add)
CHAIN(k_two_chains, asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2" : "+v"(x), "+v"(y) : "v"(z));)
// scalar ALU: a dependent chain, and one independent s_add behind every dependent v_add
CHAIN(k_salu, asm volatile("s_add_u32 %0, %0, 1" : "+s"(z) : : "scc");)
CHAIN(k_valu_salu, asm volatile("v_add_u32 %0, %0, %2\n s_add_u32 %1, %1, 1" : "+v"(x), "+s"(z) : "v"(y) : "scc");)
// a dependent v_add with TWO independent v_adds behind it (how far does the 4-cycle slot go?)
CHAIN(k_three_chains, asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n v_add_u32 %3, %3, %2"
                                   : "+v"(x), "+v"(y) , "+v"(z), "+v"(w) : );)
// dependent DPP move
CHAIN(k_dpp, asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x));)

int main() {
  uint32_t *out;
  uint64_t *cyc, h[2];
  hipMalloc(&out, 64 * 4);
  hipMalloc(&cyc, 16);
#define RUN(k, per)                                                                  \
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc); \
  hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);                                      \
  printf("%-22s %6.2f s_memtime ticks = %6.2f ns per %s  (s_memtime runs at %.0f MHz)\n", #k, \
         (double)h[0] / (16.0 * 64.0), (double)h[1] * 10.0 / (16.0 * 64.0), per,     \
         (double)h[0] / ((double)h[1] * 0.01));
  RUN(k_add, "dependent v_add_u32")
  RUN(k_xor, "dependent v_xor_b32")
  RUN(k_alignbit, "dependent v_alignbit_b32")
  RUN(k_add3, "dependent v_add3_u32")
  RUN(k_xad, "dependent v_xad_u32")
  RUN(k_round, "Threefry round (add, alignbit, xor)")
  RUN(k_round_rot_first, "Threefry round (alignbit, add, xor)")
  RUN(k_two_chains, "PAIR of independent v_add_u32")
  RUN(k_three_chains, "TRIPLE of independent v_add_u32")
  RUN(k_salu, "dependent s_add_u32")
  RUN(k_valu_salu, "dependent v_add_u32 + independent s_add_u32")
  RUN(k_dpp, "dependent s_nop 1 + v_mov_b32_dpp")
  return 0;
}
