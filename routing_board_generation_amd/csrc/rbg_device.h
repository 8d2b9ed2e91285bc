// rbg_device.h -- device-side building blocks shared by every kernel of librbg_prw.so:
// jax's Threefry-2x32 stream, sub-wave helpers, the padded LDS board.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rbg {

constexpr int kWaveSize = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kBlock = kWaveSize * kWavesPerBlock;
constexpr uint32_t kWall = 0xFFFFFFFFu;
constexpr int kMaxM = 40;
constexpr int kMaxN = 64;

// ---------------------------------------------------------------------------------------------
// Threefry-2x32-20 ([3P] jax/_src/prng.py::_threefry2x32_lowering)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rotl(uint32_t x, int r) {
  return __builtin_rotateleft32(x, r);  // v_alignbit_b32
}

#define RBG_TF_ROUND(r) \
  x0 += x1;             \
  x1 = rotl(x1, r) ^ x0;

// The 20 rounds are split in two halves so that the walk loop can place each half in the shadow
// of a different LDS round trip (see rbg_walk.h).
struct TfState {
  uint32_t x0, x1, k0, k1, k2;
};

__device__ __forceinline__ TfState threefry_first_half(uint32_t k0, uint32_t k1, uint32_t x0,
                                                       uint32_t x1) {
  const uint32_t k2 = k0 ^ k1 ^ 0x1BD11BDAu;
  x0 += k0;
  x1 += k1;
  RBG_TF_ROUND(13) RBG_TF_ROUND(15) RBG_TF_ROUND(26) RBG_TF_ROUND(6)
  x0 += k1;
  x1 += k2 + 1u;
  RBG_TF_ROUND(17) RBG_TF_ROUND(29) RBG_TF_ROUND(16) RBG_TF_ROUND(24)
  x0 += k2;
  x1 += k0 + 2u;
  TfState t = {x0, x1, k0, k1, k2};
  return t;
}

__device__ __forceinline__ void threefry_second_half(const TfState &t, uint32_t &y0,
                                                     uint32_t &y1) {
  uint32_t x0 = t.x0, x1 = t.x1;
  const uint32_t k0 = t.k0, k1 = t.k1, k2 = t.k2;
  RBG_TF_ROUND(13) RBG_TF_ROUND(15) RBG_TF_ROUND(26) RBG_TF_ROUND(6)
  x0 += k0;
  x1 += k1 + 3u;
  RBG_TF_ROUND(17) RBG_TF_ROUND(29) RBG_TF_ROUND(16) RBG_TF_ROUND(24)
  x0 += k1;
  x1 += k2 + 4u;
  RBG_TF_ROUND(13) RBG_TF_ROUND(15) RBG_TF_ROUND(26) RBG_TF_ROUND(6)
  x0 += k2;
  x1 += k0 + 5u;
  y0 = x0;
  y1 = x1;
}

__device__ __forceinline__ void threefry(uint32_t k0, uint32_t k1, uint32_t x0, uint32_t x1,
                                         uint32_t &y0, uint32_t &y1) {
  const TfState t = threefry_first_half(k0, k1, x0, x1);
  threefry_second_half(t, y0, y1);
}

// Out-of-line copy for the once-per-board code (key derivation, sort keys, pipeline prologue):
// that code is straight-line and executed once, so its cost is instruction FETCH, not issue.
__device__ __noinline__ uint2 threefry_call(uint32_t k0, uint32_t k1, uint32_t x0, uint32_t x1) {
  uint32_t y0, y1;
  threefry(k0, k1, x0, x1, y0, y1);
  return make_uint2(y0, y1);
}

struct Key2 {
  uint32_t a, b;
};

// Word w of random_bits(key, total) with jax's layout: counters iota(total) (+ one 0 pad word if
// total is odd) split in halves; block j hashes (j, half + j); out = y0-half ++ y1-half.
__device__ __forceinline__ uint32_t random_bits_word(Key2 k, uint32_t w, uint32_t total) {
  const uint32_t half = (total + 1u) >> 1;
  const bool hi = w >= half;
  const uint32_t j = hi ? w - half : w;
  const uint32_t c1 = (half + j < total) ? half + j : 0u;
  uint32_t y0, y1;
  threefry(k.a, k.b, j, c1, y0, y1);
  return hi ? y1 : y0;
}

// Child i of split(key, num): words (2i, 2i+1) of random_bits(key, 2*num).
__device__ __forceinline__ Key2 split_child(Key2 k, uint32_t i, uint32_t num) {
  Key2 r;
  r.a = random_bits_word(k, 2u * i, 2u * num);
  r.b = random_bits_word(k, 2u * i + 1u, 2u * num);
  return r;
}

// [3P] jax/_src/random.py::_uniform (float32): mantissa fill, subtract 1.
__device__ __forceinline__ float bits_to_uniform(uint32_t bits) {
  return __fsub_rn(__uint_as_float((bits >> 9) | 0x3F800000u), 1.0f);
}

// The mantissa fill alone, f = 1 + uniform in [1, 2).  1 - uniform == 2 - f exactly: f - 1, 1 - x
// and 2 - f are all exact in float32 for f a multiple of 2^-23 in [1, 2).
__device__ __forceinline__ float bits_to_unit12(uint32_t bits) {
  return __uint_as_float((bits >> 9) | 0x3F800000u);
}
// 1 - uniform(bits), exactly (see above): what jax's choice multiplies cum[-1] by
__device__ __forceinline__ float bits_to_one_minus_uniform(uint32_t bits) {
  return __fsub_rn(2.0f, bits_to_unit12(bits));
}

// ---------------------------------------------------------------------------------------------
// Wave / sub-wave helpers
// ---------------------------------------------------------------------------------------------
// LDS traffic of ONE wave is processed in issue order, so a later ds_read observes an earlier
// ds_write of another lane of the same wave; this only has to stop the compiler from reordering.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A wavefront is cut into 64/LPB sub-waves of LPB consecutive lanes; each sub-wave owns one board.
// LPB = 64 is the plain one-board-per-wave case.  Every cross-lane primitive of the walk goes
// through this struct so that the same code serves all three layouts.
template <int LPB>
struct SubWave {
  static_assert(LPB == 16 || LPB == 32 || LPB == 64, "LPB must be 16, 32 or 64");
  static constexpr int kLanes = LPB;
  static constexpr int kSubs = kWaveSize / LPB;
  int lane;  // 0..63
  int l;     // lane inside the sub-wave
  int h;     // sub-wave index inside the wave

  __device__ __forceinline__ explicit SubWave(int lane_)
      : lane(lane_), l(lane_ & (LPB - 1)), h(lane_ / LPB) {}

  // ballot over this lane's sub-wave; bit i belongs to sub-lane i
  __device__ __forceinline__ uint64_t ballot(bool p) const {
    const uint64_t m = __ballot(p);
    if (LPB == kWaveSize) return m;
    return (m >> (h * LPB)) & ((1ull << (LPB & 63)) - 1ull);
  }

  // value of `v` in sub-lane `src` (wave-uniform, < LPB) of this lane's sub-wave
  __device__ __forceinline__ uint32_t bcast(uint32_t v, int src) const {
    uint32_t r = __builtin_amdgcn_readlane(v, src);
#pragma unroll
    for (int s = 1; s < kSubs; ++s) {
      const uint32_t t = __builtin_amdgcn_readlane(v, s * LPB + src);
      r = h == s ? t : r;
    }
    return r;
  }

  __device__ __forceinline__ uint64_t min_u64(uint64_t v) const {
#pragma unroll
    for (int off = LPB / 2; off >= 1; off >>= 1) {
      const uint32_t lo = __shfl_xor((uint32_t)v, off);
      const uint32_t hi = __shfl_xor((uint32_t)(v >> 32), off);
      const uint64_t o = ((uint64_t)hi << 32) | lo;
      v = o < v ? o : v;
    }
    return v;
  }

  __device__ __forceinline__ int max_i32(int v) const {
#pragma unroll
    for (int off = LPB / 2; off >= 1; off >>= 1) {
      const int o = __shfl_xor(v, off);
      v = o > v ? o : v;
    }
    return v;
  }

};

// split(key) with the two Threefry blocks in sub-lanes 0 and 1 of ONE evaluation:
// children (y0[0], y0[1]) and (y1[0], y1[1]) of blocks (0, 2) and (1, 3).
template <int LPB>
__device__ __forceinline__ void split2_lanes(const SubWave<LPB> &sw, Key2 k, Key2 &child0,
                                             Key2 &child1) {
  const uint32_t x = (uint32_t)(sw.l & 1);
  const uint2 y = threefry_call(k.a, k.b, x, x + 2u);
  child0.a = sw.bcast(y.x, 0);
  child0.b = sw.bcast(y.x, 1);
  child1.a = sw.bcast(y.y, 0);
  child1.b = sw.bcast(y.y, 1);
}

// ---------------------------------------------------------------------------------------------
// Padded board geometry
// ---------------------------------------------------------------------------------------------
struct Geo {
  int m;        // grid side
  int n;        // m * m
  int S;        // padded row stride = m + 4 (4 wall columns on the left of every row; they also
                // serve as the right wall of the previous row)
  int words;    // (m + 4) * S: 2 wall rows above and below
  uint32_t magic;  // ceil(2^20 / m): idx / m == (idx * magic) >> 20 for idx * m < 2^20
  uint32_t magic_s;  // ceil(2^20 / S): the same for the padded stride
  uint32_t step_table;  // the four walk steps as signed bytes: [-S, +S, -1, +1] (up,down,left,right)
  uint32_t thr;    // shuffle_prefix candidate threshold for this launch's count (0 = slow path)
};

__device__ __forceinline__ int pad_index(const Geo &g, int r, int c) {
  return (r + 2) * g.S + 4 + c;
}
__device__ __forceinline__ int div_m(const Geo &g, int idx) {
  return (int)(((uint32_t)idx * g.magic) >> 20);
}
// padded index -> (row, col)
__device__ __forceinline__ void unpad_index(const Geo &g, int P, int &r, int &c) {
  const int pr = (int)(((uint32_t)P * g.magic_s) >> 20);
  r = pr - 2;
  c = P - pr * g.S - 4;
}
__device__ __forceinline__ int board_words(const Geo &g) { return (g.words + 3) & ~3; }

// Per-board scratch next to the board: Threefry pipeline words (2N <= 2*LPB) or the candidate
// list of shuffle_prefix (2*LPB words), followed by 2*LPB dump words: lanes that have nothing to
// store park their branch-free stores there, each lane in its OWN two words (same-address LDS
// stores and atomics from many lanes serialise).
template <int LPB>
__device__ __host__ constexpr int scratch_words() {
  return 4 * LPB;
}
template <int LPB>
__device__ __forceinline__ int slab_words(const Geo &g) {  // one board + its scratch
  return board_words(g) + scratch_words<LPB>();
}

// walls everywhere, zeros inside
template <int LPB>
__device__ __forceinline__ void board_clear(uint32_t *board, const Geo &g, int l) {
  if ((g.m & 3) == 0) {
    // S = m + 4 and words = S * S are multiples of 4 and rows start 16-byte aligned: 128-bit stores
    const uint4 wall = make_uint4(kWall, kWall, kWall, kWall), zero = make_uint4(0u, 0u, 0u, 0u);
    for (int w = l * 4; w < g.words; w += LPB * 4) *reinterpret_cast<uint4 *>(&board[w]) = wall;
    wave_lds_fence();
    const int per_row = g.m >> 2;  // 4-cell groups per row
    for (int q = l; q < per_row * g.m; q += LPB) {
      const int r = div_m(g, q * 4);
      *reinterpret_cast<uint4 *>(&board[pad_index(g, r, (q - r * per_row) * 4)]) = zero;
    }
    wave_lds_fence();
    return;
  }
  for (int w = l; w < g.words; w += LPB) board[w] = kWall;
  wave_lds_fence();
  for (int idx = l; idx < g.n; idx += LPB) {
    const int r = div_m(g, idx);
    board[pad_index(g, r, idx - r * g.m)] = 0u;
  }
  wave_lds_fence();
}

}  // namespace rbg
