// rbg_seqwalk.h -- SequentialRandomWalkBoard on one wavefront per board (BASELINE configs[4], the
// "divergent-path stress").
//
// Reference: JAXI/sequential_random_walk.py:34-431 (pick_start, available_cells, one_step,
// walk_randomly, add_agents, generate, generate_starts_ends).  The algorithm is strictly serial
// per board -- N wires one after the other, each up to max_length dependent steps, the whole
// board retried with a shorter max_length (same key) when a wire cannot start or move -- so the
// only parallelism inside a board is across the candidates of one draw:
//   * pick_start: jax's choice(replace=False, p) is a Gumbel top-1 = the empty cell with the
//     largest uniform (bits >> 9), lowest index on ties; the m*m sort keys come from
//     random_bits(key, m*m), spread over the lanes, winner by a wave-wide 64-bit max;
//   * one_step: the same draw over [up, down, left, right] (+ padding to rows + 1 entries, which
//     only matters for the counter layout of random_bits); lanes 0..3 evaluate one candidate each
//     (stencil + Threefry), the winner is picked on scalar registers.
#pragma once
#include "rbg_device.h"

namespace rbg {

__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const uint32_t lo = __shfl_xor((uint32_t)v, off);
    const uint32_t hi = __shfl_xor((uint32_t)(v >> 32), off);
    const uint64_t o = ((uint64_t)hi << 32) | lo;
    v = o > v ? o : v;
  }
  return v;
}

// generate(key) + generate_starts_ends: returns the attempt that succeeded (1 .. 2m) or 0 when
// all failed (board all zero).  On return the LDS board holds the solved board; lanes w < N hold
// the first / last cell (flat index) of wire w, (0, 0) for every wire after a total failure.
//
// Retries without redundant work.  Every attempt restarts from the SAME key, so an attempt with
// limit L - 1 replays the attempt with limit L exactly until some wire would have taken its L-th
// step.  Hence (a) an attempt can stop at its first failing wire (the board is discarded), and
// (b) if the wires before that one took at most s_max steps, every limit >= s_max fails the same
// way: the next attempt that can differ has limit s_max - 1.  The reported attempt number is the
// one the reference's loop would have reached.
//
// Per step ONE Threefry evaluation serves two stages: lanes 0..3 draw the candidates' uniforms
// with this step's subkey while lanes 4,5 already split the next key (the key chain does not
// depend on the board; a step that turns out impossible simply does not commit the new key).
__device__ __forceinline__ int seq_rw_generate(uint32_t *board, const Geo &g, int N, Key2 key0,
                                               int lane, int &first_cell, int &last_cell) {
  const SubWave<64> sw(lane);
  const int S = g.S, m = g.m;
  const uint32_t n = (uint32_t)g.n;
  const uint32_t half_n = (n + 1u) >> 1;
  const uint32_t total_step = (uint32_t)m + 1u;  // available_cells is padded to rows + 1 entries
  const uint32_t half_step = (total_step + 1u) >> 1;
  const int dP = lane == 0 ? -S : lane == 1 ? S : lane == 2 ? -1 : 1;  // up, down, left, right
  // lane roles of the per-step evaluation: 0..3 candidate j's word of random_bits(subkey, m + 1)
  // (y0 of block j, or y1 of block j - half for tiny grids); 4,5 blocks (0,2),(1,3) of split(key)
  const uint32_t cj = (uint32_t)(lane & 3);
  const bool cand_hi = cj >= half_step;
  const uint32_t cblk = cand_hi ? cj - half_step : cj;
  const uint32_t c0 = lane < 4 ? cblk : (uint32_t)(lane & 1);
  const uint32_t c1 = lane < 4 ? (half_step + cblk < total_step ? half_step + cblk : 0u)
                               : (uint32_t)(lane & 1) + 2u;

  int limit = 2 * m - 1;
  while (limit >= 0) {
    const int attempt = 2 * m - limit;
    board_clear<64>(board, g, lane);
    Key2 key = key0;  // every retry starts from the same key (:347-355)
    bool success = true;
    int s_max = 0;  // most steps taken by a wire of this attempt
    first_cell = last_cell = 0;
    for (int w = 0; w < N && success; ++w) {
      const uint32_t head = 3u * (uint32_t)w + 2u, path = head - 1u, tail = head + 1u;
      // add_agents: key, subkey = split(key) (the new key is dropped); pick_start(subkey):
      // key, subkey = split(key) -- the tuple carries THAT key from here on
      Key2 dropped, sub, sub2;
      split2_lanes(sw, key, dropped, sub);
      split2_lanes(sw, sub, key, sub2);
      // ---- pick_start: empty cell with the largest uniform, lowest index on ties ------------
      uint64_t best = 0ull;  // 0 = no empty cell
      for (uint32_t blk = lane; blk < half_n; blk += kWaveSize) {
        const uint32_t j_hi = half_n + blk;
        const uint2 y = threefry_call(sub2.a, sub2.b, blk, j_hi < n ? j_hi : 0u);
        const int r0 = div_m(g, (int)blk);
        if (board[pad_index(g, r0, (int)blk - r0 * m)] == 0u) {
          const uint64_t c = ((uint64_t)(y.x >> 9) << 32) | (0xFFFFFFFFu - blk);
          best = c > best ? c : best;
        }
        if (j_hi < n) {
          const int r1 = div_m(g, (int)j_hi);
          if (board[pad_index(g, r1, (int)j_hi - r1 * m)] == 0u) {
            const uint64_t c = ((uint64_t)(y.y >> 9) << 32) | (0xFFFFFFFFu - j_hi);
            best = c > best ? c : best;
          }
        }
      }
      best = wave_max_u64(best);
      if (best == 0ull) {  // can_start == False
        success = false;
        break;
      }
      const int start = (int)(0xFFFFFFFFu - (uint32_t)best);
      const int sr = div_m(g, start);
      int P = pad_index(g, sr, start - sr * m);
      const int P0 = P;
      if (lane == 0) board[P] = head;
      wave_lds_fence();
      // ---- walk_randomly: up to `limit` steps ------------------------------------------------
      // pipeline prologue: (key_next, step_key) = split(key)
      Key2 key_next, step_key;
      split2_lanes(sw, key, key_next, step_key);
      int steps = 0;
      for (; steps < limit; ++steps) {
        // lane d < 4 evaluates candidate d: empty, and touching this wire at most once
        const int Q = P + dP;
        const uint32_t vq = board[Q];
        const uint32_t a = board[Q - S], b = board[Q + S], c = board[Q - 1], d = board[Q + 1];
        // one evaluation: candidates' uniforms (subkey of THIS step) + split of the NEXT key
        const uint2 y = threefry_call(lane < 4 ? step_key.a : key_next.a,
                                      lane < 4 ? step_key.b : key_next.b, c0, c1);
        const int touching = (int)((uint32_t)(a - path) < 3u) + (int)((uint32_t)(b - path) < 3u) +
                             (int)((uint32_t)(c - path) < 3u) + (int)((uint32_t)(d - path) < 3u);
        const bool avail = lane < 4 && vq == 0u && touching <= 1;
        const uint32_t amask = (uint32_t)__ballot(avail) & 0xFu;
        if (amask == 0u) break;  // can_step == False now and for every later trip of the loop
        // one_step: key, subkey = split(key) is committed; choice over the padded candidates
        key = key_next;
        key_next.a = __builtin_amdgcn_readlane(y.x, 4);
        key_next.b = __builtin_amdgcn_readlane(y.x, 5);
        step_key.a = __builtin_amdgcn_readlane(y.y, 4);
        step_key.b = __builtin_amdgcn_readlane(y.y, 5);
        const uint32_t u = (cand_hi ? y.y : y.x) >> 9;
        int pick = -1;
        uint32_t pick_u = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t uk = __builtin_amdgcn_readlane(u, k);
          if (((amask >> k) & 1u) && (pick < 0 || uk > pick_u)) {
            pick = k;
            pick_u = uk;
          }
        }
        const int stepP = pick == 0 ? -S : pick == 1 ? S : pick == 2 ? -1 : 1;
        if (lane == 0) {
          board[P + stepP] = tail;  // the new cell is the wire's TARGET ...
          board[P] = path;          // ... the old one becomes PATH
        }
        P += stepP;
        wave_lds_fence();
      }
      if (lane == 0) board[P0] = head;  // the first cell becomes the head again (:286)
      wave_lds_fence();
      if (steps == 0) {  // moved == False
        success = false;
        break;
      }
      s_max = steps > s_max ? steps : s_max;
      if (lane == w) {
        int r, c;
        unpad_index(g, P0, r, c);
        first_cell = r * m + c;
        unpad_index(g, P, r, c);
        last_cell = r * m + c;
      }
    }
    if (success) return attempt;
    // the wires before the failing one replay identically while limit >= s_max
    limit = (s_max - 1 < limit - 1) ? s_max - 1 : limit - 1;
    if (limit < 1) break;  // limit 0 lets no wire move: every remaining attempt fails
  }
  board_clear<64>(board, g, lane);
  first_cell = last_cell = 0;
  return 0;
}

}  // namespace rbg
