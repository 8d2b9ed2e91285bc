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
//     (stencil + Threefry), the winner is a DPP max over that quad.
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
// step.  Hence (a) an attempt can stop at its first failing wire (the board is discarded), (b)
// if the wires before that one took at most s_max steps, every limit >= s_max fails the same
// way: the next attempt that can differ has limit s_max - 1, and (c) that attempt need not
// replay the common prefix either (see "Resuming" below).  The reported attempt number is the
// one the reference's loop would have reached.
//
// Per step ONE Threefry evaluation serves two stages: lanes 0..3 draw the candidates' uniforms
// with this step's subkey while lanes 4,5 already split the next key (the key chain does not
// depend on the board; a step that turns out impossible simply does not commit the new key).
__device__ __forceinline__ int seq_rw_generate(uint32_t *board, uint32_t *hist, const Geo &g,
                                               int N, Key2 key0, int lane, int &first_cell,
                                               int &last_cell) {
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

  // Resuming instead of replaying.  Attempt L - 1 replays attempt L up to the first wire w* that
  // took more than L - 1 steps, and there it only stops one step earlier: the state to continue
  // from is the failed attempt's own state with the wires after w* erased and w*'s LAST step
  // undone (its last cell emptied, the cell before it the TARGET again, the key the one it had
  // before that step's split).  `hist` (when the launch could afford the LDS: N * (2m - 1)
  // entries of 16 bytes per board) keeps (key, cell) before every committed step of every wire, so
  // a wire can be cut again and again; without it lane w keeps the record of wire w's last step
  // only, a wire that has already been cut once cannot be cut again, and then the attempt replays
  // from the start as the reference does.
  const int H = 2 * m - 1;  // the longest walk
  int rec_steps = 0, rec_last = 0, rec_prev = 0;
  Key2 rec_key = {0u, 0u};
  bool rec_valid = false;
  int limit = 2 * m - 1, w_begin = 0, s_max = 0;
  Key2 key = key0;
  bool fresh = true;
  while (limit >= 0) {
    const int attempt = 2 * m - limit;
    if (fresh) {
      board_clear<64>(board, g, lane);
      key = key0;  // every retry starts from the same key (:347-355)
      s_max = 0;   // most steps taken by a wire of this attempt
      w_begin = 0;
      first_cell = last_cell = 0;
    }
    bool success = true;
    int w = w_begin;
    for (; w < N; ++w) {
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
      best = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(best >> 32)) << 32) |
             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)best);  // uniform, and known to be
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
      int steps = 0, prevP = P;
      Key2 key_before = key;  // the key before the last committed step
      for (; steps < limit; ++steps) {
        // lane d < 4 evaluates candidate d: empty, and touching this wire at most once (the reads
        // of the other lanes and of wall candidates stay inside the two wall rows / columns)
        const uint32_t *q = board + P + dP;
        const uint32_t vq = q[0], a = q[-S], b = q[S], c = q[-1], d = q[1];
        __builtin_amdgcn_sched_barrier(0);  // all five reads in flight under the Threefry rounds
        // one evaluation: candidates' uniforms (subkey of THIS step) + split of the NEXT key
        uint32_t y0, y1;
        threefry(lane < 4 ? step_key.a : key_next.a, lane < 4 ? step_key.b : key_next.b, c0, c1,
                 y0, y1);
        const int touching = (int)((uint32_t)(a - path) < 3u) + (int)((uint32_t)(b - path) < 3u) +
                             (int)((uint32_t)(c - path) < 3u) + (int)((uint32_t)(d - path) < 3u);
        const bool avail = lane < 4 && vq == 0u && touching <= 1;
        // choice over the padded candidates = the available one with the largest uniform, the
        // lowest index on ties: a max over quad 0 of (uniform, 3 - index) + 1, 0 = not available
        const uint32_t u = (cand_hi ? y1 : y0) >> 9;
        uint32_t score = avail ? ((u << 2) | (3u - (uint32_t)lane)) + 1u : 0u;
        {
          const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)score, 0xB1, 0xF, 0xF, false);
          score = o > score ? o : score;
          const uint32_t o2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)score, 0x4E, 0xF, 0xF, false);
          score = o2 > score ? o2 : score;
        }
        const uint32_t best = (uint32_t)__builtin_amdgcn_readfirstlane((int)score);
        if (best == 0u) break;  // can_step == False now and for every later trip of the loop
        // one_step: key, subkey = split(key) is committed
        key_before = key;
        if (hist != nullptr && lane == 0)
          *reinterpret_cast<uint4 *>(&hist[(w * H + steps) * 4]) =
              make_uint4(key.a, key.b, (uint32_t)P, 0u);
        key = key_next;
        key_next.a = __builtin_amdgcn_readlane(y0, 4);
        key_next.b = __builtin_amdgcn_readlane(y0, 5);
        step_key.a = __builtin_amdgcn_readlane(y1, 4);
        step_key.b = __builtin_amdgcn_readlane(y1, 5);
        const int pick = 3 - (int)((best - 1u) & 3u);
        const int stepP = __builtin_amdgcn_sbfe((int)g.step_table, (uint32_t)pick * 8u, 8u);
        if (lane == 0) {
          board[P + stepP] = tail;  // the new cell is the wire's TARGET ...
          board[P] = path;          // ... the old one becomes PATH
        }
        prevP = P;
        P += stepP;
        wave_lds_fence();
      }
      if (lane == 0) board[P0] = head;  // the first cell becomes the head again (:286)
      wave_lds_fence();
      if (steps == 0) {  // moved == False
        success = false;
        break;
      }
      s_max = __builtin_amdgcn_readfirstlane(steps > s_max ? steps : s_max);
      if (lane == w) {
        int r, c;
        unpad_index(g, P0, r, c);
        first_cell = r * m + c;
        unpad_index(g, P, r, c);
        last_cell = r * m + c;
        rec_steps = steps;
        rec_last = P;
        rec_prev = prevP;
        rec_key = key_before;
        rec_valid = steps >= 2;
      }
    }
    if (success) return attempt;
    // `w` is the wire that failed; the wires before it replay identically while limit >= s_max
    limit = __builtin_amdgcn_readfirstlane(s_max - 1);  // (s_max <= the old limit)
    if (limit < 1) break;  // limit 0 lets no wire move: every remaining attempt fails
    // the first wire that took s_max steps is the one the next attempt cuts
    const int ws = __builtin_amdgcn_readfirstlane(
        (int)__ffsll((long long)__ballot(lane < w && rec_steps == s_max)) - 1);
    int lastP = 0, prevP = 0;
    fresh = ws < 0;
    if (!fresh) {
      lastP = __builtin_amdgcn_readlane(rec_last, ws);
      if (hist != nullptr) {  // wire ws took s_max steps: the state before its last one
        const uint4 e = *reinterpret_cast<const uint4 *>(&hist[(ws * H + s_max - 1) * 4]);
        key.a = __builtin_amdgcn_readfirstlane(e.x);
        key.b = __builtin_amdgcn_readfirstlane(e.y);
        prevP = __builtin_amdgcn_readfirstlane((int)e.z);
      } else if (__builtin_amdgcn_readlane((int)rec_valid, ws) != 0) {
        prevP = __builtin_amdgcn_readlane(rec_prev, ws);
        key.a = (uint32_t)__builtin_amdgcn_readlane((int)rec_key.a, ws);
        key.b = (uint32_t)__builtin_amdgcn_readlane((int)rec_key.b, ws);
      } else {
        fresh = true;
      }
    }
    if (!fresh) {
      const uint32_t first_gone = 3u * (uint32_t)(ws + 1) + 1u;  // smallest value of a later wire
      for (int idx = lane; idx < g.n; idx += kWaveSize) {
        const int r = div_m(g, idx);
        const int p = pad_index(g, r, idx - r * m);
        if (board[p] >= first_gone) board[p] = 0u;
      }
      wave_lds_fence();
      if (lane == 0) {
        board[lastP] = 0u;                              // the step that is taken back ...
        board[prevP] = 3u * (uint32_t)ws + 3u;          // ... leaves the TARGET one cell earlier
      }
      wave_lds_fence();
      if (lane == ws) {
        int r, c;
        unpad_index(g, prevP, r, c);
        last_cell = r * m + c;
        rec_steps = limit;
        rec_last = prevP;
        rec_valid = false;
      }
      w_begin = ws + 1;
      s_max = limit;  // wire ws has `limit` steps now, the wires before it fewer than s_max
    }
  }
  board_clear<64>(board, g, lane);
  first_cell = last_cell = 0;
  return 0;
}

}  // namespace rbg
