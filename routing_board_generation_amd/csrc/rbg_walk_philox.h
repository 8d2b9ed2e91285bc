// rbg_walk_philox.h -- the opt-in `stream = "philox"` variant of the ParallelRandomWalk generator
// (SURVEY.md 8a-PRNG).  The walk is the reference's (rbg_walk.h: available cells, jax's weighted
// choice, one ds_max_u32 per mover for the collision rule); only the randomness differs, and that
// makes it a DIFFERENT random process from the reference's: it can only be checked against this
// project's own CPU restatement of the stream, which include/rbg_prw.h specifies (Philox-4x32-10,
// counter (c0, c1, c2, tag): tag 1 start-cell sort keys, tag 2 walk uniforms, tag 3 State.key).
//
// What the layout buys on this chip.  jax's stream costs 2N + 2 Threefry blocks per iteration
// and board (split(k, N), N uniforms, the key chain), which is what sizes the default path's
// sub-waves (2N + 2 lanes) and its three-stage lane pipeline.  Here agent i's uniform at
// iteration t is word t % 4 of philox(key, (t / 4, i, 0, 2)): one evaluation per agent lane every
// FOUR iterations, no chain, no LDS scratch -- so a board needs only N lanes (up to four 16x16/10
// boards per wavefront instead of two; the host packs two when the batch is small, see
// rbg_prw_generate_philox) and an iteration ~100 instructions instead of 156.
#pragma once
#include "rbg_walk.h"

namespace rbg {

struct Philox4 {
  uint32_t w[4];
};

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0,
                                                 uint32_t c1, uint32_t c2, uint32_t c3) {
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
  Philox4 p = {{c0, c1, c2, c3}};
  return p;
}

// generate_board for the board of this lane's sub-wave; same contract as prw_walk_impl.
template <int LPB>
__device__ __forceinline__ void prw_walk_philox(const SubWave<LPB> &sw, uint32_t *board,
                                                uint32_t *W, const Geo &g, int N, Key2 key,
                                                Walker &w) {
  const int l = sw.l;
  const bool is_agent = l < N;
  // ---- start cells: the N smallest (sort key, cell) pairs, ascending (tag 1) -----------------
  for (int q = l; q * 4 < g.n; q += LPB) {
    const Philox4 p = philox4x32_10(key.a, key.b, (uint32_t)q, 0u, 0u, 1u);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (q * 4 + k < g.n) board[q * 4 + k] = p.w[k];
  }
  wave_lds_fence();
  int sel0, sel1;
  shuffle_prefix(sw, board, W, g, key, N, g.thr, sel0, sel1, /*keys_ready=*/true);
  w.iters = 0;
  board_clear<LPB>(board, g, l);
  const uint32_t wire = 3u * (uint32_t)l + 1u;
  w.r0 = w.c0 = w.r = w.c = 0;
  {  // lanes without an agent park on a word of the wall strip (rbg_walk.h)
    const int q = div_m(g, l);
    w.P = pad_index(g, l - q * g.m, -3 + (q & 1));
  }
  if (is_agent) {
    w.r0 = div_m(g, sel0);
    w.c0 = sel0 - w.r0 * g.m;
    w.r = w.r0;
    w.c = w.c0;
    w.P = pad_index(g, w.r, w.c);
    board[w.P] = wire;
  }
  wave_lds_fence();

  // ---- the walk (tag 2): four iterations per Philox evaluation -------------------------------
  const int S = g.S;
  int last_live = 0;
  bool done = false;  // wave-uniform
  for (int t4 = 0; t4 * 4 < g.n && !done; ++t4) {
    const Philox4 p = philox4x32_10(key.a, key.b, (uint32_t)t4, (uint32_t)l, 0u, 2u);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (done) break;
      const int t = t4 * 4 + k;
      const float u = bits_to_unit12(p.w[k]);  // 1 + uniform
      const int P = w.P;
      const uint32_t *mid = board + P, *up = mid - S, *dn = mid + S, *up2 = up - S, *dn2 = dn + S;
      const uint32_t vu = up[0], vd = dn[0], vl = mid[-1], vr = mid[1];
      const uint32_t uu = up2[0], dd = dn2[0];
      const uint32_t ll = mid[-2], rr = mid[2];
      const uint32_t ul = up[-1], ur = up[1];
      const uint32_t dl = dn[-1], dr = dn[1];
      const bool own_ul = ul == wire, own_ur = ur == wire, own_dl = dl == wire, own_dr = dr == wire;
      const bool a_up = vu == 0u && !(uu == wire || own_ul || own_ur);
      const bool a_dn = vd == 0u && !(dd == wire || own_dl || own_dr);
      const bool a_lf = vl == 0u && !(ll == wire || own_ul || own_dl);
      const bool a_rt = vr == 0u && !(rr == wire || own_ur || own_dr);
      const float c1 = a_up ? 1.0f : 0.0f, c2 = c1 + (a_dn ? 1.0f : 0.0f);
      const float c3 = c2 + (a_lf ? 1.0f : 0.0f), c4 = c3 + (a_rt ? 1.0f : 0.0f);
      const bool mover = c4 > 0.0f;
      last_live = mover ? t + 1 : last_live;
      if (__ballot(mover) == 0ull) {
        done = true;
        break;
      }
      const float rsel = __fmul_rn(c4, __fsub_rn(2.0f, u));  // cum[-1] * (1 - uniform)
      int dir = 3;
      if (c3 >= rsel) dir = 2;
      if (c2 >= rsel) dir = 1;
      if (c1 >= rsel) dir = 0;
      const int dP = __builtin_amdgcn_sbfe((int)g.step_table, (uint32_t)dir * 8u, 8u);
      const int Q = mover ? P + dP : P;
      atomicMax(&board[Q], wire);  // contested cell: the highest id wins, losers stay
      wave_lds_fence();
      const uint32_t after = board[Q];
      w.P = after == wire ? Q : P;
      wave_lds_fence();
    }
  }
  w.iters = sw.max_i32(last_live);
  unpad_index(g, w.P, w.r, w.c);
  if (is_agent) board[pad_index(g, w.r0, w.c0)] = wire + 1u;  // heads first ...
  wave_lds_fence();
  if (is_agent) board[w.P] = wire + 2u;  // ... then targets (a never-moved agent is a lone target)
  wave_lds_fence();
}

}  // namespace rbg
