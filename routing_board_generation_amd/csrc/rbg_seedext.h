// rbg_seedext.h -- SeedExtensionBoard, one board per sub-wave of 16/32/64 lanes (SURVEY.md 8f rank
// 4b, BASELINE configs[4]).
//
// Reference (JAXI = routing_board_generation/board_generation_methods/jax_implementation):
//   JAXI/board_generation/seed_extension.py:93-147 (return_seeded_board), :149-227
//   (return_solved_board), :257-304 (generate_starts_ends);
//   JAXI/utils/post_processor_utils_jax.py:24-195 (extend_wires_jax) + helpers :200-429;
//   JAXI/utils/grid_utils.py:59-178, 187-224, 242-260, 299-346, 491-612 (optimise_wire).
//
// What the reference does per board: seed N adjacent (target, head) pairs on a checkerboard of
// candidate cells; then sweep the board cell by cell, row-major in a randomly flipped frame,
// over and over until an un-flipped sweep changes nothing: every head/target cell met by the
// sweep moves one cell on (straight ahead if it can, else a random free neighbour that does not
// touch its own wire); finally re-route every wire along a BFS-shortest path.  The PRNG is
// threaded through EVERY cell of every sweep: key_{c+1} = split(key_c)[0], the cell's random
// direction comes from randint(split(key_{c+j})[1], 4) for the tries j = 0, 1, ... and its
// "ignore the straight direction" draw from uniform(split(key_c)[1]).
//
// Formulation here.  That key chain does not depend on the board, so a sweep is cut in two:
//   phase A  the chain K_0 .. K_{n+margin} of the sweep, one Threefry evaluation per link, with the
//            two dependent levels (split of the choice key, the final random_bits word) riding in
//            otherwise idle lanes of the same evaluation two and three links behind (lane roles
//            below; operands travel by DPP inside the 16-lane row, no LDS on the chain).  It
//            leaves one byte per chain position in LDS: the direction number a try at that
//            position would draw, plus the use_random bit per cell.
//   phase B  event driven: instead of visiting m*m cells, the sub-wave jumps from one head/target
//            to the next in sweep order (a min-reduction over the endpoint positions the lanes
//            hold: lane w owns wire w) and applies the reference's rule to it with a 16-read
//            stencil on the padded LDS board.
// A sweep costs (m*m + margin) Threefry evaluations whatever the board holds, so the board with
// the most sweeps bounds the launch; boards of one wave run in lock-step and idle once done.
// Only sweeps that can change the board are run: a board whose endpoints are all stuck is
// finished, although the reference keeps sweeping it until an un-flipped sweep comes up (the
// loop in seed_extension_generate says why that is output-equivalent).
// optimise_wire is a level-synchronous BFS whose FIFO order -- hence the reference's tie-breaking
// between equally short paths -- is kept exactly: all (node, direction) pairs of a level claim
// their neighbour with ds_min_u32(key = FIFO position * 4 + direction number); the smallest key
// is the pop that would have come first.
#pragma once
#include "rbg_device.h"

namespace rbg {

#ifndef RBG_SE_MARGIN
#define RBG_SE_MARGIN 8  // chain positions computed past the last cell (tries that spill over)
#endif
constexpr int kSeMargin = RBG_SE_MARGIN;

struct SeParams {
  int N;
  int iterations;     // extension_iterations
  int max_steps;      // sweeps allowed per extend_wires_jax call (extension_steps)
  int two_sided;
  int rnd_mode;       // use_random = randomness > uniform: 0 never, 1 always, 2 draw
  float randomness;
  int bfs_last;       // run optimise_wire after the last extension too (solved board wanted)
  int half;           // m / 2: candidate cells per side
  int seed_only;      // return_seeded_board(key): the key IS the seed key, nothing else runs
  int extra_words;    // se_extra_words() as the host sized the launch's LDS with
};

// extra LDS words per board after the slab of rbg_device.h:
//   extension: idx bytes [4 + n + margin] + rnd bytes [4 + n + margin + 4] (the byte pointers of
//   phase A run from two positions before to a few past their arrays);  seeding: nc sort keys;
//   BFS: claim words [padded board] + FIFO u16 [n + 2]
// `bfs` = optimise_wire can run at all (a solved board is wanted or another extension iteration
// follows): without it a 16x16 board needs 0.55 instead of 2.1 KB here, which is what lets two
// refill launches and the step launches share a CU's LDS.
__host__ __device__ inline int se_extra_words(int words_padded, int n, bool bfs_possible = true) {
  const int bfs = bfs_possible ? ((words_padded + 3) & ~3) + (n + 2 + 1) / 2 : 0;
  const int ext = (2 * (n + kSeMargin) + 12 + 3) / 4;
  const int seed = (n + 3) / 4;  // sort keys of the (m / 2)^2 candidate cells
  int w = bfs > ext ? bfs : ext;
  w = w > seed ? w : seed;
  return ((w + 3) & ~3) + 4;
}

// ---------------------------------------------------------------------------------------------
// Phase A: `count` links of the sweep's key chain.  Lane roles (sub-lane, all in the first row of
// 16 lanes), one Threefry evaluation per link for all of them:
//   0,1  the chain itself: blocks (0,2),(1,3) of split(K_s) -> K_{s+1} (y0), choice key C_s (y1)
//   4,5  blocks (0,2),(1,3) of split(C_{s-1})            -> its second child D_{s-1} (y1)
//   6    random_bits(C_{s-1}, ()) -> uniform of cell s-1  (use_random byte, kRnd only)
//   8    random_bits(D_{s-2}, ()) -> direction number of chain position s-2 (idx byte)
// Operands move by DPP (gfx9 dpp_ctrl encodings): quad_perm broadcasts lane 0 / lane 1 of the
// quad (the chain's y0 pair, one move deep), row_shr:4 hands the y1 pair of a quad to the next
// quad, and a bank-masked quad_perm leaves quad 0 as it is -- five moves, no selects, no LDS on
// the chain.
// Every lane stores one byte per link through its own running pointer (se_chain_links): no
// exec-mask branches in the loop.
// ---------------------------------------------------------------------------------------------
template <bool kRnd>
__device__ __forceinline__ void se_chain_link(Key2 &ck, uint32_t cx0, uint32_t cx1,
                                              uint8_t *idx_ptr, uint8_t *rnd_ptr, int ofs,
                                              float randomness) {
  uint32_t y0, y1;
  // (a formulation with 36 dependent levels instead of 45 -- rotations distributed over the
  // xors, v_xad_u32 + v_bitop3_b32, 10 more instructions -- was bit-exact and SLOWER, 75 vs 67 us
  // per 16x16 sweep: a lone wave pays ~7 cycles per instruction of this loop whatever their
  // dependencies, so the instruction count is what counts)
  threefry(ck.a, ck.b, cx0, cx1, y0, y1);
  // quad 0 (the chain) takes its y0 pair; the other quads get the y1 pair of the quad before
  const int a0 = __builtin_amdgcn_update_dpp(0, (int)y0, 0x00, 0xF, 0xF, true);  // quad_perm:[0,0,0,0]
  const int b0 = __builtin_amdgcn_update_dpp(0, (int)y0, 0x55, 0xF, 0xF, true);  // quad_perm:[1,1,1,1]
  const int sh = __builtin_amdgcn_update_dpp(0, (int)y1, 0x114, 0xF, 0xF, true);  // row_shr:4
  ck.a = (uint32_t)__builtin_amdgcn_update_dpp(a0, sh, 0x00, 0xF, 0xE, false);  // lanes 4..15
  ck.b = (uint32_t)__builtin_amdgcn_update_dpp(b0, sh, 0x55, 0xF, 0xE, false);
  idx_ptr[ofs] = (uint8_t)y0;  // low two bits = the direction number (masked by the reader)
  if (kRnd) rnd_ptr[ofs] = (uint8_t)(randomness > bits_to_uniform(y0));
}

// `count` links.  Every lane stores one byte per link at its running pointer + 0..3 and then
// moves the pointer on by 4 * inc: inc is 1 for the lane whose bytes are wanted and 0 for the
// others, whose four stores keep hitting the four bytes of their private dump word.
template <bool kRnd>
__device__ __forceinline__ void se_chain_links(Key2 &ck, uint32_t cx0, uint32_t cx1, int count,
                                               uint8_t *&idx_ptr, int idx_inc, uint8_t *&rnd_ptr,
                                               int rnd_inc, float randomness) {
  int s = 0;
#pragma unroll 1
  for (; s + 4 <= count; s += 4) {
    se_chain_link<kRnd>(ck, cx0, cx1, idx_ptr, rnd_ptr, 0, randomness);  // immediate ds offsets
    se_chain_link<kRnd>(ck, cx0, cx1, idx_ptr, rnd_ptr, 1, randomness);
    se_chain_link<kRnd>(ck, cx0, cx1, idx_ptr, rnd_ptr, 2, randomness);
    se_chain_link<kRnd>(ck, cx0, cx1, idx_ptr, rnd_ptr, 3, randomness);
    idx_ptr += 4 * idx_inc;
    rnd_ptr += 4 * rnd_inc;
  }
#pragma unroll 1
  for (; s < count; ++s) {
    se_chain_link<kRnd>(ck, cx0, cx1, idx_ptr, rnd_ptr, 0, randomness);
    idx_ptr += idx_inc;
    rnd_ptr += rnd_inc;
  }
}

// min over the sub-wave, result in every lane
template <int LPB>
__device__ __forceinline__ uint32_t subwave_min_u32(uint32_t v) {
  uint32_t o;
  o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false);  // [1,0,3,2]
  v = o < v ? o : v;
  o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false);  // [2,3,0,1]
  v = o < v ? o : v;
  o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false);  // half mirror
  v = o < v ? o : v;
  o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false);  // row mirror
  v = o < v ? o : v;
  if (LPB >= 32) {
    o = (uint32_t)__shfl_xor((int)v, 16);
    v = o < v ? o : v;
  }
  if (LPB >= 64) {
    o = (uint32_t)__shfl_xor((int)v, 32);
    v = o < v ? o : v;
  }
  return v;
}

// jax.random.randint(key, (), 0, 2^k) for a power-of-two span: the multiplier ((2^16 % span)^2 %
// span) is 0, so the draw is random_bits(split(key)[1], ()) % span.  Every lane evaluates it.
template <int LPB>
__device__ __forceinline__ uint32_t randint_pow2(const SubWave<LPB> &sw, Key2 key, uint32_t mask) {
  Key2 k1, k2;
  split2_lanes(sw, key, k1, k2);
  return threefry_call(k2.a, k2.b, 0u, 0u).x & mask;
}

// Direction number a try at chain position `rel` past K_n would draw (slow exact path, taken
// only when a cell's tries run past the positions phase A prepared).
__device__ __noinline__ uint32_t se_idx_after(uint32_t ka, uint32_t kb, int rel) {
  // rel links K <- split(K)[0], then choice key = split(K)[1], then its second child
  for (int i = 0; i < rel + 2; ++i) {
    uint32_t a0, a1, b0, b1;
    threefry(ka, kb, 0u, 2u, a0, a1);
    threefry(ka, kb, 1u, 3u, b0, b1);
    ka = i < rel ? a0 : a1;
    kb = i < rel ? b0 : b1;
  }
  uint32_t y0, y1;
  threefry(ka, kb, 0u, 0u, y0, y1);  // random_bits(split(choice key)[1], ()) % 4
  return y0 & 3u;
}

// Can the head/target at padded index P0 of the wire whose path value is `path` extend anywhere?
// (poss_extension_list :80-97 non-empty: an empty neighbour that touches the wire only through
// this cell.)  Which cells qualify does not depend on the sweep's frame.
__device__ __forceinline__ bool se_can_extend(const uint32_t *board, int P0, uint32_t path, int S) {
  bool any = false;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const int step = d == 0 ? -S : d == 1 ? -1 : d == 2 ? S : 1;
    const int perp = (d & 1) ? S : 1;
    const int Q = P0 + step;
    const uint32_t vq = board[Q];
    const uint32_t a = board[Q + step], b = board[Q + perp], e = board[Q - perp];
    any = any || (vq == 0u && !((a - path) < 3u || (b - path) < 3u || (e - path) < 3u));
  }
  return any;
}

// ---------------------------------------------------------------------------------------------
// The generator.  Lane w < N of the sub-wave owns wire w: (pr, pc) its POSITION cell, (tr, tc) its
// TARGET cell.  On return the LDS board holds the solved board; returns the number of sweeps
// that changed the board (NOT the reference's step_num, which also counts the idle sweeps a finished
// board waits through), with the sign bit set if a try loop hit its 512-try cap.
// `xtra` = se_extra_words() words of LDS private to the board.
// ---------------------------------------------------------------------------------------------
template <int LPB>
__device__ __forceinline__ int seed_extension_generate(const SubWave<LPB> &sw, uint32_t *board,
                                                       uint32_t *scr, uint32_t *xtra, const Geo &g,
                                                       const SeParams &p, Key2 key, int &pr,
                                                       int &pc, int &tr, int &tc) {
  const int l = sw.l, m = g.m, n = g.n, S = g.S, N = p.N;
  const uint64_t lt_mask = (1ull << l) - 1ull;

  // ---- return_solved_board :171  key, seedkey = split(key) -----------------------------------
  Key2 seedkey = key;
  if (!p.seed_only) split2_lanes(sw, key, key, seedkey);

  // ---- return_seeded_board :93-147 ------------------------------------------------------------
  // randint(seedkey, (), 0, 2), permutation(seedkey, nc) and randint(seedkey, (N,), 0, 2) all
  // draw from k2 = split(seedkey)[1] (the high words of _randint never matter for span 2)
  {
    Key2 k1, k2;
    split2_lanes(sw, seedkey, k1, k2);
    const uint32_t side = threefry_call(k2.a, k2.b, 0u, 0u).x & 1u;
    const int h = p.half, nc = h * h;
    uint32_t *sk = xtra;  // sort keys of the nc candidate cells
    {
      const uint32_t halfw = ((uint32_t)nc + 1u) >> 1;
      for (uint32_t j = l; j < halfw; j += LPB) {
        const bool has_hi = halfw + j < (uint32_t)nc;
        const uint2 y = threefry_call(k2.a, k2.b, j, has_hi ? halfw + j : 0u);
        sk[j] = y.x;
        if (has_hi) sk[halfw + j] = y.y;
      }
    }
    board_clear<LPB>(board, g, l);  // also fences the sort keys
    // stable ascending order: candidate j takes place #{k : (sk[k], k) < (sk[j], j)}
    for (int j = l; j < nc; j += LPB) {
      const uint32_t mine = sk[j];
      int rank = 0;
      for (int k = 0; k < nc; ++k) {
        const uint32_t o = sk[k];
        rank += (int)(o < mine || (o == mine && k < j));
      }
      if (rank < N) scr[rank] = (uint32_t)j;
    }
    wave_lds_fence();
    pr = pc = tr = tc = 0;
    if (l < N) {
      const int j = (int)scr[l];
      const int jr = j / h, jc = j - jr * h;
      const uint32_t which = random_bits_word(k2, (uint32_t)l, (uint32_t)N) & 1u;
      // :103-137  side 1: odd cells, offsets (-1, 0) / (0, -1); side 0: even cells, (0, 1) / (1, 0)
      tr = side ? 1 + 2 * jr : 2 * jr;
      tc = side ? 1 + 2 * jc : 2 * jc;
      pr = tr + (side ? (which ? 0 : -1) : (which ? 1 : 0));
      pc = tc + (side ? (which ? -1 : 0) : (which ? 0 : 1));
      board[pad_index(g, tr, tc)] = 3u * (uint32_t)l + 3u;  // :145 head_indices get TARGET
      board[pad_index(g, pr, pc)] = 3u * (uint32_t)l + 2u;  // :146 tail_indices get POSITION
    }
    wave_lds_fence();
  }

  uint8_t *idxb = reinterpret_cast<uint8_t *>(xtra) + 4;  // [n + margin] direction number per position
  uint8_t *rndb = idxb + n + kSeMargin + 4;               // [n] use_random per cell
  uint8_t *dump = reinterpret_cast<uint8_t *>(scr + l);   // this lane's private dump byte
  // counters of this lane's role in the phase-A evaluation (se_chain_links)
  const uint32_t cx0 = (l == 1 || l == 5) ? 1u : 0u;
  const uint32_t cx1 = (l < 2 || l == 4 || l == 5) ? cx0 + 2u : 0u;

  int total_sweeps = 0;
  bool capped = false;  // a try loop ran into its 512-try cap (reported in the sign bit of `sweeps`)
  for (int it = 0; it < p.iterations; ++it) {
    // :189  key, extkey, optkey = split(key, 3): blocks (0,3), (1,4), (2,5)
    Key2 extkey, optkey;
    {
      const uint32_t x = (uint32_t)(l % 3);
      const uint2 y = threefry_call(key.a, key.b, x, x + 3u);
      key.a = sw.bcast(y.x, 0);
      key.b = sw.bcast(y.x, 1);
      extkey.a = sw.bcast(y.x, 2);
      extkey.b = sw.bcast(y.y, 0);
      optkey.a = sw.bcast(y.y, 1);
      optkey.b = sw.bcast(y.y, 2);
    }

    // ---- extend_wires_jax ---------------------------------------------------------------------
    Key2 K = extkey;
    bool running = true;  // per board (sub-wave uniform)
    int step_num = 0;
    while (true) {
      // A board none of whose endpoints can extend is finished: no later sweep can change it
      // (which cells an endpoint may enter does not depend on the frame), and all the reference
      // still does with it is draw flips until an un-flipped sweep lets its comparison of the
      // flipped "before" with the un-flipped "after" succeed (:51-70, :186-189) -- on average
      // four more sweeps of m*m key-chain links each, a third of all the sweeps of a 16x16/10
      // batch and most of those of its slowest board.  None of that reaches an output (the
      // extension's key is local to extend_wires_jax), so the board stops here; `sweeps` counts
      // the sweeps that changed the board.  Conversely a board with an endpoint that can extend
      // always changes in the next sweep (the first such endpoint in sweep order still finds the
      // board as it was), so the reference's comparison never succeeds on a live board.
      if (running) {
        bool can = false;
        if (l < N) {
          const uint32_t path = 3u * (uint32_t)l + 1u;
          can = se_can_extend(board, pad_index(g, tr, tc), path, S) ||
                (p.two_sided && se_can_extend(board, pad_index(g, pr, pc), path, S));
        }
        running = sw.ballot(can) != 0ull && step_num < p.max_steps;
      }
      if (__ballot(running) == 0ull) break;
      // :63-65  key, flipkey, flopkey = split(key, 3); do_flip = [True, False][randint(.., 2)]
      Key2 fk, pk;
      {
        const uint32_t x = (uint32_t)(l % 3);
        const uint2 y = threefry_call(K.a, K.b, x, x + 3u);
        K.a = sw.bcast(y.x, 0);
        K.b = sw.bcast(y.x, 1);
        fk.a = sw.bcast(y.x, 2);
        fk.b = sw.bcast(y.y, 0);
        pk.a = sw.bcast(y.y, 1);
        pk.b = sw.bcast(y.y, 2);
      }
      const bool flip = randint_pow2(sw, fk, 1u) == 0u;
      const bool flop = randint_pow2(sw, pk, 1u) == 0u;

      // ---- phase A: the sweep's chain --------------------------------------------------------
      Key2 ck = K;
      // lane 8's byte of link s is the direction number of position s - 2, lane 6's the
      // use_random bit of cell s - 1
      uint8_t *ip = l == 8 ? idxb - 2 : dump, *rp = l == 6 ? rndb - 1 : dump;
      const int ii = l == 8 ? 1 : 0, ri = l == 6 ? 1 : 0;
      if (p.rnd_mode == 2) {
        se_chain_links<true>(ck, cx0, cx1, n, ip, ii, rp, ri, p.randomness);
      } else {
        se_chain_links<false>(ck, cx0, cx1, n, ip, ii, rp, ri, p.randomness);
      }
      const Key2 kn = ck;  // lanes 0,1 now hold K_n, the key the next sweep starts from
      if (p.rnd_mode == 2) {
        se_chain_links<true>(ck, cx0, cx1, kSeMargin + 2, ip, ii, rp, ri, p.randomness);
      } else {
        se_chain_links<false>(ck, cx0, cx1, kSeMargin + 2, ip, ii, rp, ri, p.randomness);
      }
      K.a = sw.bcast(kn.a, 0);
      K.b = sw.bcast(kn.b, 0);
      wave_lds_fence();

      // ---- phase B: the endpoints, in sweep order -------------------------------------------
      // frame -> physical: frame row fr is physical row (flip ? m-1-fr : fr), same for columns
      const int upF = flip ? S : -S, leftF = flop ? 1 : -1;
      const int d_step[4] = {upF, leftF, -upF, -leftF};  // poss order: up, left, down, right
      int cursor = 0;
      while (true) {
        uint32_t cand = 0xFFFFFFFFu;
        if (running && l < N) {
          const int sT = (flip ? m - 1 - tr : tr) * m + (flop ? m - 1 - tc : tc);
          if (sT >= cursor) cand = ((uint32_t)sT << 8) | ((uint32_t)l << 1) | 1u;
          if (p.two_sided) {
            const int sP = (flip ? m - 1 - pr : pr) * m + (flop ? m - 1 - pc : pc);
            const uint32_t cP = ((uint32_t)sP << 8) | ((uint32_t)l << 1);
            if (sP >= cursor && cP < cand) cand = cP;
          }
        }
        const uint32_t ev = subwave_min_u32<LPB>(cand);
        const bool act = ev != 0xFFFFFFFFu;
        if (__ballot(act) == 0ull) break;
        const int s = (int)(ev >> 8), w = (int)((ev >> 1) & 0x7Fu);
        const bool is_t = (ev & 1u) != 0u;
        const int fr = div_m(g, act ? s : 0), fc = (act ? s : 0) - fr * m;
        const int r = flip ? m - 1 - fr : fr, c = flop ? m - 1 - fc : fc;
        const int P0 = pad_index(g, r, c);
        const uint32_t path = 3u * (uint32_t)w + 1u;
        // poss_extension_list :80-97: open neighbours that touch the wire only through this cell
        uint32_t vq[4];
        uint32_t okm = 0u;  // bit d: direction d is a valid extension
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const int Q = P0 + d_step[d];
          const int perp = (d & 1) ? upF : leftF;
          vq[d] = board[Q];
          const uint32_t a = board[Q + d_step[d]], b = board[Q + perp], e = board[Q - perp];
          const bool touches = (a - path) < 3u || (b - path) < 3u || (e - path) < 3u;
          okm |= (uint32_t)(vq[d] == 0u && !touches) << d;
        }
        const bool any_ok = okm != 0u;
        // get_previous_neighbor_jax :200-234: above, below, left -- the last match wins; the
        // priority cell lies opposite to it
        const bool same_u = (vq[0] - path) < 3u, same_l = (vq[1] - path) < 3u,
                   same_d = (vq[2] - path) < 3u;
        const int pri = same_l ? 3 : same_d ? 0 : same_u ? 2 : -1;
        const bool pri_ok = pri >= 0 && ((okm >> (pri & 3)) & 1u) != 0u;
        const bool use_random = p.rnd_mode == 1 || (p.rnd_mode == 2 && rndb[act ? s : 0] != 0);
        const bool move = act && any_ok;
        int chosen = pri < 0 ? 0 : pri;
        bool need_try = move && !(pri_ok && !use_random);
        // :127-144  tries j = 0, 1, ...: direction number of chain position s + j until valid
        int pos = s;
        while (__ballot(need_try) != 0ull) {
          if (need_try) {
            uint32_t idx;
            if (pos < n + kSeMargin) {
              idx = idxb[pos] & 3u;
            } else {
              idx = se_idx_after(K.a, K.b, pos - n);
            }
            // (a valid direction comes up with probability >= 1/4 per try; the cap only keeps a
            // corrupted stream from spinning forever: 0.75^512 < 1e-63)
            if (((okm >> idx) & 1u) || pos - s >= 512) {
              capped = capped || !((okm >> idx) & 1u);
              chosen = ((okm >> idx) & 1u) ? (int)idx : __builtin_ctz(okm);
              need_try = false;
            }
            ++pos;
          }
        }
        if (move) {
          const int E = P0 + (chosen == 0 ? upF : chosen == 1 ? leftF : chosen == 2 ? -upF : -leftF);
          if (l == 0) {
            board[E] = path + (is_t ? 2u : 1u);  // the head/target value moves on ...
            board[P0] = path;                    // ... and leaves a path cell behind (:165-173)
          }
          if (l == w) {
            // physical step of frame direction `chosen`
            const int dr = chosen == 0 ? (flip ? 1 : -1) : chosen == 2 ? (flip ? -1 : 1) : 0;
            const int dc = chosen == 1 ? (flop ? 1 : -1) : chosen == 3 ? (flop ? -1 : 1) : 0;
            if (is_t) {
              tr += dr;
              tc += dc;
            } else {
              pr += dr;
              pc += dc;
            }
          }
        }
        if (act) cursor = s + 1;
        wave_lds_fence();
      }

      if (running) ++step_num;  // (this sweep changed the board: see the top of the loop)
    }
    total_sweeps += step_num;

    // ---- optimise_wire for every wire :194-208 ---------------------------------------------
    if (it + 1 < p.iterations || p.bfs_last) {
      // optkeys = split(optkey, N); per wire: permutation(key, arange(4)) = stable argsort of
      // random_bits(split(key)[1], 4)
      uint32_t my_perm = 0u;  // direction numbers in search order, 2 bits each
      if (l < N) {
        const Key2 ok_w = split_child(optkey, (uint32_t)l, (uint32_t)N);
        const uint2 a = threefry_call(ok_w.a, ok_w.b, 0u, 2u), b = threefry_call(ok_w.a, ok_w.b, 1u, 3u);
        const uint2 c = threefry_call(a.y, b.y, 0u, 2u), d = threefry_call(a.y, b.y, 1u, 3u);
        const uint32_t k4[4] = {c.x, d.x, c.y, d.y};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int rank = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) rank += (int)(k4[k] < k4[i] || (k4[k] == k4[i] && k < i));
          my_perm |= (uint32_t)i << (2 * rank);
        }
      }
      uint32_t *claim = xtra;  // [padded board]: FIFO position * 4 + direction number, or ~0
      uint16_t *fifo = reinterpret_cast<uint16_t *>(xtra + ((g.words + 3) & ~3));
      // grid_utils.py:87-88: moves numbered up, right, down, left
#pragma unroll 1
      for (int w = 0; w < N; ++w) {
        const uint32_t perm = sw.bcast(my_perm, w);
        const int startP = (int)sw.bcast((uint32_t)pad_index(g, pr, pc), w);
        const int endP = (int)sw.bcast((uint32_t)pad_index(g, tr, tc), w);
        const uint32_t path = 3u * (uint32_t)w + 1u;
        for (int i = l; i < g.words; i += LPB) claim[i] = 0xFFFFFFFFu;
        wave_lds_fence();
        if (l == 0) {
          board[startP] = path;  // :566-568
          board[endP] = path;
          claim[startP] = 0u;
          fifo[0] = (uint16_t)startP;
        }
        wave_lds_fence();
        int head = 0, tail = 1;
        bool found = false;
        bool searching = true;
        while (__ballot(searching) != 0ull) {
          const int level_end = tail;
          for (int base = head; __ballot(searching && base < level_end) != 0ull; base += LPB / 4) {
            const int fpos = base + (l >> 2), j = l & 3;
            const bool valid = searching && fpos < level_end;
            const int cur = valid ? (int)fifo[fpos] : startP;
            const uint32_t mv = (perm >> (2 * j)) & 3u;
            const int q = cur + (mv == 0u ? -S : mv == 1u ? 1 : mv == 2u ? S : -1);
            const uint32_t bq = board[q];
            const bool elig = valid && (bq == 0u || bq == path);
            const uint32_t ckey = ((uint32_t)fpos << 2) | (uint32_t)j;
            if (elig) atomicMin(&claim[q], ckey);
            wave_lds_fence();
            const bool win = elig && claim[q] == ckey;
            const uint64_t wm = sw.ballot(win);
            if (win) fifo[tail + __popcll(wm & lt_mask)] = (uint16_t)q;
            tail += __popcll(wm);
            wave_lds_fence();
            found = claim[endP] != 0xFFFFFFFFu;
            if (found) searching = false;
          }
          head = level_end;
          if (head >= tail) searching = false;  // nothing left (cannot happen: a wire is connected)
        }
        // get_path / remove_path / jax_fill_grid :607-612
        if (found) {
          for (int idx = l; idx < n; idx += LPB) {
            const int rr = div_m(g, idx);
            const int P = pad_index(g, rr, idx - rr * m);
            if (board[P] == path) board[P] = 0u;
          }
          wave_lds_fence();
          int cur = endP;
          int guard = 0;
          while (__ballot(found && cur != startP && guard < n) != 0ull) {
            if (found && cur != startP) {
              if (l == 0) board[cur] = cur == endP ? path + 2u : path;
              cur = (int)fifo[claim[cur] >> 2];
              ++guard;
            }
          }
        }
        if (l == 0) {
          board[startP] = path + 1u;
          if (!found) board[endP] = path + 2u;
        }
        wave_lds_fence();
      }
    }
  }
  // never observed (0.75^512 < 1e-63): made visible rather than silently picking a direction
  return sw.ballot(capped) != 0ull ? (int)((uint32_t)total_sweeps | 0x80000000u) : total_sweeps;
}

}  // namespace rbg
