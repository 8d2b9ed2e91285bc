// rbg_regen.h -- Generator.__call__(key) -> State on an LDS-resident env, for every generator the
// reference can put behind Connector (routing_board_generation/rl_training/setup_train.py:112-152):
// what the env kernels run when an episode (re)starts and what the stash refill runs ahead of time.
//
// All of the reference's share one key protocol -- `key, _ = split(key)` first, State.key = that child
// (parallel_random_walk_generator.py:52, uniform_generator.py:76, sequential_random_walk_generator.py
// :40, random_seed_generator.py:34, offline_generation/dataset_generator_jax.py:114) -- which is
// what lets the next-episode stash name the episode after the next without generating anything
// (rbg_kernels_stash.h::request_refill).
#pragma once
#include "rbg_env.h"
#include "rbg_seedext.h"
#include "rbg_seqwalk.h"
#include "rbg_walk_philox.h"

namespace rbg {

// include/rbg_prw.h::RBG_GEN_*
enum { kGenPrw = 0, kGenUniform = 1, kGenSeqRw = 2, kGenSeedExt = 3, kGenDataset = 4,
       kGenPhilox = 5 };

// State.key of the episode AFTER the one whose State.key is k: VmapAutoResetWrapper resets from
// split(k)[0], and the generator derives its State.key from that -- for the reference's generators
// by another split (see above), for the opt-in Philox walk as words 0, 1 of
// philox(key, (0, 0, 0, 3)) (include/rbg_prw.h).  No generator has to run to know it, which is
// what lets a stash request rows several episodes ahead (rbg_kernels_stash.h::request_refill).
template <int LPB, int KIND>
__device__ __forceinline__ Key2 next_state_key(const SubWave<LPB> &sw, Key2 k) {
  Key2 rk, unused, out;
  split2_lanes(sw, k, rk, unused);
  if constexpr (KIND == kGenPhilox) {
    const Philox4 p = philox4x32_10(rk.a, rk.b, 0u, 0u, 0u, 3u);
    out.a = p.w[0];
    out.b = p.w[1];
  } else {
    split2_lanes(sw, rk, out, unused);
  }
  return out;
}

struct GenSpec {
  int kind;
  int extra_words;         // LDS words per env after its slab (SeedExtension: chain bytes / BFS)
  SeParams se;             // kGenSeedExt
  const int32_t *ds_heads;    // kGenDataset: [n_boards][N][2]
  const int32_t *ds_targets;  // kGenDataset
  uint32_t ds_span, ds_mult;  // n_boards and jax's _randint multiplier ((2^16 % span)^2 % span)
};

// LDS words of one env: padded board + per-board scratch (+ the generator's own area)
template <int LPB>
__device__ __forceinline__ int env_slab_words(const Geo &g, const GenSpec &gs) {
  return slab_words<LPB>(g) + gs.extra_words;
}

// Runs generator KIND for `key_in` (the generator's argument): leaves the TRAINING grid (heads and
// targets only) on the padded LDS board, fills the agent registers of sub-lanes < N, returns
// State.key.  `W` = the env's scratch words; the generator's own area follows the slab.
// g.thr must be the shuffle threshold of this kind's prefix (N for the walk, 2N for Uniform).
template <int LPB, int KIND>
__device__ __forceinline__ Key2 regenerate_kind(const SubWave<LPB> &sw, uint32_t *board,
                                                uint32_t *W, const Geo &g, int N, Key2 key_in,
                                                AgentRegs &ag, const GenSpec &gs) {
  if constexpr (KIND == kGenPrw) {
    return regenerate(sw, board, W, g, N, key_in, ag);
  } else if constexpr (KIND == kGenUniform) {
    // UniformRandomGenerator.__call__ (uniform_generator.py:70-109): key, pos_key = split(key);
    // first 2N entries of permutation(pos_key, m * m): N starts, then N targets
    Key2 kept, pos_key, unused, sub;
    split2_lanes(sw, key_in, kept, pos_key);
    split2_lanes(sw, pos_key, unused, sub);  // _shuffle: key, subkey = split(key)
    int sel0, sel1;
    shuffle_prefix(sw, board, W, g, sub, 2 * N, g.thr, sel0, sel1);
    const int e_t = sw.l + N;  // prefix entry holding this agent's target (< 2 * LPB)
    const int src = sw.h * LPB + (e_t & (LPB - 1));
    const int t_lo = __shfl(sel0, src), t_hi = __shfl(sel1, src);
    const int tcell = e_t < LPB ? t_lo : t_hi;
    board_clear<LPB>(board, g, sw.l);
    if (sw.l < N) {
      ag.r0 = ag.r = div_m(g, sel0);
      ag.c0 = ag.c = sel0 - ag.r0 * g.m;
      ag.tr = div_m(g, tcell);
      ag.tc = tcell - ag.tr * g.m;
      board[pad_index(g, ag.r0, ag.c0)] = 3u * (uint32_t)sw.l + 2u;
      board[pad_index(g, ag.tr, ag.tc)] = 3u * (uint32_t)sw.l + 3u;  // distinct cells
    }
    wave_lds_fence();
    return kept;
  } else if constexpr (KIND == kGenSeqRw) {
    static_assert(LPB == kWaveSize, "SequentialRandomWalk runs one board per wavefront");
    Key2 key, unused;
    split2_lanes(sw, key_in, key, unused);
    int first_cell, last_cell;
    const int attempts = seq_rw_generate(board, nullptr, g, N, key, sw.l, first_cell, last_cell);
    if (attempts == 0 && sw.l == 0) {
      // total failure: every start/end is (0, 0); duplicate-index scatter applied in agent order
      // (the last agent's target; rbg_kernels_generators.h::seq_rw_generate_kernel)
      board[pad_index(g, 0, 0)] = 3u * (uint32_t)N;
    }
    wave_lds_fence();
    board_to_training(sw, board, g);
    if (sw.l < N) {
      ag.r0 = ag.r = div_m(g, first_cell);
      ag.c0 = ag.c = first_cell - ag.r0 * g.m;
      ag.tr = div_m(g, last_cell);
      ag.tc = last_cell - ag.tr * g.m;
    }
    return key;
  } else if constexpr (KIND == kGenSeedExt) {
    Key2 key, unused;
    split2_lanes(sw, key_in, key, unused);
    int pr, pc, tr, tc;
    seed_extension_generate(sw, board, W, board + slab_words<LPB>(g), g, gs.se, key, pr, pc, tr, tc);
    board_to_training(sw, board, g);
    ag.r0 = ag.r = pr;
    ag.c0 = ag.c = pc;
    ag.tr = tr;
    ag.tc = tc;
    return key;
  } else if constexpr (KIND == kGenPhilox) {
    // ParallelRandomWalkGenerator(stream = "philox") (rbg_walk_philox.h; project-defined stream):
    // the key is used as it comes, State.key = words 0, 1 of philox(key, (0, 0, 0, 3))
    Walker w;
    prw_walk_philox(sw, board, W, g, N, key_in, w);
    board_to_training(sw, board, g);
    ag.r0 = ag.r = w.r0;
    ag.c0 = ag.c = w.c0;
    ag.tr = w.r;
    ag.tc = w.c;
    const Philox4 p = philox4x32_10(key_in.a, key_in.b, 0u, 0u, 0u, 3u);
    Key2 out = {p.w[0], p.w[1]};
    return out;
  } else {
    static_assert(KIND == kGenDataset, "unknown generator kind");
    // BoardDatasetGeneratorJAX.__call__ (dataset_generator_jax.py:112-141): key, _ = split(key);
    // which = randint(key, (), 0, n_boards) ([3P] jax/_src/random.py::_randint: k1, k2 =
    // split(key); ((hi % span) * mult + lo % span) % span in wrapping uint32 arithmetic)
    Key2 k0, unused, k1, k2;
    split2_lanes(sw, key_in, k0, unused);
    split2_lanes(sw, k0, k1, k2);
    const uint2 hi = threefry_call(k1.a, k1.b, 0u, 0u), lo = threefry_call(k2.a, k2.b, 0u, 0u);
    const uint32_t which = ((hi.x % gs.ds_span) * gs.ds_mult + (lo.x % gs.ds_span)) % gs.ds_span;
    board_clear<LPB>(board, g, sw.l);
    if (sw.l < N) {
      const size_t o = ((size_t)which * N + sw.l) * 2;
      const int2 h = *reinterpret_cast<const int2 *>(&gs.ds_heads[o]);
      const int2 t = *reinterpret_cast<const int2 *>(&gs.ds_targets[o]);
      ag.r0 = ag.r = h.x;
      ag.c0 = ag.c = h.y;
      ag.tr = t.x;
      ag.tc = t.y;
      board[pad_index(g, h.x, h.y)] = 3u * (uint32_t)sw.l + 2u;
    }
    wave_lds_fence();
    if (sw.l < N) board[pad_index(g, ag.tr, ag.tc)] = 3u * (uint32_t)sw.l + 3u;  // targets last
    wave_lds_fence();
    return k0;
  }
}

}  // namespace rbg
