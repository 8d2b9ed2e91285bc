// rbg_kernels_stash.h -- the next-episode stash (include/rbg_prw.h::rbg_stash) on the device side:
// lookup, early probes, row prefetch, refill requests.  Shared by the env kernels and the rollout.
#pragma once
#include "rbg_kernels_common.h"
#include "rbg_regen.h"

namespace {

// Pre-generated next-episode states (include/rbg_prw.h::rbg_stash).  `next` of env b is the
// State the auto-reset of an episode with State.key == tag[b] must produce.
struct StashPtrs {
  StatePtrs next;        // depth * B rows; slot s of env b = row b * depth + s (stash_row)
  uint32_t *tag;         // [depth*B][2]
  int32_t *ready_at;     // [depth*B] first launch tick that may consume the row
  uint32_t *refill_key;  // [depth*B][2] the State.key whose successor the row should hold next
  int32_t *hits;         // [B] nullable diagnostics: resets served from the stash
  int32_t *misses;       // [B] nullable: resets that ran the generator inline although a stash exists
  int32_t *scratch;      // [depth*B + 1] nullable: [0] = number of listed rows, [1..] = the rows
  int32_t *inflight;     // [depth*B] nullable: 1 while a listed refill is generating the row
  uint32_t *far_key;     // [B][2] nullable: the farthest key env b has requested a row for
  int depth;             // 1 .. kMaxStashDepth
};
constexpr int kMaxStashDepth = 12;

// Row of slot s of env b.  The slots of an env are ADJACENT: its `depth` tags are one or two
// cache lines and its ready ticks one, so the early probe of a deep stash (every wave of every
// step launch reads them) touches 3 lines per env instead of 2 x depth lines strided B rows
// apart (round 4; rounds 2-3 had row s * B + b).  At 4096 envs the step time did not move either
// way (16.5 / 17.5 us for the two rolling generators before and after: those lines sit in L2 /
// Infinity Cache); the layout is kept because it is the one that stays cheap at larger batches.
__device__ __forceinline__ int stash_row(const StashPtrs &st, int b, int s) {
  return b * st.depth + s;
}

// Row of the stash that holds the successor of the episode with State.key `key`, or -1.
__device__ __forceinline__ int stash_lookup(const StashPtrs &st, int B, int b, Key2 key,
                                            int tick) {
  int row = -1;
  if (st.tag) {
    for (int s = 0; s < st.depth; ++s) {
      const int r = stash_row(st, b, s);
      const uint2 tag = *reinterpret_cast<const uint2 *>(&st.tag[2 * (size_t)r]);
      if (tick >= st.ready_at[r] && tag.x == key.a && tag.y == key.b) row = r;
    }
  }
  return row;
}

// The same lookup with its loads issued early: lane s < depth fetches (tag, ready_at) of slot s
// of the env when the wave starts, so that an env that ends its episode a microsecond later finds
// the answer in registers instead of starting two dependent global loads on the launch's tail.
struct StashProbe {
  uint2 tag;
  int ready_at;
  uint2 far;  // lane 0: StashPtrs::far_key of the env (0 without one)
};
// DEEP = false: the launch was given no stash or a plain one (one slot per env, no in-flight
// claims -- what the walk's env path runs with), and the kernel carries none of the deep stash's
// probes, request rule or priorities (the host picks the instantiation: launch_env).
template <bool DEEP = true>
__device__ __forceinline__ StashProbe stash_probe_issue(const StashPtrs &st, int B, int b,
                                                        int lane) {
  StashProbe p = {make_uint2(0u, 0u), 0x7FFFFFFF, make_uint2(0u, 0u)};
  if (st.tag && lane < (DEEP ? st.depth : 1)) {
    const int r = DEEP ? stash_row(st, b, lane) : b;  // (the plain form: one slot, row b)
    p.tag = *reinterpret_cast<const uint2 *>(&st.tag[2 * (size_t)r]);
    p.ready_at = st.ready_at[r];
  }
  if constexpr (DEEP) {
    if (st.far_key && lane == 0)
      p.far = *reinterpret_cast<const uint2 *>(&st.far_key[2 * (size_t)b]);
  }
  return p;
}
template <bool DEEP = true>
__device__ __forceinline__ int stash_probe_row(const StashPtrs &st, const StashProbe &p, int B,
                                               int b, int lane, Key2 key, int tick) {
  const bool ok = st.tag && lane < (DEEP ? st.depth : 1) && tick >= p.ready_at &&
                  p.tag.x == key.a && p.tag.y == key.b;
  if constexpr (!DEEP) return (__ballot(ok) & 1ull) ? b : -1;
  const uint32_t m = (uint32_t)__ballot(ok) & ((1u << kMaxStashDepth) - 1u);
  return m == 0u ? -1 : stash_row(st, b, 31 - __builtin_clz(m));  // the later slot wins, as in stash_lookup
}

// An episode that ends on the time limit is known to end as soon as step_count is loaded: its
// wave fetches the stash row while it steps (registers only; one dwordx4 per lane covers grids
// up to 16x16) and finds the next episode's State ready when it gets there.  Other terminations
// (every agent connected or blocked) take the plain load_state path.
struct RowPrefetch {
  int row;  // -1: nothing prefetched
  int4 grid;
  int2 s, t, p;
  uint2 key;
};
__device__ __forceinline__ RowPrefetch stash_row_issue(const StatePtrs &next, const Geo &g, int N,
                                                       int row, int lane) {
  RowPrefetch pf;
  pf.row = row;
  pf.grid = make_int4(0, 0, 0, 0);
  pf.s = pf.t = pf.p = make_int2(0, 0);
  pf.key = make_uint2(0u, 0u);
  if (row >= 0) {
    if (lane * 4 < g.n)
      pf.grid = *reinterpret_cast<const int4 *>(&next.grid[(size_t)row * g.n + lane * 4]);
    if (lane < N) {
      const size_t o = ((size_t)row * N + lane) * 2;
      pf.s = *reinterpret_cast<const int2 *>(&next.start[o]);
      pf.t = *reinterpret_cast<const int2 *>(&next.target[o]);
      pf.p = *reinterpret_cast<const int2 *>(&next.position[o]);
    }
    pf.key = *reinterpret_cast<const uint2 *>(&next.key[2 * (size_t)row]);
  }
  return pf;
}
__device__ __forceinline__ void stash_row_commit(const SubWave<64> &sw, uint32_t *board,
                                                 const Geo &g, const RowPrefetch &pf,
                                                 AgentRegs &ag, Key2 &key) {
  const int lane = sw.l;
  for (int w = lane; w < g.words; w += kWaveSize) board[w] = kWall;
  wave_lds_fence();
  if (lane * 4 < g.n) {
    const int r = div_m(g, lane * 4);
    *reinterpret_cast<uint4 *>(&board[pad_index(g, r, lane * 4 - r * g.m)]) =
        make_uint4((uint32_t)pf.grid.x, (uint32_t)pf.grid.y, (uint32_t)pf.grid.z,
                   (uint32_t)pf.grid.w);
  }
  wave_lds_fence();
  ag.r0 = pf.s.x; ag.c0 = pf.s.y; ag.tr = pf.t.x; ag.tc = pf.t.y; ag.r = pf.p.x; ag.c = pf.p.y;
  key.a = pf.key.x;
  key.b = pf.key.y;
}

// After a (re)start with State.key `key`: ask the next refill for what comes after it
// (include/rbg_prw.h::rbg_stash).  With `depth` slots per env the env wants the successors of
// its next `depth` episodes: the rows tagged key, f(key), f(f(key)), ... with
// f = rbg_regen.h::next_state_key<KIND> -- split(split(k)[0])[0] for the reference's generators:
// VmapAutoResetWrapper's split, then the generator's -- i.e. the State.key of the episode after
// the one with State.key k; no generator has to run to know it.  Rows that already carry one of those requests
// keep it (and their ready_at); the others -- just consumed, stale, or never used -- take the
// missing ones.  depth 1: the one row is requested for `key`.  Called by the whole sub-wave.
// `hit_row` >= 0: the restart consumed that row of the stash.  With a far key per env
// (StashPtrs::far_key, deep stashes) that is all the bookkeeping there is: the rows other than
// the consumed one already carry the requests of the chain (by induction: every restart of the
// env goes through here, and a restart that found no row takes the general rule below), so the
// consumed row takes the one new request, f(far), and far moves on -- two Threefry evaluations
// and two stores instead of up to 2 (D - 1) evaluations and D^2 compares on the wave the launch
// is waiting for.  `far_in` = the env's far key as fetched with the early probe (lane 0).
template <int KIND, int MAXD = 12, int LPB>
__device__ __forceinline__ void request_refill(const SubWave<LPB> &sw, const StashPtrs &st, int B,
                                               int b, Key2 key, int ready_at, int hit_row = -1,
                                               bool have_far = false,
                                               uint2 far_in = make_uint2(0u, 0u)) {
  if (!st.refill_key) return;
  if (st.depth == 1) {
    if (sw.l == 0) {
      *reinterpret_cast<uint2 *>(&st.refill_key[2 * (size_t)b]) = make_uint2(key.a, key.b);
      st.ready_at[b] = ready_at;
    }
    return;
  }
  if constexpr (MAXD == 1) return;  // (a kernel built for plain stashes: launch_env)
  if (st.far_key && hit_row >= 0) {  // (uniform in the sub-wave)
    Key2 far;
    if (have_far) {
      far.a = sw.bcast(far_in.x, 0);
      far.b = sw.bcast(far_in.y, 0);
    } else {
      const uint2 f = *reinterpret_cast<const uint2 *>(&st.far_key[2 * (size_t)b]);
      far.a = f.x;
      far.b = f.y;
    }
    const Key2 nf = next_state_key<LPB, KIND>(sw, far);
    if (sw.l == 0) {
      *reinterpret_cast<uint2 *>(&st.refill_key[2 * (size_t)hit_row]) = make_uint2(nf.a, nf.b);
      st.ready_at[hit_row] = ready_at;
      *reinterpret_cast<uint2 *>(&st.far_key[2 * (size_t)b]) = make_uint2(nf.a, nf.b);
    }
    return;
  }
  const int D = st.depth < MAXD ? st.depth : MAXD;  // (MAXD: what this kernel is built for)
  Key2 want[MAXD];
  want[0] = key;
#pragma unroll
  for (int j = 1; j < MAXD; ++j) {
    want[j] = want[j - 1];
    if (j < D) want[j] = next_state_key<LPB, KIND>(sw, want[j - 1]);  // (uniform branch)
  }
  if (sw.l == 0) {
    uint2 cur[MAXD];
    uint32_t row_used = 0u, want_met = 0u;
#pragma unroll
    for (int s = 0; s < MAXD; ++s)
      cur[s] = s < D ? *reinterpret_cast<const uint2 *>(&st.refill_key[2 * (size_t)stash_row(st, b, s)])
                     : make_uint2(0u, 0u);
#pragma unroll
    for (int j = 0; j < MAXD; ++j) {
#pragma unroll
      for (int s = 0; s < MAXD; ++s) {
        const bool hit = j < D && s < D && !((row_used >> s) & 1u) && !((want_met >> j) & 1u) &&
                         cur[s].x == want[j].a && cur[s].y == want[j].b;
        row_used |= (uint32_t)hit << s;
        want_met |= (uint32_t)hit << j;
      }
    }
#pragma unroll
    for (int j = 0; j < MAXD; ++j) {
      if (j < D && !((want_met >> j) & 1u)) {
        const int s = __builtin_ctz(~row_used);  // a free row exists: D rows, fewer than D kept
        const size_t r = (size_t)stash_row(st, b, s);
        *reinterpret_cast<uint2 *>(&st.refill_key[2 * r]) = make_uint2(want[j].a, want[j].b);
        st.ready_at[r] = ready_at;
        row_used |= 1u << s;
      }
    }
    if (st.far_key) {
      Key2 far = want[0];
#pragma unroll
      for (int j = 1; j < MAXD; ++j) far = j < D ? want[j] : far;
      *reinterpret_cast<uint2 *>(&st.far_key[2 * (size_t)b]) = make_uint2(far.a, far.b);
    }
  }
}

}  // namespace
