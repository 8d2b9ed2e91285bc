/*
 * prw_philox.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (part of libprw_oracle.so).
 *
 * CPU restatement of the opt-in `stream="philox"` variant of the ParallelRandomWalk generator
 * (north_star asks for a counter-based Philox kernel; the reference's stream is
 * jax's Threefry, which the default path reproduces bit-exactly).  PARITY UNPINNED BY DESIGN: this
 * is a different random process, defined by this project; its only oracle is this file.  The walk
 * itself -- available cells, weighted choice, max-join collision rule, head/target encodings -- is
 * the reference's (JAXI/parallel_random_walk.py:60-447) and is shared with prw_oracle.c; only the
 * sources of randomness differ:
 *
 *   Philox-4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11; M0 = 0xD2511F53, M1 = 0xCD9E8D57,
 *   W0 = 0x9E3779B9, W1 = 0xBB67AE85), key = the two words of the board's input key, counter
 *   (c0, c1, c2, c3) with a domain tag in c3:
 *     tag 1  start cells: the sort key of cell j is word j % 4 of philox(key, (j / 4, 0, 0, 1));
 *            the N cells with the smallest (sort key, j), ascending, are the starts of agents
 *            0 .. N-1 (what jax's permutation-by-sort does with its own keys);
 *     tag 2  the walk: agent i's uniform at iteration t comes from word t % 4 of
 *            philox(key, (t / 4, i, 0, 2)) through jax's float32 mantissa fill; the choice is
 *            jax's (cumsum, r = cum[-1] * (1 - u), first cum >= r);
 *     tag 3  State.key of the generator = words 0, 1 of philox(key, (0, 0, 0, 3)).
 *   No key chain, no per-iteration split: 1/4 Philox evaluation per agent and iteration instead
 *   of 2 + 2/N Threefry evaluations.
 */
#include <stdint.h>
#include <string.h>

enum { PATH = 1, POSITION = 2, TARGET = 3 };
#define PHILOX_MAX_AGENTS 64
#define PHILOX_MAX_CELLS 1600

typedef struct {
  int32_t id;
  int32_t start[2];
  int32_t target[2];
  int32_t position[2];
} agent_t; /* same layout as prw_oracle.c */

/* shared with prw_oracle.c (same shared object) */
void oracle_available_cells(const int32_t *grid, int rows, int cols, int32_t cell, int32_t out[4]);
int oracle_action_from_positions(int32_t p1, int32_t p2, int cols);
int oracle_continue_stepping(const int32_t *grid, const agent_t *agents, int m, int N);
void oracle_step_agents_with_actions(int32_t *grid, agent_t *agents, int m, int N,
                                     const int32_t *actions);

void oracle_philox4x32_10(const uint32_t key[2], const uint32_t ctr[4], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static float bits_to_uniform(uint32_t bits) {
  uint32_t fb = (bits >> 9) | 0x3F800000u;
  float f;
  memcpy(&f, &fb, 4);
  return f - 1.0f;
}

/* generate_board with the Philox stream; heads/targets agent-major [N][2]; returns the trip count */
int oracle_prw_philox_generate_board(const uint32_t key[2], int m, int N, int32_t *heads,
                                     int32_t *targets, int32_t *solved) {
  const int n = m * m;
  agent_t agents[PHILOX_MAX_AGENTS];
  memset(solved, 0, sizeof(int32_t) * (size_t)n);
  /* ---- start cells (tag 1) */
  uint32_t sk[PHILOX_MAX_CELLS + 4];
  for (int q = 0; q * 4 < n; ++q) {
    uint32_t ctr[4] = {(uint32_t)q, 0u, 0u, 1u};
    oracle_philox4x32_10(key, ctr, &sk[4 * q]);
  }
  uint8_t taken[PHILOX_MAX_CELLS];
  memset(taken, 0, (size_t)n);
  for (int i = 0; i < N; ++i) { /* i-th smallest (sort key, index) */
    int best = -1;
    for (int j = 0; j < n; ++j)
      if (!taken[j] && (best < 0 || sk[j] < sk[best])) best = j;
    taken[best] = 1;
    agents[i].id = i;
    agents[i].start[0] = agents[i].position[0] = best / m;
    agents[i].start[1] = agents[i].position[1] = best % m;
    agents[i].target[0] = agents[i].target[1] = -1;
    solved[best] = 3 * i + POSITION;
  }
  /* ---- the walk (tag 2) */
  int iters = 0;
  while (oracle_continue_stepping(solved, agents, m, N)) {
    int32_t actions[PHILOX_MAX_AGENTS];
    for (int i = 0; i < N; ++i) {
      uint32_t ctr[4] = {(uint32_t)iters >> 2, (uint32_t)i, 0u, 2u}, w[4];
      oracle_philox4x32_10(key, ctr, w);
      const float u = bits_to_uniform(w[iters & 3]);
      int32_t cell = agents[i].position[0] * m + agents[i].position[1], avail[4];
      oracle_available_cells(solved, m, m, cell, avail);
      float cum[4], acc = 0.0f;
      for (int k = 0; k < 4; ++k) {
        acc = acc + ((avail[k] != -1) ? 1.0f : 0.0f);
        cum[k] = acc;
      }
      volatile float one_minus_u = 1.0f - u; /* no FMA contraction */
      const float r = cum[3] * one_minus_u;
      int idx = 0;
      while (idx < 4 && cum[idx] < r) ++idx;
      if (idx > 3) idx = 3;
      actions[i] = oracle_action_from_positions(cell, avail[idx], m);
    }
    oracle_step_agents_with_actions(solved, agents, m, N, actions);
    ++iters;
  }
  for (int i = 0; i < N; ++i) {
    heads[2 * i] = agents[i].start[0];
    heads[2 * i + 1] = agents[i].start[1];
    targets[2 * i] = agents[i].position[0];
    targets[2 * i + 1] = agents[i].position[1];
  }
  for (int i = 0; i < N; ++i) solved[heads[2 * i] * m + heads[2 * i + 1]] = 3 * i + POSITION;
  for (int i = 0; i < N; ++i) solved[targets[2 * i] * m + targets[2 * i + 1]] = 3 * i + TARGET;
  return iters;
}

/* vmap(ParallelRandomWalkGenerator(stream="philox").__call__) over keys [B][2] */
int oracle_prw_philox_batch(const uint32_t *keys, int B, int m, int N, int32_t *grid,
                            int32_t *step_count, int32_t *start, int32_t *target,
                            int32_t *position, uint32_t *key_out, int32_t *solved, int32_t *iters,
                            int threads) {
  if (B < 0 || m < 2 || m * m > PHILOX_MAX_CELLS || N < 1 || N > PHILOX_MAX_AGENTS || N > m * m)
    return -1;
  (void)threads;
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads > 0 ? threads : 1)
  for (int b = 0; b < B; ++b) {
    const int n = m * m;
    int32_t scratch[PHILOX_MAX_CELLS];
    int32_t *sv = solved ? &solved[(size_t)b * n] : scratch;
    int32_t *st = &start[(size_t)b * N * 2], *tg = &target[(size_t)b * N * 2];
    const int it = oracle_prw_philox_generate_board(&keys[2 * b], m, N, st, tg, sv);
    if (iters) iters[b] = it;
    if (grid) { /* training grid: heads then targets */
      int32_t *g = &grid[(size_t)b * n];
      memset(g, 0, sizeof(int32_t) * (size_t)n);
      for (int i = 0; i < N; ++i) g[st[2 * i] * m + st[2 * i + 1]] = 3 * i + POSITION;
      for (int i = 0; i < N; ++i) g[tg[2 * i] * m + tg[2 * i + 1]] = 3 * i + TARGET;
    }
    if (position) memcpy(&position[(size_t)b * N * 2], st, sizeof(int32_t) * 2 * (size_t)N);
    if (step_count) step_count[b] = 0;
    if (key_out) {
      uint32_t ctr[4] = {0u, 0u, 0u, 3u}, w[4];
      oracle_philox4x32_10(&keys[2 * b], ctr, w);
      key_out[2 * b] = w[0];
      key_out[2 * b + 1] = w[1];
    }
  }
  return 0;
}
