/*
 * prw_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C CPU restatement of the reference's ParallelRandomWalk board generator and of the
 * Jumanji `Connector` reset/step it is coupled to.  It exists only as the *checker*:
 * tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load it; nothing under
 * routing_board_generation_amd/ (the product) may import, link or call it.
 *
 * Parity status
 *   PINNED   (bit-exact against the reference's own goldens, tests/golden/reference_goldens.json):
 *            Threefry/split/random_bits/uniform/choice/shuffle, _initialise_agents, _step,
 *            generate_board, ParallelRandomWalkGenerator.__call__ (single board and the 100-board
 *            vmapped batch printed in package_evaluation/profiling_generators.ipynb),
 *            UniformRandomGenerator.__call__, and the pure helpers tested by
 *            test_parallel_random_walk_board.py.
 *   UNPINNED (no golden exists in the reference; semantics "defined by this project, modelled on
 *            jumanji 0.2.2" -- see DESIGN.md): Connector.step reward/discount/termination,
 *            _obs_from_grid, _get_extras, auto-reset.  The max-join/correction branch shared with
 *            the generator is exercised by the 100-board golden batch.
 *
 * The code deliberately follows the reference's *formulation* (N private grid copies, max-join,
 * correction masks, cumsum + searchsorted) rather than the wave-parallel formulation used by the
 * HIP kernels, so that the two implementations do not share bugs.
 *
 * Reference files followed (paths relative to /root/reference):
 *   JAXI = routing_board_generation/board_generation_methods/jax_implementation/board_generation
 *   JAXI/parallel_random_walk.py:49-447
 *   routing_board_generation/rl_training/online_generators/parallel_random_walk_generator.py:46-77
 *   routing_board_generation/rl_training/online_generators/uniform_generator.py:70-109
 * Third-party semantics (not vendored in the reference; pinned versions in requirements.txt:4-5):
 *   jax==0.4.8     jax/_src/prng.py (threefry_2x32, threefry_split, threefry_random_bits),
 *                  jax/_src/random.py (uniform, choice, permutation/_shuffle)
 *   jumanji==0.2.2 jumanji/environments/routing/connector/{env,utils,types,constants,reward}.py,
 *                  jumanji/wrappers.py (VmapAutoResetWrapper, MultiToSingleWrapper)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_MAX_AGENTS 64
#define ORACLE_MAX_CELLS 1600 /* 40x40: one _shuffle round holds up to n = 1625 (jax _shuffle) */

/* jumanji connector/constants.py */
enum { EMPTY = 0, PATH = 1, POSITION = 2, TARGET = 3 };
enum { NOOP = 0, UP = 1, RIGHT = 2, DOWN = 3, LEFT = 4 };

/* ------------------------------------------------------------------------------------------
 * jax PRNG (threefry2x32, default impl of jax 0.4.8)
 * ---------------------------------------------------------------------------------------- */
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

/* jax/_src/prng.py::_threefry2x32_lowering -- Threefry-2x32, 20 rounds. */
void oracle_threefry2x32(uint32_t k0, uint32_t k1, uint32_t x0, uint32_t x1, uint32_t *o0,
                         uint32_t *o1) {
  static const int R0[4] = {13, 15, 26, 6};
  static const int R1[4] = {17, 29, 16, 24};
  uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  x0 += ks[0];
  x1 += ks[1];
  for (int g = 0; g < 5; ++g) {
    const int *R = (g & 1) ? R1 : R0;
    for (int r = 0; r < 4; ++r) {
      x0 += x1;
      x1 = rotl32(x1, R[r]);
      x1 ^= x0;
    }
    x0 += ks[(g + 1) % 3];
    x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
  }
  *o0 = x0;
  *o1 = x1;
}

/* jax/_src/prng.py::threefry_random_bits (non-partitionable path) + threefry_2x32's odd padding:
 * counters iota(n), padded with one 0 if n is odd, split in halves (x0 | x1), hashed pairwise,
 * re-concatenated and truncated to n. */
void oracle_random_bits(const uint32_t key[2], int n, uint32_t *out) {
  int half = (n + 1) / 2;
  for (int j = 0; j < half; ++j) {
    uint32_t c0 = (uint32_t)j;
    uint32_t c1 = (half + j < n) ? (uint32_t)(half + j) : 0u; /* the pad word is 0 */
    uint32_t y0, y1;
    oracle_threefry2x32(key[0], key[1], c0, c1, &y0, &y1);
    out[j] = y0;
    if (half + j < n) out[half + j] = y1;
  }
}

/* jax/_src/prng.py::threefry_split: 2*num counters, reshaped (num, 2). */
void oracle_split(const uint32_t key[2], int num, uint32_t *out /* [num][2] */) {
  oracle_random_bits(key, 2 * num, out);
}

/* jax.random.PRNGKey(seed) for the threefry impl: (seed >> 32, seed & 0xffffffff). */
void oracle_prng_key(uint64_t seed, uint32_t key[2]) {
  key[0] = (uint32_t)(seed >> 32);
  key[1] = (uint32_t)(seed & 0xFFFFFFFFu);
}

/* jax/_src/random.py::_uniform, float32, [0, 1): mantissa fill then subtract 1. */
float oracle_uniform(const uint32_t key[2]) {
  uint32_t bits;
  oracle_random_bits(key, 1, &bits);
  uint32_t fb = (bits >> 9) | 0x3F800000u;
  float f;
  memcpy(&f, &fb, 4);
  return f - 1.0f;
}

/* jax/_src/random.py::choice(key, a, shape=(), replace=True, p): cumsum, r = cum[-1]*(1-u),
 * searchsorted(side="left").  Returns the index into `a`. */
int oracle_choice_index_p(const uint32_t key[2], const float *p, int n) {
  float cum[8];
  float acc = 0.0f;
  for (int i = 0; i < n; ++i) {
    acc = acc + p[i];
    cum[i] = acc;
  }
  volatile float one_minus_u = 1.0f - oracle_uniform(key); /* volatile: no FMA contraction */
  float r = cum[n - 1] * one_minus_u;
  int lo = 0;
  while (lo < n && cum[lo] < r) ++lo; /* first i with cum[i] >= r */
  if (lo >= n) lo = n - 1;            /* jnp indexing clamps; unreachable for finite input */
  return lo;
}

typedef struct {
  uint32_t k;
  int32_t v;
} kv_t;

/* stable merge sort by k (lax.sort_key_val is stable: is_stable=True) */
static void kv_sort(kv_t *a, kv_t *tmp, int n) {
  if (n < 2) return;
  int h = n / 2;
  kv_sort(a, tmp, h);
  kv_sort(a + h, tmp, n - h);
  int i = 0, j = h, o = 0;
  while (i < h && j < n) tmp[o++] = (a[j].k < a[i].k) ? a[j++] : a[i++];
  while (i < h) tmp[o++] = a[i++];
  while (j < n) tmp[o++] = a[j++];
  memcpy(a, tmp, sizeof(kv_t) * (size_t)n);
}

/* jax/_src/random.py::_shuffle applied to arange(n): `rounds` rounds of
 *   key, sub = split(key); sort_keys = random_bits(sub, n); stable sort_key_val.
 * rounds = ceil(3*ln(n)/ln(2^32-1)). */
void oracle_permutation(const uint32_t key_in[2], int n, int32_t *perm) {
  uint32_t key[2] = {key_in[0], key_in[1]};
  int rounds = (int)ceil(3.0 * log((double)(n > 1 ? n : 1)) / log(4294967295.0));
  kv_t a[2 * (size_t)n];
  uint32_t bits[n];
  for (int i = 0; i < n; ++i) a[i].v = i;
  for (int r = 0; r < rounds; ++r) {
    uint32_t ks[4];
    oracle_split(key, 2, ks);
    key[0] = ks[0];
    key[1] = ks[1];
    oracle_random_bits(&ks[2], n, bits);
    for (int i = 0; i < n; ++i) a[i].k = bits[i];
    kv_sort(a, a + n, n);
  }
  for (int i = 0; i < n; ++i) perm[i] = a[i].v;
}

/* ------------------------------------------------------------------------------------------
 * jumanji connector helpers (connector/utils.py, types.py)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int32_t id;
  int32_t start[2];
  int32_t target[2];
  int32_t position[2];
} agent_t;

static inline int32_t get_path(int32_t id) { return PATH + 3 * id; }
static inline int32_t get_position(int32_t id) { return POSITION + 3 * id; }
static inline int32_t get_target(int32_t id) { return TARGET + 3 * id; }
/* types.py: Agent.connected = all(position == target) */
static inline int agent_connected(const agent_t *a) {
  return a->position[0] == a->target[0] && a->position[1] == a->target[1];
}
/* python floor division / modulo */
static inline int32_t floordiv(int32_t a, int32_t b) {
  int32_t q = a / b;
  if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
  return q;
}
static inline int32_t floormod(int32_t a, int32_t b) { return a - floordiv(a, b) * b; }

/* utils.py::move_position -- NOOP, UP, RIGHT, DOWN, LEFT */
static void move_position(const int32_t pos[2], int action, int32_t out[2]) {
  static const int dr[5] = {0, -1, 0, 1, 0};
  static const int dc[5] = {0, 0, 1, 0, -1};
  action = action < 0 ? 0 : action > 4 ? 4 : action; /* lax.switch clamps its index */
  out[0] = pos[0] + dr[action];
  out[1] = pos[1] + dc[action];
}

/* utils.py::is_valid_position == JAXI/parallel_random_walk.py:401-429 */
static int is_valid_position(const int32_t *grid, int m, const agent_t *agent,
                             const int32_t pos[2]) {
  int row = pos[0], col = pos[1];
  int in_bounds = (0 <= row) && (row < m) && (0 <= col) && (col < m);
  if (!in_bounds) return 0; /* the gather result is masked by in_bounds in the reference */
  int32_t v = grid[row * m + col];
  int open_cell = (v == EMPTY) || (v == get_target(agent->id));
  return open_cell && !agent_connected(agent);
}

/* utils.py::move_agent: new cell <- head value, old cell <- path value */
static void move_agent(agent_t *agent, int32_t *grid, int m, const int32_t new_pos[2]) {
  grid[new_pos[0] * m + new_pos[1]] = get_position(agent->id);
  grid[agent->position[0] * m + agent->position[1]] = get_path(agent->id);
  agent->position[0] = new_pos[0];
  agent->position[1] = new_pos[1];
}

/* Connector._step_agent == JAXI/parallel_random_walk.py:376-399 (private copy of the grid) */
void oracle_step_agent(const agent_t *agent, const int32_t *grid, int m, int action,
                       agent_t *new_agent, int32_t *new_grid) {
  int32_t new_pos[2];
  *new_agent = *agent;
  memcpy(new_grid, grid, sizeof(int32_t) * (size_t)(m * m));
  move_position(agent->position, action, new_pos);
  if (is_valid_position(grid, m, agent, new_pos) && action != NOOP)
    move_agent(new_agent, new_grid, m, new_pos);
}

/* Connector._step_agents == JAXI/parallel_random_walk.py:119-145:
 * step every agent on a private grid, keep only own cells (get_agent_grid), max-join, then
 * get_correction_mask: an agent whose head value is absent from the joined grid has collided;
 * its previous head cell (now holding its PATH value) gets +1, and its Agent is reverted. */
void oracle_step_agents_with_actions(int32_t *grid, agent_t *agents, int m, int N,
                                     const int32_t *actions) {
  int n = m * m;
  int32_t grids[(size_t)n * (size_t)N]; /* VLA: <= 64*1600*4 B = 400 KiB of stack */
  int32_t joined[n];
  agent_t new_agents[ORACLE_MAX_AGENTS];
  memset(joined, 0, sizeof(int32_t) * (size_t)n);
  for (int i = 0; i < N; ++i)
    oracle_step_agent(&agents[i], grid, m, actions[i], &new_agents[i], grids + (size_t)i * n);
  for (int i = 0; i < N; ++i) { /* get_agent_grid + max over agents */
    int32_t pv = get_position(i), tv = get_target(i), hv = get_path(i);
    const int32_t *g = grids + (size_t)i * n;
    for (int c = 0; c < n; ++c) {
      int32_t v = g[c];
      int32_t own = (v == pv) ? pv : (v == tv) ? tv : (v == hv) ? hv : 0;
      if (own > joined[c]) joined[c] = own;
    }
  }
  for (int i = 0; i < N; ++i) { /* get_correction_mask */
    int32_t pv = get_position(i);
    int present = 0;
    for (int c = 0; c < n; ++c) present |= (joined[c] == pv);
    if (!present) {
      for (int c = 0; c < n; ++c)
        if (grid[c] == pv) joined[c] += 1;
      /* agent reverts: keep agents[i] */
    } else {
      agents[i] = new_agents[i];
    }
  }
  memcpy(grid, joined, sizeof(int32_t) * (size_t)n);
}

/* ------------------------------------------------------------------------------------------
 * ParallelRandomWalkBoard  (JAXI/parallel_random_walk.py)
 * ---------------------------------------------------------------------------------------- */
/* :262-291 -- order [up, down, left, right]; -1 when off-grid or wrapping a row */
void oracle_adjacent_cells(int cell, int rows, int cols, int32_t out[4]) {
  int32_t d[4] = {-rows, rows, -1, 1};
  int32_t cr = floordiv(cell, cols), cc = floormod(cell, cols);
  for (int k = 0; k < 4; ++k) {
    int32_t c = cell + d[k];
    int mask = (0 <= c) && (c < rows * cols);
    int32_t r2 = floordiv(c, cols), c2 = floormod(c, cols);
    mask = mask && ((r2 == cr) || (c2 == cc));
    out[k] = mask ? c : -1;
  }
}

/* jnp gather semantics for grid[divmod(cell, cols)]: negative indices wrap once, then clamp */
static int32_t gather2d(const int32_t *grid, int rows, int cols, int32_t r, int32_t c) {
  if (r < 0) r += rows;
  if (c < 0) c += cols;
  if (r < 0) r = 0;
  if (r >= rows) r = rows - 1;
  if (c < 0) c = 0;
  if (c >= cols) c = cols - 1;
  return grid[r * cols + c];
}

/* :321-338 */
int oracle_is_cell_free(const int32_t *grid, int rows, int cols, int32_t cell) {
  if (cell == -1) return 0;
  return gather2d(grid, rows, cols, floordiv(cell, cols), floormod(cell, cols)) == 0;
}

/* :340-374 -- True when the candidate touches at most one cell of wire `wire_id` */
static int is_cell_not_doubling_back(const int32_t *grid, int rows, int cols, int32_t wire_id,
                                     int32_t cell) {
  int32_t adj[4];
  oracle_adjacent_cells(cell, rows, cols, adj);
  int count = 0;
  for (int k = 0; k < 4; ++k) {
    if (adj[k] == -1) continue;
    int32_t v = gather2d(grid, rows, cols, floordiv(adj[k], cols), floormod(adj[k], cols));
    count += (v == 3 * wire_id + POSITION) || (v == 3 * wire_id + PATH) ||
             (v == 3 * wire_id + TARGET);
  }
  return count <= 1;
}

int oracle_is_cell_not_doubling_back(const int32_t *grid, int rows, int cols, int32_t wire_id,
                                     int32_t cell) {
  return is_cell_not_doubling_back(grid, rows, cols, wire_id, cell);
}

/* :293-319 */
void oracle_available_cells(const int32_t *grid, int rows, int cols, int32_t cell,
                            int32_t out[4]) {
  int32_t adj[4];
  oracle_adjacent_cells(cell, rows, cols, adj);
  int32_t value = gather2d(grid, rows, cols, floordiv(cell, cols), floormod(cell, cols));
  int32_t wire_id = floordiv(value - 1, 3);
  for (int k = 0; k < 4; ++k) {
    int ok = oracle_is_cell_free(grid, rows, cols, adj[k]) &&
             is_cell_not_doubling_back(grid, rows, cols, wire_id, adj[k]);
    out[k] = ok ? adj[k] : -1;
  }
}

/* :238-260 -- delta -> action id; no match -> 0 */
int oracle_action_from_positions(int32_t p1, int32_t p2, int cols) {
  int32_t dr = floordiv(p2, cols) - floordiv(p1, cols);
  int32_t dc = floormod(p2, cols) - floormod(p1, cols);
  if (dr == -1 && dc == 0) return UP;
  if (dr == 1 && dc == 0) return DOWN;
  if (dr == 0 && dc == -1) return LEFT;
  if (dr == 0 && dc == 1) return RIGHT;
  return NOOP;
}

/* :205-230 */
int oracle_select_action(const uint32_t key[2], const int32_t *grid, int m,
                         const agent_t *agent) {
  int32_t cell = agent->position[0] * m + agent->position[1];
  int32_t avail[4];
  float p[4];
  oracle_available_cells(grid, m, m, cell, avail);
  for (int k = 0; k < 4; ++k) p[k] = (avail[k] != -1) ? 1.0f : 0.0f;
  int idx = oracle_choice_index_p(key, p, 4);
  return oracle_action_from_positions(cell, avail[idx], m);
}

/* :192-203 */
int oracle_continue_stepping(const int32_t *grid, const agent_t *agents, int m, int N) {
  for (int i = 0; i < N; ++i) {
    int32_t avail[4];
    oracle_available_cells(grid, m, m, agents[i].position[0] * m + agents[i].position[1],
                           avail);
    for (int k = 0; k < 4; ++k)
      if (avail[k] != -1) return 1;
  }
  return 0;
}

/* :101-145 with random actions */
void oracle_prw_step_agents(const uint32_t key[2], int32_t *grid, agent_t *agents, int m,
                            int N) {
  uint32_t keys[2 * ORACLE_MAX_AGENTS];
  int32_t actions[ORACLE_MAX_AGENTS];
  oracle_split(key, N, keys);
  for (int i = 0; i < N; ++i) actions[i] = oracle_select_action(&keys[2 * i], grid, m, &agents[i]);
  oracle_step_agents_with_actions(grid, agents, m, N, actions);
}

/* :92-99 */
void oracle_prw_step(uint32_t key[2], int32_t *grid, agent_t *agents, int m, int N) {
  uint32_t ks[4];
  oracle_prw_step_agents(key, grid, agents, m, N);
  oracle_split(key, 2, ks);
  key[0] = ks[2];
  key[1] = ks[3];
}

/* :147-186 */
void oracle_initialise_agents(const uint32_t key[2], int32_t *grid, agent_t *agents, int m,
                              int N) {
  int n = m * m;
  int32_t perm[n];
  oracle_permutation(key, n, perm);
  for (int i = 0; i < N; ++i) {
    agents[i].id = i;
    agents[i].start[0] = agents[i].position[0] = floordiv(perm[i], m);
    agents[i].start[1] = agents[i].position[1] = floormod(perm[i], m);
    agents[i].target[0] = agents[i].target[1] = -1;
  }
  /* vmap(place head) then max over agents */
  for (int i = 0; i < N; ++i) {
    int c = agents[i].start[0] * m + agents[i].start[1];
    if (get_position(i) > grid[c]) grid[c] = get_position(i);
  }
}

/* :60-90.  heads/targets are written agent-major [N][2] (the reference returns the transpose). */
int oracle_generate_board(const uint32_t key_in[2], int m, int N, int32_t *heads /*[N][2]*/,
                          int32_t *targets /*[N][2]*/, int32_t *solved_grid /*[m][m]*/) {
  agent_t agents[ORACLE_MAX_AGENTS];
  uint32_t ks[4], step_key[2];
  int n = m * m, iters = 0;
  memset(solved_grid, 0, sizeof(int32_t) * (size_t)n);
  oracle_split(key_in, 2, ks); /* key, step_key = split(key) */
  step_key[0] = ks[2];
  step_key[1] = ks[3];
  oracle_initialise_agents(&ks[0], solved_grid, agents, m, N);
  while (oracle_continue_stepping(solved_grid, agents, m, N)) {
    oracle_prw_step(step_key, solved_grid, agents, m, N);
    ++iters;
  }
  for (int i = 0; i < N; ++i) {
    heads[2 * i] = agents[i].start[0];
    heads[2 * i + 1] = agents[i].start[1];
    targets[2 * i] = agents[i].position[0];
    targets[2 * i + 1] = agents[i].position[1];
  }
  /* :435-447 heads first, then targets (target wins where an agent never moved) */
  for (int i = 0; i < N; ++i) solved_grid[heads[2 * i] * m + heads[2 * i + 1]] = get_position(i);
  for (int i = 0; i < N; ++i) solved_grid[targets[2 * i] * m + targets[2 * i + 1]] = get_target(i);
  return iters;
}

/* ------------------------------------------------------------------------------------------
 * Generators -> Connector State
 * ---------------------------------------------------------------------------------------- */
typedef struct { /* one env; pointers into caller-owned batched buffers */
  int32_t *grid;       /* [m][m] */
  int32_t *step_count; /* [1] */
  int32_t *start;      /* [N][2] */
  int32_t *target;     /* [N][2] */
  int32_t *position;   /* [N][2] */
  uint32_t *key;       /* [2] */
} state_view_t;

/* parallel_random_walk_generator.py:46-77 */
int oracle_prw_generator(const uint32_t key_in[2], int m, int N, state_view_t st,
                         int32_t *solved_grid /* nullable [m][m] */) {
  uint32_t ks[4];
  int n = m * m;
  int32_t scratch[n];
  int32_t *solved = solved_grid ? solved_grid : scratch;
  oracle_split(key_in, 2, ks); /* key, pos_key = split(key) */
  int iters = oracle_generate_board(&ks[0], m, N, st.start, st.target, solved);
  memset(st.grid, 0, sizeof(int32_t) * (size_t)n);
  for (int i = 0; i < N; ++i) st.grid[st.start[2 * i] * m + st.start[2 * i + 1]] = get_position(i);
  for (int i = 0; i < N; ++i) st.grid[st.target[2 * i] * m + st.target[2 * i + 1]] = get_target(i);
  memcpy(st.position, st.start, sizeof(int32_t) * 2 * (size_t)N);
  *st.step_count = 0;
  st.key[0] = ks[0];
  st.key[1] = ks[1];
  return iters;
}

/* uniform_generator.py:70-109: choice(pos_key, arange(m^2), (2, N), replace=False) */
void oracle_uniform_generator(const uint32_t key_in[2], int m, int N, state_view_t st) {
  uint32_t ks[4];
  int n = m * m;
  int32_t perm[n];
  oracle_split(key_in, 2, ks);
  oracle_permutation(&ks[2], n, perm);
  memset(st.grid, 0, sizeof(int32_t) * (size_t)n);
  for (int i = 0; i < N; ++i) {
    st.start[2 * i] = floordiv(perm[i], m);
    st.start[2 * i + 1] = floormod(perm[i], m);
    st.target[2 * i] = floordiv(perm[N + i], m);
    st.target[2 * i + 1] = floormod(perm[N + i], m);
  }
  for (int i = 0; i < N; ++i) st.grid[st.start[2 * i] * m + st.start[2 * i + 1]] = get_position(i);
  for (int i = 0; i < N; ++i) st.grid[st.target[2 * i] * m + st.target[2 * i + 1]] = get_target(i);
  memcpy(st.position, st.start, sizeof(int32_t) * 2 * (size_t)N);
  *st.step_count = 0;
  st.key[0] = ks[0];
  st.key[1] = ks[1];
}

/* ------------------------------------------------------------------------------------------
 * Connector  (jumanji 0.2.2 connector/env.py) -- semantics UNPINNED by the reference, see header
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int8_t *step_type;          /* [1]  0 FIRST, 1 MID, 2 LAST */
  float *reward;              /* [N] */
  float *discount;            /* [N] */
  int32_t *obs_grid;          /* [N][m][m] */
  uint8_t *action_mask;       /* [N][5] */
  int32_t *obs_step_count;    /* [1] */
  int32_t *num_connections;   /* [1] */
  float *ratio_connections;   /* [1] */
  int32_t *total_path_length; /* [1] */
} timestep_view_t;

static void load_agents(state_view_t st, int N, agent_t *agents) {
  for (int i = 0; i < N; ++i) {
    agents[i].id = i;
    for (int d = 0; d < 2; ++d) {
      agents[i].start[d] = st.start[2 * i + d];
      agents[i].target[d] = st.target[2 * i + d];
      agents[i].position[d] = st.position[2 * i + d];
    }
  }
}

/* env.py::_get_action_mask: mask[0] = True; mask[a] = is_valid_position(move(pos, a)) */
void oracle_action_mask(const int32_t *grid, int m, const agent_t *agent, uint8_t mask[5]) {
  mask[0] = 1;
  for (int a = 1; a < 5; ++a) {
    int32_t np_[2];
    move_position(agent->position, a, np_);
    mask[a] = (uint8_t)is_valid_position(grid, m, agent, np_);
  }
}

/* env.py::_obs_from_grid: agent i sees itself as wire 0: ((v - 1 - 3 i) mod 3N) + 1, 0 stays 0 */
void oracle_obs_from_grid(const int32_t *grid, int m, int N, int32_t *obs /*[N][m][m]*/) {
  int n = m * m;
  for (int i = 0; i < N; ++i)
    for (int c = 0; c < n; ++c) {
      int32_t v = grid[c];
      obs[(size_t)i * n + c] = (v == 0) ? 0 : floormod(v - 1 - 3 * i, 3 * N) + 1;
    }
}

/* env.py::_get_extras */
static void get_extras(const int32_t *grid, int m, const agent_t *agents, int N,
                       timestep_view_t ts) {
  int conn = 0, plen = 0;
  for (int i = 0; i < N; ++i) conn += agent_connected(&agents[i]);
  for (int c = 0; c < m * m; ++c) plen += (grid[c] % 3 == PATH);
  *ts.num_connections = conn;
  *ts.ratio_connections = (float)conn / (float)N;
  *ts.total_path_length = plen;
}

static void observe(state_view_t st, int m, int N, timestep_view_t ts) {
  agent_t agents[ORACLE_MAX_AGENTS];
  load_agents(st, N, agents);
  for (int i = 0; i < N; ++i) oracle_action_mask(st.grid, m, &agents[i], ts.action_mask + 5 * i);
  oracle_obs_from_grid(st.grid, m, N, ts.obs_grid);
  *ts.obs_step_count = *st.step_count;
}

/* env.py::reset (state already produced by a generator): restart(...) */
void oracle_connector_reset_from_state(state_view_t st, int m, int N, timestep_view_t ts) {
  agent_t agents[ORACLE_MAX_AGENTS];
  load_agents(st, N, agents);
  observe(st, m, N, ts);
  get_extras(st.grid, m, agents, N, ts);
  *ts.step_type = 0;
  for (int i = 0; i < N; ++i) {
    ts.reward[i] = 0.0f;
    ts.discount[i] = 1.0f;
  }
}

void oracle_connector_reset(const uint32_t key[2], int m, int N, state_view_t st,
                            timestep_view_t ts) {
  oracle_prw_generator(key, m, N, st, NULL);
  oracle_connector_reset_from_state(st, m, N, ts);
}

/* env.py::step (+ wrappers.py::VmapAutoResetWrapper._auto_reset when auto_reset != 0) */
void oracle_connector_step(state_view_t st, const int32_t *action, int m, int N, int time_limit,
                           float timestep_reward, float connected_reward, int auto_reset,
                           timestep_view_t ts) {
  agent_t agents[ORACLE_MAX_AGENTS];
  int was_connected[ORACLE_MAX_AGENTS];
  load_agents(st, N, agents);
  for (int i = 0; i < N; ++i) was_connected[i] = agent_connected(&agents[i]);
  oracle_step_agents_with_actions(st.grid, agents, m, N, action);
  *st.step_count += 1;
  for (int i = 0; i < N; ++i) {
    st.position[2 * i] = agents[i].position[0];
    st.position[2 * i + 1] = agents[i].position[1];
  }
  /* reward.py::DenseRewardFn */
  for (int i = 0; i < N; ++i) {
    int now = agent_connected(&agents[i]);
    float cr = connected_reward * (float)(!was_connected[i] && now);
    float tr = timestep_reward * (float)(!was_connected[i]);
    ts.reward[i] = cr + tr;
  }
  observe(st, m, N, ts);
  get_extras(st.grid, m, agents, N, ts);
  /* utils.py::connected_or_blocked */
  int all_done = 1;
  int done[ORACLE_MAX_AGENTS];
  for (int i = 0; i < N; ++i) {
    const uint8_t *mk = ts.action_mask + 5 * i;
    done[i] = agent_connected(&agents[i]) || !(mk[1] || mk[2] || mk[3] || mk[4]);
    all_done = all_done && done[i];
  }
  int last = all_done || (*st.step_count >= time_limit);
  *ts.step_type = last ? 2 : 1;
  for (int i = 0; i < N; ++i) ts.discount[i] = last ? 0.0f : (done[i] ? 0.0f : 1.0f);
  if (last && auto_reset) {
    /* key, _ = split(state.key); state, reset_ts = env.reset(key); keep reward/discount/
     * step_type/extras of the terminal step, replace the observation only. */
    uint32_t ks[4];
    oracle_split(st.key, 2, ks);
    oracle_prw_generator(&ks[0], m, N, st, NULL);
    observe(st, m, N, ts);
  }
}

/* ------------------------------------------------------------------------------------------
 * Batched entry points (OpenMP over boards) -- used by tests and by bench.py's cpu_baseline
 * ---------------------------------------------------------------------------------------- */
static state_view_t state_at(int b, int m, int N, int32_t *grid, int32_t *step_count,
                             int32_t *start, int32_t *target, int32_t *position, uint32_t *key) {
  state_view_t s;
  s.grid = grid + (size_t)b * m * m;
  s.step_count = step_count + b;
  s.start = start + (size_t)b * N * 2;
  s.target = target + (size_t)b * N * 2;
  s.position = position + (size_t)b * N * 2;
  s.key = key + (size_t)b * 2;
  return s;
}

static timestep_view_t ts_at(int b, int m, int N, int8_t *step_type, float *reward,
                             float *discount, int32_t *obs_grid, uint8_t *action_mask,
                             int32_t *obs_step_count, int32_t *num_connections,
                             float *ratio_connections, int32_t *total_path_length) {
  timestep_view_t t;
  t.step_type = step_type + b;
  t.reward = reward + (size_t)b * N;
  t.discount = discount + (size_t)b * N;
  t.obs_grid = obs_grid + (size_t)b * N * m * m;
  t.action_mask = action_mask + (size_t)b * N * 5;
  t.obs_step_count = obs_step_count + b;
  t.num_connections = num_connections + b;
  t.ratio_connections = ratio_connections + b;
  t.total_path_length = total_path_length + b;
  return t;
}

/* generator == 0: ParallelRandomWalkGenerator, 1: UniformRandomGenerator */
int oracle_generate_batch(const uint32_t *keys, int B, int m, int N, int generator,
                          int32_t *grid, int32_t *step_count, int32_t *start, int32_t *target,
                          int32_t *position, uint32_t *key_out, int32_t *solved /*nullable*/,
                          int32_t *iters /*nullable*/, int threads) {
  if (N > ORACLE_MAX_AGENTS || m * m > ORACLE_MAX_CELLS || N > m * m) return -1;
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads)
  for (int b = 0; b < B; ++b) {
    state_view_t st = state_at(b, m, N, grid, step_count, start, target, position, key_out);
    uint32_t kin[2] = {keys[2 * b], keys[2 * b + 1]}; /* key_out may alias keys */
    int it = 0;
    if (generator == 0)
      it = oracle_prw_generator(kin, m, N, st, solved ? solved + (size_t)b * m * m : NULL);
    else
      oracle_uniform_generator(kin, m, N, st);
    if (iters) iters[b] = it;
  }
  return 0;
}

/* generate_board only (no generator-level split): heads/targets [B][N][2], solved [B][m][m] */
int oracle_generate_board_batch(const uint32_t *keys, int B, int m, int N, int32_t *heads,
                                int32_t *targets, int32_t *solved, int32_t *iters, int threads) {
  if (N > ORACLE_MAX_AGENTS || m * m > ORACLE_MAX_CELLS || N > m * m) return -1;
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads)
  for (int b = 0; b < B; ++b) {
    int it = oracle_generate_board(keys + 2 * b, m, N, heads + (size_t)b * N * 2,
                                   targets + (size_t)b * N * 2, solved + (size_t)b * m * m);
    if (iters) iters[b] = it;
  }
  return 0;
}

int oracle_connector_reset_batch(const uint32_t *keys, int B, int m, int N, int32_t *grid,
                                 int32_t *step_count, int32_t *start, int32_t *target,
                                 int32_t *position, uint32_t *key_out, int8_t *step_type,
                                 float *reward, float *discount, int32_t *obs_grid,
                                 uint8_t *action_mask, int32_t *obs_step_count,
                                 int32_t *num_connections, float *ratio_connections,
                                 int32_t *total_path_length, int threads) {
  if (N > ORACLE_MAX_AGENTS || m * m > ORACLE_MAX_CELLS || N > m * m) return -1;
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads)
  for (int b = 0; b < B; ++b) {
    state_view_t st = state_at(b, m, N, grid, step_count, start, target, position, key_out);
    timestep_view_t ts = ts_at(b, m, N, step_type, reward, discount, obs_grid, action_mask,
                               obs_step_count, num_connections, ratio_connections,
                               total_path_length);
    uint32_t kin[2] = {keys[2 * b], keys[2 * b + 1]};
    oracle_connector_reset(kin, m, N, st, ts);
  }
  return 0;
}

/* observation/extras of an existing state (env._get_action_mask/_obs_from_grid/_get_extras) */
int oracle_connector_observe_batch(int B, int m, int N, int32_t *grid, int32_t *step_count,
                                   int32_t *start, int32_t *target, int32_t *position,
                                   uint32_t *key, int8_t *step_type, float *reward,
                                   float *discount, int32_t *obs_grid, uint8_t *action_mask,
                                   int32_t *obs_step_count, int32_t *num_connections,
                                   float *ratio_connections, int32_t *total_path_length,
                                   int threads) {
  if (N > ORACLE_MAX_AGENTS || m * m > ORACLE_MAX_CELLS) return -1;
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads)
  for (int b = 0; b < B; ++b) {
    state_view_t st = state_at(b, m, N, grid, step_count, start, target, position, key);
    timestep_view_t ts = ts_at(b, m, N, step_type, reward, discount, obs_grid, action_mask,
                               obs_step_count, num_connections, ratio_connections,
                               total_path_length);
    oracle_connector_reset_from_state(st, m, N, ts);
  }
  return 0;
}

int oracle_connector_step_batch(int B, int m, int N, int32_t *grid, int32_t *step_count,
                                int32_t *start, int32_t *target, int32_t *position,
                                uint32_t *key, const int32_t *action, int time_limit,
                                float timestep_reward, float connected_reward, int auto_reset,
                                int8_t *step_type, float *reward, float *discount,
                                int32_t *obs_grid, uint8_t *action_mask, int32_t *obs_step_count,
                                int32_t *num_connections, float *ratio_connections,
                                int32_t *total_path_length, int threads) {
  if (N > ORACLE_MAX_AGENTS || m * m > ORACLE_MAX_CELLS) return -1;
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads)
  for (int b = 0; b < B; ++b) {
    state_view_t st = state_at(b, m, N, grid, step_count, start, target, position, key);
    timestep_view_t ts = ts_at(b, m, N, step_type, reward, discount, obs_grid, action_mask,
                               obs_step_count, num_connections, ratio_connections,
                               total_path_length);
    oracle_connector_step(st, action + (size_t)b * N, m, N, time_limit, timestep_reward,
                          connected_reward, auto_reset, ts);
  }
  return 0;
}

/* Masked-uniform random policy "defined by this project" (the reference's RandomAgent lives in
 * jumanji.training and draws a Gumbel categorical; nothing pins it).  At global step t, agent i of
 * env b picks legal action number floor(threefry2x32(key_b, counter (t, i)).y0 * #legal / 2^32),
 * legal actions in ascending order. */
void oracle_random_actions_batch(const uint32_t *keys /*[B][2]*/, int B, int N, uint32_t t,
                                 const uint8_t *action_mask /*[B][N][5]*/,
                                 int32_t *action /*[B][N]*/) {
  for (int b = 0; b < B; ++b) {
    for (int i = 0; i < N; ++i) {
      const uint8_t *mk = action_mask + ((size_t)b * N + i) * 5;
      uint32_t y0, y1;
      oracle_threefry2x32(keys[2 * b], keys[2 * b + 1], t, (uint32_t)i, &y0, &y1);
      int cnt = 0;
      for (int a = 0; a < 5; ++a) cnt += mk[a] != 0;
      int pick = (int)(((uint64_t)y0 * (uint64_t)cnt) >> 32), a = 0, chosen = 0;
      for (a = 0; a < 5; ++a)
        if (mk[a] && pick-- == 0) {
          chosen = a;
          break;
        }
      action[(size_t)b * N + i] = chosen;
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * SequentialRandomWalkBoard (JAXI/sequential_random_walk.py) -- SURVEY.md 8f rank 3, pinned by
 * the single-board golden of package_evaluation/profiling_generators.ipynb cell 7 (G7).
 *
 * choice(key, a, shape=(), replace=False, p=mask) is jax's Gumbel top-1:
 *   g = -gumbel(key, (n,)) - log(p); argsort(g)[0]      ([3P] jax/_src/random.py::choice)
 * gumbel is monotone in the uniform u, so the winner is the valid entry with the largest
 * u = bits >> 9 (lowest index on ties), evaluated here on the integers.  (float32 rounding of
 * log(-log(u)) could merge two adjacent u into one g and then prefer the lower index; that
 * cannot be reproduced without XLA's exact log and is not attempted -- SURVEY Appendix A.6.)
 * ---------------------------------------------------------------------------------------- */
/* Rounding audit of that shortcut (oracle_seq_rw_rounding_audit below): per board, how many
 * draws had >= 2 candidates, in how many of them evaluating jax's float32 formula
 *   u = max(tiny, bitcast(bits >> 9 | 0x3F800000) - 1);  g = logf(-logf(u));  argmin g, stable
 * with this libm picks another entry than the integer rule, and in how many the runner-up's g
 * (with another u) lies within 4 float32 ulps of the winner's (a differently rounded logf --
 * XLA's is not glibc's -- could then pick it). */
typedef struct {
  int32_t draws, differs, within_4ulp, min_ulps;
} seq_audit_t;

static int32_t float_order(float x) { /* monotone map float -> int */
  int32_t i;
  memcpy(&i, &x, sizeof(i));
  return i < 0 ? (int32_t)0x80000000 - i : i;
}

static float gumbel_key(uint32_t bits) {
  uint32_t w = (bits >> 9) | 0x3F800000u;
  float f;
  memcpy(&f, &w, sizeof(f));
  f -= 1.0f;
  if (f < 1.17549435e-38f) f = 1.17549435e-38f; /* uniform(minval=finfo.tiny) */
  return logf(-logf(f));
}

static int gumbel_top1(const uint32_t key[2], const uint8_t *valid, int n, seq_audit_t *audit) {
  uint32_t bits[ORACLE_MAX_CELLS];
  oracle_random_bits(key, n, bits);
  int best = 0, have = 0;
  uint32_t best_u = 0;
  for (int j = 0; j < n; ++j) {
    if (!valid[j]) continue;
    uint32_t u = bits[j] >> 9;
    if (!have || u > best_u) {
      best = j;
      best_u = u;
      have = 1;
    }
  }
  if (audit && have) {
    int fbest = -1, others = 0;
    float gbest = 0.0f;
    int32_t gap = 0x7FFFFFFF;
    const float gwin = gumbel_key(bits[best]);
    for (int j = 0; j < n; ++j) {
      if (!valid[j]) continue;
      const float g = gumbel_key(bits[j]);
      if (fbest < 0 || g < gbest) { /* stable argsort: the first of equal keys */
        fbest = j;
        gbest = g;
      }
      if (j != best) ++others;
      if (j != best && (bits[j] >> 9) != best_u) { /* (equal u: equal g under any logf) */
        int64_t d = (int64_t)float_order(g) - (int64_t)float_order(gwin);
        if (d < 0) d = -d;
        if (d < (int64_t)gap) gap = (int32_t)d;
      }
    }
    if (others > 0) {
      audit->draws += 1;
      audit->differs += fbest != best;
      audit->within_4ulp += gap <= 4;
      if (gap < audit->min_ulps) audit->min_ulps = gap;
    }
  }
  return best; /* all-invalid: argsort of all-inf keeps index 0 */
}

/* :138-140 available_cells: PRW's rule, padded to rows + 1 entries */
static void seq_available(const int32_t *grid, int m, int32_t cell, int32_t *avail /*[m+1]*/) {
  int32_t a4[4];
  oracle_available_cells(grid, m, m, cell, a4);
  for (int k = 0; k < m + 1; ++k) avail[k] = k < 4 ? a4[k] : -1;
}

/* add_agents (:290-320) for one max_length; returns success, fills first/last cell per wire */
static int seq_add_agents(const uint32_t key_in[2], int m, int N, int max_length, int32_t *grid,
                          int32_t *first /*[N]*/, int32_t *last /*[N]*/, seq_audit_t *audit) {
  int n = m * m, success = 1;
  uint32_t key[2] = {key_in[0], key_in[1]};
  memset(grid, 0, sizeof(int32_t) * (size_t)n);
  for (int w = 0; w < N; ++w) {
    uint32_t ks[4], ps[4];
    oracle_split(key, 2, ks);        /* key, subkey = split(key): the new key is dropped ... */
    oracle_split(&ks[2], 2, ps);     /* pick_start(subkey): key, subkey = split(key) */
    key[0] = ps[0];                  /* ... the tuple carries pick_start's key from here on */
    key[1] = ps[1];
    uint8_t valid[ORACLE_MAX_CELLS];
    int can_start = 0;
    for (int c = 0; c < n; ++c) {
      valid[c] = grid[c] == 0;
      can_start |= valid[c];
    }
    first[w] = last[w] = -1;
    if (!can_start) {
      success = 0;
      continue;
    }
    int cell = gumbel_top1(&ps[2], valid, n, audit);
    grid[cell] = 3 * w + POSITION;
    first[w] = last[w] = cell;
    int moved = 0;
    for (int t = 0; t < max_length; ++t) { /* walk_randomly (:244-288) */
      int32_t avail[64];
      uint8_t ok[64];
      int can_step = 0;
      seq_available(grid, m, last[w], avail);
      for (int k = 0; k < m + 1; ++k) {
        ok[k] = avail[k] != -1;
        can_step |= ok[k];
      }
      if (!can_step) break; /* the grid no longer changes: every later trip is a no-op too */
      moved = 1;
      uint32_t ss[4];
      oracle_split(key, 2, ss); /* one_step: key, subkey = split(key) */
      key[0] = ss[0];
      key[1] = ss[1];
      int pick = gumbel_top1(&ss[2], ok, m + 1, audit);
      int32_t nxt = avail[pick];
      grid[nxt] = 3 * w + TARGET;
      grid[last[w]] = 3 * w + PATH;
      last[w] = nxt;
    }
    grid[first[w]] = 3 * w + POSITION; /* the first cell becomes the head again (:286) */
    success = success && moved;
  }
  return success;
}

/* generate (:322-393) + generate_starts_ends (:395-431).  Returns the number of attempts
 * (1 .. 2m), or 0 when every attempt failed (board all zero, starts = ends = (0, 0)). */
static int seq_rw_generate(const uint32_t key[2], int m, int N, int32_t *board,
                           int32_t *starts, int32_t *ends, seq_audit_t *audit);

int oracle_seq_rw_generate(const uint32_t key[2], int m, int N, int32_t *board /*[m][m]*/,
                           int32_t *starts /*[N][2]*/, int32_t *ends /*[N][2]*/) {
  return seq_rw_generate(key, m, N, board, starts, ends, NULL);
}

static int seq_rw_generate(const uint32_t key[2], int m, int N, int32_t *board,
                           int32_t *starts, int32_t *ends, seq_audit_t *audit) {
  int32_t first[ORACLE_MAX_AGENTS], last[ORACLE_MAX_AGENTS];
  int start_max = 2 * m;
  for (int i = 1; i <= start_max; ++i) {
    if (seq_add_agents(key, m, N, start_max - i, board, first, last, audit)) {
      for (int w = 0; w < N; ++w) {
        starts[2 * w] = first[w] / m;
        starts[2 * w + 1] = first[w] % m;
        ends[2 * w] = last[w] / m;
        ends[2 * w + 1] = last[w] % m;
      }
      return i;
    }
  }
  memset(board, 0, sizeof(int32_t) * (size_t)(m * m));
  memset(starts, 0, sizeof(int32_t) * 2 * (size_t)N);
  memset(ends, 0, sizeof(int32_t) * 2 * (size_t)N);
  return 0;
}

/* SequentialRandomWalkGenerator.__call__
 * (rl_training/online_generators/sequential_random_walk_generator.py:34-66), batched.
 * On total failure all indices are (0, 0); the reference's scatter with duplicate indices is
 * unspecified there -- this project applies the updates in agent order (last agent's target). */
int oracle_seq_rw_generator_batch(const uint32_t *keys, int B, int m, int N, int32_t *grid,
                                  int32_t *step_count, int32_t *start, int32_t *target,
                                  int32_t *position, uint32_t *key_out, int32_t *solved,
                                  int32_t *attempts, int threads) {
  if (N > ORACLE_MAX_AGENTS || m * m > ORACLE_MAX_CELLS || m + 1 > 64) return -1;
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads)
  for (int b = 0; b < B; ++b) {
    state_view_t st = state_at(b, m, N, grid, step_count, start, target, position, key_out);
    uint32_t ks[4];
    int32_t scratch[m * m];
    int32_t *board = solved ? solved + (size_t)b * m * m : scratch;
    oracle_split(keys + 2 * b, 2, ks);
    int tries = oracle_seq_rw_generate(&ks[0], m, N, board, st.start, st.target);
    memset(st.grid, 0, sizeof(int32_t) * (size_t)(m * m));
    for (int i = 0; i < N; ++i) st.grid[st.start[2 * i] * m + st.start[2 * i + 1]] = get_position(i);
    for (int i = 0; i < N; ++i) st.grid[st.target[2 * i] * m + st.target[2 * i + 1]] = get_target(i);
    memcpy(st.position, st.start, sizeof(int32_t) * 2 * (size_t)N);
    *st.step_count = 0;
    st.key[0] = ks[0];
    st.key[1] = ks[1];
    if (attempts) attempts[b] = tries;
  }
  return 0;
}

/* The float32 rounding audit of gumbel_top1 over SequentialRandomWalkGenerator.__call__ of B keys
 * (every attempt of every board, as the reference runs them): per board the number of draws with
 * >= 2 candidates, of draws where this libm's float32 evaluation of jax's formula picks another
 * entry, of draws whose runner-up is within 4 ulps, and the smallest gap in ulps. */
int oracle_seq_rw_rounding_audit(const uint32_t *keys, int B, int m, int N, int32_t *draws,
                                 int32_t *differs, int32_t *within_4ulp, int32_t *min_ulps,
                                 int threads) {
  if (N > ORACLE_MAX_AGENTS || m * m > ORACLE_MAX_CELLS || m + 1 > 64) return -1;
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads)
  for (int b = 0; b < B; ++b) {
    uint32_t ks[4];
    int32_t board[ORACLE_MAX_CELLS], st[2 * ORACLE_MAX_AGENTS], en[2 * ORACLE_MAX_AGENTS];
    seq_audit_t a = {0, 0, 0, 0x7FFFFFFF};
    oracle_split(keys + 2 * b, 2, ks);
    seq_rw_generate(&ks[0], m, N, board, st, en, &a);
    draws[b] = a.draws;
    differs[b] = a.differs;
    within_4ulp[b] = a.within_4ulp;
    min_ulps[b] = a.min_ulps;
  }
  return 0;
}
