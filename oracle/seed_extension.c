/*
 * seed_extension.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (part of libprw_oracle.so).
 *
 * A plain-C CPU restatement of the reference's SeedExtension board generator (SURVEY.md 8f rank 4b,
 * BASELINE configs[4]).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it; nothing under routing_board_generation_amd/ may import, link or call it.
 *
 * Reference files followed (paths relative to /root/reference;
 * JAXI = routing_board_generation/board_generation_methods/jax_implementation):
 *   JAXI/board_generation/seed_extension.py:52-63,93-147,149-227,257-304
 *   JAXI/utils/post_processor_utils_jax.py:24-195 (extend_wires_jax), :200-234, :283-317,
 *                                          :322-369, :373-429
 *   JAXI/utils/grid_utils.py:51-57,59-126,128-178,180-224,242-260,299-346,491-612
 *   JAXI/data_model/wire.py:84-102,126-142
 *   routing_board_generation/rl_training/online_generators/random_seed_generator.py:28-57
 * Third-party semantics (jax==0.4.8, not vendored): jax/_src/random.py::_randint, choice,
 * permutation/_shuffle, uniform; PRNG primitives are those of prw_oracle.c.
 *
 * Parity status
 *   PINNED   by the reference's own executed notebook, package_evaluation/profiling_generators.ipynb
 *            cell 23: vmap(SeedExtensionGenerator(10, 5).__call__) over split(PRNGKey(0), 100).  All
 *            100 x 5 x 2 start coordinates, all 100 x 5 x 2 target coordinates, the 100 State keys
 *            and the 6x6 corner windows of the 100 training grids are reproduced exactly
 *            (tests/test_oracle_goldens.py).  That pins _randint, the seeding, extend_wires_jax
 *            with its random flips, the per-cell key threading, the choice loop and the stopping
 *            rule for the defaults randomness=0.0, two_sided=True, extension_iterations=1.
 *   STALE    the single-board output of the same notebook (cell 10) was produced by an earlier
 *            version of the reference (the notebook still imports the class under its former
 *            name, RandomSeedGenerator): only its State.key and the endpoints that never move
 *            (wire 0 head (6,5), wire 3 target (5,5)) agree with today's reference code.
 *   UNPINNED optimise_wire (the BFS re-routing) cannot change a head or a target cell, so no
 *            golden of the reference sees it; neither do randomness > 0, two_sided=False or
 *            extension_iterations > 1.  These follow the source text only.
 *
 * The code follows the reference's formulation (INVALID-padded candidate lists, a queue array with
 * running-maximum stamps and argmin pops, a -1 terminated path stack) rather than the event-driven
 * formulation of the HIP kernel, so that the two do not share bugs.  Source-level quirks kept:
 *   - get_previous_neighbor_jax scans fori_loop(0, 3): the RIGHT neighbour is never found
 *     (post_processor_utils_jax.py:233);
 *   - prev_layout is captured after the random flip/flop and compared with the un-flipped result
 *     (:66-70, :186-189, :53): a sweep can end the loop only when it was neither flipped nor
 *     flopped (or the board is mirror-symmetric);
 *   - the key advanced inside the choice while-loop is discarded (:127-144); the cell's next key
 *     is split(key)[0] and random_key is split(key)[1], the first choice key (:156);
 *   - every cell, extendable or not, runs the choice loop once and advances the key;
 *   - `~two_sided` on a Python bool is -2/-1 (:110-113); the value of is_extendable comes out as
 *     intended either way, so two_sided is a plain flag here;
 *   - the BFS start cell keeps visited == -1 and is queued a second time by its first neighbour
 *     (grid_utils.py:145-146,171-175); the queue stamps go through float32 (:164-170; exact below
 *     2^24).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { EMPTY = 0, PATH = 1, POSITION = 2, TARGET = 3 };
#define SE_INVALID (-999)
#define SE_MAX_SIDE 40
#define SE_MAX_CELLS (SE_MAX_SIDE * SE_MAX_SIDE)

/* PRNG primitives of prw_oracle.c (same shared object) */
void oracle_threefry2x32(uint32_t k0, uint32_t k1, uint32_t x0, uint32_t x1, uint32_t *o0,
                         uint32_t *o1);
void oracle_random_bits(const uint32_t key[2], int n, uint32_t *out);
void oracle_split(const uint32_t key[2], int num, uint32_t *out);
float oracle_uniform(const uint32_t key[2]);
void oracle_permutation(const uint32_t key_in[2], int n, int32_t *perm);

typedef struct {
  int32_t r, c;
} pos_t;
static const pos_t INVALID_POS = {SE_INVALID, SE_INVALID};
static inline int pos_eq(pos_t a, pos_t b) { return a.r == b.r && a.c == b.c; }

/* [3P] jax/_src/random.py::_randint(key, shape=(n,), minval, maxval, int32):
 *   k1, k2 = split(key); higher_bits, lower_bits = random_bits(k1), random_bits(k2);
 *   span = maxval - minval (1 if maxval <= minval); multiplier = ((2^16 % span)^2) % span;
 *   offset = ((higher % span) * multiplier + lower % span) % span, all in uint32. */
void oracle_randint(const uint32_t key[2], int n, int32_t minval, int32_t maxval, int32_t *out) {
  uint32_t ks[4];
  oracle_split(key, 2, ks);
  uint32_t hi[n > 0 ? n : 1], lo[n > 0 ? n : 1];
  oracle_random_bits(&ks[0], n, hi);
  oracle_random_bits(&ks[2], n, lo);
  uint32_t span = (uint32_t)(maxval - minval);
  if (maxval <= minval) span = 1;
  uint32_t mult = 65536u % span;
  mult = (mult * mult) % span;
  for (int i = 0; i < n; ++i) {
    uint32_t off = (hi[i] % span) * mult + (lo[i] % span);
    out[i] = minval + (int32_t)(off % span);
  }
}

/* ------------------------------------------------------------------------------------------
 * seed_extension.py:93-147  return_seeded_board
 * ---------------------------------------------------------------------------------------- */
static int return_seeded_board(const uint32_t key[2], int m, int N, int32_t *board) {
  int32_t side;
  oracle_randint(key, 1, 0, 2, &side); /* lax.cond(randint(key, (), 0, 2), one_side, other_side) */
  /* :105-112 / :123-130  meshgrid(arange(1, rows, 2), arange(1, cols, 2), "ij") or
   *                      meshgrid(arange(0, rows-1, 2), arange(0, cols-1, 2), "ij"), row-major */
  int first = side ? 1 : 0, stop = side ? m : m - 1;
  pos_t index_choice[SE_MAX_CELLS / 4 + 1];
  int n_choice = 0;
  for (int r = first; r < stop; r += 2)
    for (int c = first; c < stop; c += 2) index_choice[n_choice++] = (pos_t){r, c};
  if (N > n_choice) return -1; /* jax: "Cannot take a larger sample than population" */
  /* :113-115  choice(key, index_choice, (N,), replace=False) = take(a, permutation(key, n)[:N]) */
  int32_t perm[n_choice];
  oracle_permutation(key, n_choice, perm);
  /* :116-118  choice(key, offset_array, (N,)) = take(offset_array, randint(key, (N,), 0, 2)) */
  static const pos_t off_one[2] = {{-1, 0}, {0, -1}};
  static const pos_t off_other[2] = {{0, 1}, {1, 0}};
  int32_t which[N];
  oracle_randint(key, N, 0, 2, which);
  memset(board, 0, sizeof(int32_t) * (size_t)(m * m));
  pos_t heads[N], tails[N];
  for (int i = 0; i < N; ++i) {
    heads[i] = index_choice[perm[i]];
    pos_t o = side ? off_one[which[i]] : off_other[which[i]];
    tails[i] = (pos_t){heads[i].r + o.r, heads[i].c + o.c};
  }
  /* :145-146  xs (head_indices) get TARGET, then ys (tail_indices) get POSITION */
  for (int i = 0; i < N; ++i) board[heads[i].r * m + heads[i].c] = 3 * i + TARGET;
  for (int i = 0; i < N; ++i) board[tails[i].r * m + tails[i].c] = 3 * i + POSITION;
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * post_processor_utils_jax.py helpers
 * ---------------------------------------------------------------------------------------- */
/* :396-407 */
static int32_t cell_encoding_to_wire_num(int32_t v) { return v == 0 ? -1 : (v - 1) / 3; }
/* :373-392  (-1 when out of bounds) */
static int32_t position_to_wire_num(pos_t p, const int32_t *board, int m) {
  if (p.r < 0 || p.r >= m || p.c < 0 || p.c >= m) return -1;
  return cell_encoding_to_wire_num(board[p.r * m + p.c]);
}
/* :411-429 */
static int32_t position_to_cell_type(pos_t p, const int32_t *board, int m) {
  int32_t v = board[p.r * m + p.c];
  return v == 0 ? 0 : ((v - 1) % 3) + 1;
}
/* :322-369  order: above, left, below, right */
static void get_open_adjacent_cells(pos_t p, const int32_t *board, int m, pos_t out[4]) {
  pos_t cand[4] = {{p.r - 1, p.c}, {p.r, p.c - 1}, {p.r + 1, p.c}, {p.r, p.c + 1}};
  int inb[4] = {cand[0].r >= 0, cand[1].c >= 0, cand[2].r < m, cand[3].c < m};
  for (int k = 0; k < 4; ++k)
    out[k] = (inb[k] && board[cand[k].r * m + cand[k].c] == EMPTY) ? cand[k] : INVALID_POS;
}
/* :283-317 */
static int num_wire_adjacencies(pos_t p, int32_t wire_num, const int32_t *board, int m) {
  pos_t nb[4] = {{p.r - 1, p.c}, {p.r + 1, p.c}, {p.r, p.c - 1}, {p.r, p.c + 1}};
  int n = 0;
  for (int k = 0; k < 4; ++k) n += position_to_wire_num(nb[k], board, m) == wire_num;
  return n;
}
/* :200-234  fori_loop(0, 3): above, below, left; the last match wins */
static pos_t get_previous_neighbor(pos_t p, const int32_t *board, int m) {
  int32_t w = position_to_wire_num(p, board, m);
  pos_t nb[4] = {{p.r - 1, p.c}, {p.r + 1, p.c}, {p.r, p.c - 1}, {p.r, p.c + 1}};
  pos_t out = INVALID_POS;
  for (int i = 0; i < 3; ++i) {
    int inb = nb[i].r >= 0 && nb[i].r < m && nb[i].c >= 0 && nb[i].c < m;
    if (position_to_wire_num(nb[i], board, m) == w && inb) out = nb[i];
  }
  return out;
}

static void flip_rows(int32_t *b, int m) {
  for (int r = 0; r < m / 2; ++r)
    for (int c = 0; c < m; ++c) {
      int32_t t = b[r * m + c];
      b[r * m + c] = b[(m - 1 - r) * m + c];
      b[(m - 1 - r) * m + c] = t;
    }
}
static void flip_cols(int32_t *b, int m) {
  for (int r = 0; r < m; ++r)
    for (int c = 0; c < m / 2; ++c) {
      int32_t t = b[r * m + c];
      b[r * m + c] = b[r * m + m - 1 - c];
      b[r * m + m - 1 - c] = t;
    }
}

/* jax.random.choice(key, jnp.array([True, False])) = a[randint(key, (), 0, 2)] */
static int choice_true_false(const uint32_t key[2]) {
  int32_t ind;
  oracle_randint(key, 1, 0, 2, &ind);
  return ind == 0;
}

/* ------------------------------------------------------------------------------------------
 * post_processor_utils_jax.py:24-195  extend_wires_jax.  Returns the number of while iterations.
 * ---------------------------------------------------------------------------------------- */
static int extend_wires(int32_t *board, int m, const uint32_t key_in[2], float randomness,
                        int two_sided, double extension_steps, int *live_sweeps) {
  const int cells = m * m;
  uint32_t key[2] = {key_in[0], key_in[1]};
  int32_t prev[SE_MAX_CELLS];
  memcpy(prev, board, sizeof(int32_t) * (size_t)cells);
  prev[0] += 1; /* :47 */
  int32_t step_num = 0;
  /* :51-55  int32 step_num against the float32 extension_steps */
  while (memcmp(prev, board, sizeof(int32_t) * (size_t)cells) != 0 &&
         (float)step_num < (float)extension_steps) {
    step_num += 1;
    int32_t before[SE_MAX_CELLS]; /* (diagnostic only: did this sweep change the board?) */
    memcpy(before, board, sizeof(int32_t) * (size_t)cells);
    uint32_t ks[6];
    oracle_split(key, 3, ks); /* :63 key, flipkey, flopkey */
    key[0] = ks[0];
    key[1] = ks[1];
    int do_flip = choice_true_false(&ks[2]);
    int do_flop = choice_true_false(&ks[4]);
    if (do_flip) flip_rows(board, m);
    if (do_flop) flip_cols(board, m);
    memcpy(prev, board, sizeof(int32_t) * (size_t)cells); /* :70 (after the flip) */

    for (int row = 0; row < m; ++row) {
      for (int col = 0; col < m; ++col) {
        pos_t cur = {row, col};
        pos_t poss[4];
        get_open_adjacent_cells(cur, board, m, poss); /* :80-82 */
        int32_t cur_wire = position_to_wire_num(cur, board, m);
        for (int k = 0; k < 4; ++k) /* :86-97 */
          if (num_wire_adjacencies(poss[k], cur_wire, board, m) > 1) poss[k] = INVALID_POS;
        pos_t prev_nb = get_previous_neighbor(cur, board, m); /* :100 */
        pos_t priority = {row + (row - prev_nb.r), col + (col - prev_nb.c)};
        int32_t cell_type = position_to_cell_type(cur, board, m);
        int is_extendable = two_sided ? (cell_type == POSITION || cell_type == TARGET)
                                      : (cell_type == TARGET); /* :110-113 */
        if (!is_extendable)
          for (int k = 0; k < 4; ++k) poss[k] = INVALID_POS;
        int stop_extension = 1; /* :119 */
        for (int k = 0; k < 4; ++k) stop_extension &= pos_eq(poss[k], INVALID_POS);
        if (stop_extension)
          for (int k = 0; k < 4; ++k) poss[k] = cur;

        /* :127-144  draw until a valid entry comes up; the loop's key is local */
        pos_t extension_pos = INVALID_POS;
        uint32_t lk[2] = {key[0], key[1]};
        while (pos_eq(extension_pos, INVALID_POS)) {
          uint32_t s[4];
          oracle_split(lk, 2, s);
          lk[0] = s[0];
          lk[1] = s[1];
          int32_t ind;
          oracle_randint(&s[2], 1, 0, 4, &ind); /* choice(choice_key, poss_extension_list) */
          extension_pos = poss[ind];
        }
        int priority_available = 0; /* :146-152 */
        for (int k = 0; k < 4; ++k) priority_available |= pos_eq(poss[k], priority);
        uint32_t s[4];
        oracle_split(key, 2, s); /* :156 key, random_key */
        key[0] = s[0];
        key[1] = s[1];
        int use_random = randomness > oracle_uniform(&s[2]);
        if (priority_available && !use_random) extension_pos = priority;
        /* :165-173 */
        board[extension_pos.r * m + extension_pos.c] = board[cur.r * m + cur.c];
        cell_type = position_to_cell_type(cur, board, m);
        board[cur.r * m + cur.c] += stop_extension ? 0 : PATH - cell_type;
      }
    }
    if (do_flip) flip_rows(board, m); /* :186-187 */
    if (do_flop) flip_cols(board, m);
    if (live_sweeps && memcmp(before, board, sizeof(int32_t) * (size_t)cells) != 0)
      *live_sweeps += 1;
  }
  return step_num;
}

/* ------------------------------------------------------------------------------------------
 * grid_utils.py  Grid + optimise_wire
 * ---------------------------------------------------------------------------------------- */
/* :59-126 with :128-178 inlined for use_empty=True */
static void update_queue_and_visited(int32_t *queue, int32_t *visited, const int32_t *board, int m,
                                     const int32_t perm[4], int fill_num) {
  const int cells = m * m;
  /* :81  argmin(where(queue > 0, queue, inf)) -- first minimum; 0 when nothing is queued */
  int curr_int = 0;
  float best = INFINITY;
  for (int i = 0; i < cells; ++i) {
    float v = queue[i] > 0 ? (float)queue[i] : INFINITY;
    if (v < best) {
      best = v;
      curr_int = i;
    }
  }
  int cr = curr_int / m, cc = curr_int % m;
  static const int row[4] = {-1, 0, 1, 0}, col[4] = {0, 1, 0, -1}; /* :87-88 */
  for (int j = 0; j < 4; ++j) {
    int nr = cr + row[perm[j]], nc = cc + col[perm[j]];
    int pos_int = nr * m + nc;
    int size_cond = 0 <= nr && nr < m && 0 <= nc && nc < m;
    if (!size_cond) continue; /* the other conditions are and-ed with it */
    int cond_1 = visited[pos_int] == -1;
    int cond_2 = queue[pos_int] == 0;
    int32_t b = board[nr * m + nc];
    int cond_3 = (b == 3 * fill_num + PATH) || (b == EMPTY);
    if (cond_1 && cond_2 && cond_3) {
      float curr_val = -INFINITY; /* :164 max(where(queue > 0, queue, -inf)) */
      for (int i = 0; i < cells; ++i)
        if (queue[i] > 0 && (float)queue[i] > curr_val) curr_val = (float)queue[i];
      /* unreachable with an empty queue (curr_int itself is queued while its pop runs) */
      queue[pos_int] = isfinite(curr_val) ? (int32_t)(curr_val + 1.0f) : INT32_MIN;
      visited[pos_int] = curr_int;
    }
  }
  queue[curr_int] = 0; /* :125 */
}

/* :491-612.  Returns 0, or -2 if the walk back from the end never meets the start (the
 * reference would not terminate; unreachable because a wire is a connected chain). */
static int optimise_wire(const uint32_t key[2], int32_t *board, int m, int wire_num) {
  const int cells = m * m;
  int32_t start_num = 3 * wire_num + POSITION, end_num = 3 * wire_num + TARGET;
  /* :551-555  argmax(where(board == num, board, 0)): first match, 0 when absent */
  int flat_start = 0, flat_end = 0;
  for (int i = cells - 1; i >= 0; --i) {
    if (board[i] == start_num) flat_start = i;
    if (board[i] == end_num) flat_end = i;
  }
  board[flat_start] = 3 * wire_num + 1; /* :566-568 */
  board[flat_end] = 3 * wire_num + 1;
  int32_t queue[SE_MAX_CELLS], visited[SE_MAX_CELLS];
  for (int i = 0; i < cells; ++i) {
    queue[i] = 0;
    visited[i] = -1;
  }
  queue[flat_start] = 1; /* :574 */
  /* :91  permutation(key, arange(4), independent=True): the same key, hence the same order of
   * the four moves, in every iteration of the search */
  int32_t perm[4];
  oracle_permutation(key, 4, perm);
  int i = 0, end_reached = 0;
  while (i < cells && !end_reached) { /* :593-603 */
    update_queue_and_visited(queue, visited, board, m, perm, wire_num);
    end_reached = visited[flat_end] != -1;
    ++i;
  }
  /* :187-224  get_path: push from the end along visited[] until the start has been pushed */
  int32_t path[SE_MAX_CELLS + 1];
  int ins = 0;
  int32_t cur = flat_end;
  for (int k = 0; k <= cells; ++k) path[k] = -1;
  while (path[ins > 0 ? ins - 1 : 0] != flat_start) {
    if (ins >= cells || cur < 0) return -2;
    path[ins++] = cur;
    cur = visited[cur];
  }
  /* :242-260  remove_path */
  for (int k = 0; k < cells; ++k)
    if (board[k] == 3 * wire_num + PATH) board[k] = 0;
  /* :326-346  jax_fill_grid */
  int n = 0;
  while (path[n] != -1) {
    board[path[n]] = 3 * wire_num + PATH;
    ++n;
  }
  board[path[n - 1]] = 3 * wire_num + POSITION;
  board[path[0]] = 3 * wire_num + TARGET;
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * seed_extension.py:149-227  return_solved_board.  sweeps (nullable) = total while iterations
 * of extend_wires_jax over all extension iterations; live_sweeps (nullable) = those of them that
 * changed the board.  (Once a sweep changes nothing, no later one can -- which cells an endpoint
 * may extend into does not depend on the frame -- and the reference only keeps drawing flips until
 * an un-flipped sweep lets its comparison succeed: the HIP kernel stops there, see rbg_seedext.h.)
 * ---------------------------------------------------------------------------------------- */
int oracle_seed_extension_solved_board(const uint32_t key_in[2], int m, int N, float randomness,
                                       int two_sided, int extension_iterations,
                                       double extension_steps, int32_t *board, int32_t *sweeps,
                                       int32_t *live_sweeps) {
  if (m < 2 || m > SE_MAX_SIDE || N < 1) return -1;
  uint32_t ks[6];
  oracle_split(key_in, 2, ks); /* :171 key, seedkey */
  uint32_t key[2] = {ks[0], ks[1]};
  if (return_seeded_board(&ks[2], m, N, board) != 0) return -1;
  int wires_on_board = N > (m * m) / 3 ? (m * m) / 3 : N; /* :60-63 */
  int total = 0, live = 0;
  for (int it = 0; it < extension_iterations; ++it) { /* :180-224, converged stays False */
    oracle_split(key, 3, ks); /* :189 key, extkey, optkey */
    key[0] = ks[0];
    key[1] = ks[1];
    total += extend_wires(board, m, &ks[2], randomness, two_sided, extension_steps, &live);
    uint32_t optkeys[2 * (size_t)wires_on_board];
    oracle_split(&ks[4], wires_on_board, optkeys); /* :194 */
    for (int w = 0; w < wires_on_board; ++w) /* :199-208 */
      if (optimise_wire(&optkeys[2 * w], board, m, w) != 0) return -2;
  }
  if (sweeps) *sweeps = total;
  if (live_sweeps) *live_sweeps = live; /* sweeps that changed the board */
  return 0;
}

/* seed_extension.py:257-304 generate_starts_ends: argwhere(board == v, size=2)[0], i.e. the first
 * match in row-major order, (0, 0) when there is none. */
static void find_first(const int32_t *board, int m, int32_t v, int32_t out[2]) {
  out[0] = 0;
  out[1] = 0;
  for (int i = 0; i < m * m; ++i)
    if (board[i] == v) {
      out[0] = i / m;
      out[1] = i % m;
      return;
    }
}

/* split_first == 0: SeedExtensionBoard(m, m, N).generate_starts_ends(key, ...) (+ the solved board)
 * split_first != 0: SeedExtensionGenerator(m, N).__call__(key) (random_seed_generator.py:28-57):
 *                   key, _ = split(key); State.key = key; State.grid = heads then targets.
 * start/target are agent-major [N][2]. */
int oracle_seed_extension_generate(const uint32_t key_in[2], int m, int N, int split_first,
                                   float randomness, int two_sided, int extension_iterations,
                                   double extension_steps, int32_t *solved, int32_t *train_grid,
                                   int32_t *start, int32_t *target, uint32_t *key_out,
                                   int32_t *sweeps, int32_t *live_sweeps) {
  uint32_t key[2] = {key_in[0], key_in[1]};
  if (split_first) {
    uint32_t ks[4];
    oracle_split(key_in, 2, ks);
    key[0] = ks[0];
    key[1] = ks[1];
  }
  int32_t board[SE_MAX_CELLS];
  int rc = oracle_seed_extension_solved_board(key, m, N, randomness, two_sided,
                                              extension_iterations, extension_steps, board, sweeps,
                                              live_sweeps);
  if (rc != 0) return rc;
  for (int w = 0; w < N; ++w) {
    find_first(board, m, 3 * w + POSITION, &start[2 * w]);
    find_first(board, m, 3 * w + TARGET, &target[2 * w]);
  }
  if (solved) memcpy(solved, board, sizeof(int32_t) * (size_t)(m * m));
  if (train_grid) { /* random_seed_generator.py:36-44 */
    memset(train_grid, 0, sizeof(int32_t) * (size_t)(m * m));
    for (int w = 0; w < N; ++w) train_grid[start[2 * w] * m + start[2 * w + 1]] = 3 * w + POSITION;
    for (int w = 0; w < N; ++w) train_grid[target[2 * w] * m + target[2 * w + 1]] = 3 * w + TARGET;
  }
  if (key_out) {
    key_out[0] = key[0];
    key_out[1] = key[1];
  }
  return 0;
}

/* vmap over keys [B][2]; State arrays as in prw_oracle.c's batch entry points. */
int oracle_seed_extension_batch(const uint32_t *keys, int B, int m, int N, int split_first,
                                float randomness, int two_sided, int extension_iterations,
                                double extension_steps, int32_t *grid, int32_t *step_count,
                                int32_t *start, int32_t *target, int32_t *position,
                                uint32_t *key_out, int32_t *solved, int32_t *sweeps,
                                int32_t *live_sweeps, int threads) {
  if (B < 0 || m < 2 || m > SE_MAX_SIDE || N < 1 || N > (m / 2) * (m / 2)) return -1;
  int bad = 0;
  (void)threads;
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads > 0 ? threads : 1) reduction(| : bad)
  for (int b = 0; b < B; ++b) {
    int rc = oracle_seed_extension_generate(
        &keys[2 * b], m, N, split_first, randomness, two_sided, extension_iterations,
        extension_steps, solved ? &solved[(size_t)b * m * m] : NULL,
        grid ? &grid[(size_t)b * m * m] : NULL, &start[(size_t)b * N * 2],
        &target[(size_t)b * N * 2], key_out ? &key_out[2 * b] : NULL, sweeps ? &sweeps[b] : NULL,
        live_sweeps ? &live_sweeps[b] : NULL);
    bad |= rc != 0;
    if (position)
      memcpy(&position[(size_t)b * N * 2], &start[(size_t)b * N * 2], sizeof(int32_t) * 2 * N);
    if (step_count) step_count[b] = 0;
  }
  return bad ? -2 : 0;
}
