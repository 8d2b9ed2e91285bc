/*
 * rbg_prw.h -- C-ABI of librbg_prw.so: the MI355X (gfx950) implementation of the reference's
 * ParallelRandomWalk board generator and the coupled Jumanji Connector reset/step hot path.
 *
 * The reference is pure Python/JAX and has no FFI of its own; each entry point below names the
 * reference interface it replaces (paths relative to the reference root;
 * JAXI = routing_board_generation/board_generation_methods/jax_implementation/board_generation,
 * [3P] = the un-vendored pip dependency that holds the behaviour, pinned in requirements.txt:4-5).
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned HBM; nothing is allocated or freed here
 *   - every call is asynchronous on the caller-supplied `stream` (a hipStream_t passed as void*;
 *     NULL = the default stream); no hidden synchronisation, no global state, re-entrant,
 *     capturable into a hipGraph
 *   - arrays are dense, C-contiguous, batch-major:
 *       keys        uint32 [B][2]          jax.random key data (Threefry-2x32)
 *       grid        int32  [B][m][m]       0 empty, 1+3i path, 2+3i head/position, 3+3i target
 *       start/target/position int32 [B][N][2]   (row, col) per agent
 *       obs_grid    int32  [B][N][m][m]    per-agent view (_obs_from_grid)
 *       action_mask uint8  [B][N][5]       NOOP, UP, RIGHT, DOWN, LEFT
 *   - limits: 2 <= m <= 40 (one _shuffle round, as in jax for m*m <= 1625), 1 <= N <= 64,
 *     N <= m*m (2N <= m*m for the uniform generator)
 *   - return value: 0 ok; < 0 argument error (RBG_E_*); > 0 a hipError_t from the launch.
 *     rbg_last_error() gives a thread-local message for the last non-zero return.
 */
#ifndef RBG_PRW_H_
#define RBG_PRW_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RBG_ABI_VERSION 9

#define RBG_E_NULL (-1)   /* a required pointer is NULL */
#define RBG_E_SHAPE (-2)  /* B, m, N or num out of range */
#define RBG_E_PARAM (-3)  /* bad flag / parameter value */

/* [3P] jumanji.environments.routing.connector.types.State (field order grid, step_count, agents,
 * key as printed at package_evaluation/profiling_generators.ipynb cell 13); Agent.id is implicit
 * (arange(N)). */
typedef struct rbg_state {
  int32_t *grid;       /* [B][m][m] */
  int32_t *step_count; /* [B] */
  int32_t *start;      /* [B][N][2] */
  int32_t *target;     /* [B][N][2] */
  int32_t *position;   /* [B][N][2] */
  uint32_t *key;       /* [B][2] */
} rbg_state;

/* [3P] jumanji.types.TimeStep[Observation] + extras (keys confirmed by
 * routing_board_generation/benchmarking/utils/benchmark_data_model.py:24-29).
 * Any member may be NULL: that output is then skipped (e.g. obs_grid for a random policy). */
typedef struct rbg_timestep {
  int8_t *step_type;          /* [B]  0 FIRST, 1 MID, 2 LAST */
  float *reward;              /* [B][N] */
  float *discount;            /* [B][N] */
  int32_t *obs_grid;          /* [B][N][m][m] */
  uint8_t *action_mask;       /* [B][N][5] */
  int32_t *obs_step_count;    /* [B] */
  int32_t *num_connections;   /* [B] */
  float *ratio_connections;   /* [B] */
  int32_t *total_path_length; /* [B] */
  /* [3P] jumanji.wrappers.MultiToSingleWrapper (setup_train.py:158-161) applied inside the
   * launch: reward summed over the agents in float32, agent 0 first ((0 + r_0) + r_1) + ...,
   * and the largest per-agent discount. */
  float *reward_total;        /* [B] */
  float *discount_max;        /* [B] */
} rbg_timestep;

/* [3P] Connector.__init__(time_limit=50) + DenseRewardFn(timestep_reward, connected_reward) +
 * VmapAutoResetWrapper (auto_reset != 0 fuses its reset into the step). */
typedef struct rbg_env_params {
  int32_t time_limit;
  float timestep_reward;
  float connected_reward;
  int32_t auto_reset;
} rbg_env_params;

int rbg_abi_version(void);
const char *rbg_version(void);
const char *rbg_last_error(void);

/* ---- PRNG plumbing ([3P] jax.random.split, jax/_src/prng.py::threefry_split) -------------- */

/* split(key, num) for ONE host-side key: out [num][2].
 * Replaces `jax.random.split(jax.random.PRNGKey(0), batch_size)`
 * (package_evaluation/profiling_generators.ipynb cell 16;
 *  routing_board_generation/rl_training/setup_train.py:397). */
int rbg_split(uint32_t key0, uint32_t key1, int num, uint32_t *out, void *stream);

/* vmap(split)(keys, num): keys [B][2] -> out [B][num][2]. */
int rbg_split_batch(const uint32_t *keys, int B, int num, uint32_t *out, void *stream);

/* ---- board generation -------------------------------------------------------------------- */

/* vmap(ParallelRandomWalkBoard(m, m, N).generate_board)(keys)          [split_first == 0]
 *   JAXI/parallel_random_walk.py:60-90
 * vmap(ParallelRandomWalkGenerator(m, N).__call__)(keys)               [split_first != 0]
 *   routing_board_generation/rl_training/online_generators/parallel_random_walk_generator.py:46-77
 *   (adds `key, _ = split(key)`; key_out = State.key)
 * Outputs: solved_grid (nullable) = generate_board's 3rd result; train_grid (nullable) = the
 * State.grid built at parallel_random_walk_generator.py:63-64 (heads and targets only);
 * start = heads, target = final positions, both agent-major [B][N][2]; position (nullable,
 * = start) and step_count (nullable, = 0) complete the State of :68-77; key_out (nullable);
 * iters (nullable) = while_loop trip count per board. */
int rbg_prw_generate(const uint32_t *keys, int B, int m, int N, int split_first,
                     int32_t *solved_grid, int32_t *train_grid, int32_t *start, int32_t *target,
                     int32_t *position, int32_t *step_count, uint32_t *key_out, int32_t *iters,
                     void *stream);

/* vmap(ParallelRandomWalkGenerator(m, N, stream="philox").__call__)(keys) -- OPT-IN, defined by
 * this project: the walk of rbg_prw_generate driven by a counter-based Philox-4x32-10 stream
 * instead of jax's Threefry key splitting (north_star's "Philox kernel replacing jax.random";
 * SURVEY.md 8a-PRNG).  NOT the reference's random process: boards are valid and deterministic per
 * key but differ from the reference's for the same key, and the only checker is this project's
 * own CPU restatement of the stream (tests).  The stream: Philox-4x32-10 (Salmon et al., SC'11;
 * M0 = 0xD2511F53, M1 = 0xCD9E8D57, W0 = 0x9E3779B9, W1 = 0xBB67AE85), key = the two words of the
 * board's key, counter (c0, c1, c2, c3) with a domain tag in c3:
 *   tag 1  start cells: the sort key of cell j is word j % 4 of philox(key, (j / 4, 0, 0, 1));
 *          the N cells with the smallest (sort key, j), ascending, are the starts of agents 0..N-1;
 *   tag 2  the walk: agent i's uniform at iteration t is word t % 4 of
 *          philox(key, (t / 4, i, 0, 2)) through jax's float32 mantissa fill, and the choice is
 *          jax's (cumsum of the availability flags, r = cum[-1] * (1 - u), first cum >= r);
 *   tag 3  State.key (key_out) = words 0, 1 of philox(key, (0, 0, 0, 3)).
 * Everything else (availability rule, max-join collision rule, head/target encodings, the
 * training grid) is rbg_prw_generate's.  Same outputs as rbg_prw_generate(split_first = 1). */
int rbg_prw_generate_philox(const uint32_t *keys, int B, int m, int N, int32_t *solved_grid,
                            int32_t *train_grid, int32_t *start, int32_t *target,
                            int32_t *position, int32_t *step_count, uint32_t *key_out,
                            int32_t *iters, void *stream);

/* vmap(SequentialRandomWalkBoard(m, m, N).generate)(keys)                   [split_first == 0]
 *   JAXI/sequential_random_walk.py:322-431 (generate + generate_starts_ends)
 * vmap(SequentialRandomWalkGenerator(m, N).__call__)(keys)                  [split_first != 0]
 *   routing_board_generation/rl_training/online_generators/sequential_random_walk_generator.py:34-66
 * Same outputs as rbg_prw_generate; attempts (nullable) int32 [B] = the retry that succeeded
 * (1 .. 2m; 0 = every attempt failed: board all zero, all starts/targets (0, 0)).  m >= 3. */
int rbg_seq_rw_generate(const uint32_t *keys, int B, int m, int N, int split_first,
                        int32_t *solved_grid, int32_t *train_grid, int32_t *start,
                        int32_t *target, int32_t *position, int32_t *step_count,
                        uint32_t *key_out, int32_t *attempts, void *stream);

/* vmap(SeedExtensionBoard(m, m, N).generate_starts_ends)(keys, randomness, two_sided,
 *      extension_iterations, extension_steps)                               [split_first == 0]
 *   JAXI/seed_extension.py:257-304 (and :149-227 return_solved_board for solved_grid); the call
 *   shape of routing_board_generation/rl_training/offline_generation/dataset_generator_jax.py:79-98
 * vmap(SeedExtensionGenerator(m, N).__call__)(keys)                          [split_first != 0]
 *   routing_board_generation/rl_training/online_generators/random_seed_generator.py:28-57
 *   (adds `key, _ = split(key)`; key_out = State.key; the reference's defaults there are
 *    randomness = 0, two_sided = 1, extension_iterations = 1, no step limit)
 * The algorithm: seed N adjacent (target, head) pairs (seed_extension.py:93-147), extend both
 * ends cell by cell over randomly flipped row-major sweeps until an un-flipped sweep changes
 * nothing (JAXI/../utils/post_processor_utils_jax.py:24-195), then re-route every wire along a
 * BFS-shortest path (JAXI/../utils/grid_utils.py:491-612), extension_iterations times.
 *   randomness            float32: P(ignore the straight-ahead direction) per extension
 *   two_sided             != 0: heads and targets extend; 0: targets only
 *   max_extension_steps   sweeps allowed per extension (the reference's `extension_steps`, whose
 *                         default 1e23 means no limit: pass INT32_MAX)
 * Same outputs as rbg_prw_generate (start = the wires' POSITION cells, target = their TARGET
 * cells); solved_grid, when given, also triggers the final re-routing pass, which no other
 * output can see.  sweeps (nullable) int32 [B] = extension sweeps that changed the board (a
 * board none of whose ends can extend is finished; the reference sweeps it a few more times,
 * until an un-flipped sweep lets its stopping comparison succeed, with no effect on any output --
 * so this is NOT the reference's `step_num`, and max_extension_steps counts these live sweeps).
 * The sign bit of sweeps[b] is set if a random-direction try loop of board b ran 512 tries
 * without a valid direction and took the lowest valid one instead (probability < 1e-63 per
 * extension with jax's stream; a set bit means the board is not the reference's).
 * extension_iterations == -1 (with split_first == 0): return_seeded_board(keys)
 * (seed_extension.py:93-147) -- the keys are the seed keys themselves and nothing is extended.
 * N <= (m / 2)^2 (the candidate seed cells). */
int rbg_seed_extension_generate(const uint32_t *keys, int B, int m, int N, int split_first,
                                float randomness, int two_sided, int extension_iterations,
                                int max_extension_steps, int32_t *solved_grid, int32_t *train_grid,
                                int32_t *start, int32_t *target, int32_t *position,
                                int32_t *step_count, uint32_t *key_out, int32_t *sweeps,
                                void *stream);

/* vmap(UniformRandomGenerator(m, N).__call__)(keys)
 *   routing_board_generation/rl_training/online_generators/uniform_generator.py:70-109
 * train_grid, position (= start), step_count (= 0) and key_out are nullable. */
int rbg_uniform_generate(const uint32_t *keys, int B, int m, int N, int32_t *train_grid,
                         int32_t *start, int32_t *target, int32_t *position,
                         int32_t *step_count, uint32_t *key_out, void *stream);

/* vmap(BoardDatasetGeneratorJAX.__call__)(keys): sample one of `n_boards` pre-generated boards
 *   routing_board_generation/rl_training/offline_generation/dataset_generator_jax.py:112-141
 * (the dataset itself, :57-110, is rbg_split + rbg_prw_generate(split_first=0) over
 * split(PRNGKey(0), n_boards)).  heads/targets int32 [n_boards][N][2] (agent-major);
 * which (nullable) int32 [B] = the sampled board index ([3P] jax.random.randint semantics). */
int rbg_dataset_sample(const uint32_t *keys, int B, int m, int N, int n_boards,
                       const int32_t *heads, const int32_t *targets, const rbg_state *state,
                       int32_t *which, void *stream);

/* ---- Connector environment ([3P] jumanji/environments/routing/connector/env.py) ----------- */

/* vmap(Connector(generator=ParallelRandomWalkGenerator(m, N)).reset)(keys)
 *   call site: routing_board_generation/rl_training/setup_train.py:400 (pmap(env.reset)). */
int rbg_connector_reset(const uint32_t *keys, int B, int m, int N, const rbg_state *state,
                        const rbg_timestep *timestep, void *stream);

/* Timestep for an existing state: vmap of _get_action_mask / _obs_from_grid / _get_extras +
 * jumanji.types.restart, i.e. demos/board_generator_demo.py:80-96 (board_to_env) and
 * package_evaluation/load_and_test_agents.ipynb:385-394. */
int rbg_connector_observe(int B, int m, int N, const rbg_state *state,
                          const rbg_timestep *timestep, void *stream);

/* VmapAutoResetWrapper(Connector).step(state, action): state_in -> state_out (may alias).
 *   call site: agent_training/training_script.py:70-79 via [3P] jumanji.training agents.
 * action int32 [B][N] in 0..4.  With params->auto_reset, envs whose timestep is LAST are
 * regenerated from split(state.key)[0] inside the same launch and only their observation is
 * replaced ([3P] jumanji/wrappers.py::VmapAutoResetWrapper._auto_reset). */
int rbg_connector_step(int B, int m, int N, const rbg_state *state_in, const rbg_state *state_out,
                       const int32_t *action, const rbg_env_params *params,
                       const rbg_timestep *timestep, void *stream);

/* ---- auto-reset without the walk on the step's critical path --------------------------------
 * VmapAutoResetWrapper restarts a finished episode from split(state.key)[0], and state.key does
 * not change during an episode: the next episode's State is known as soon as the current one
 * starts.  A stash holds that State per env, produced by rbg_stash_refill in ONE launch for all
 * the envs that restarted recently (a throughput-shaped batch of walks) instead of by every step
 * launch for its own few finished envs (a launch that lasts as long as its longest walk); the
 * step then copies 1.3 KB instead of walking.
 *   (tick, ready_delay: int32, 0 <= tick, 0 <= ready_delay, tick + ready_delay <= INT32_MAX, else
 *   RBG_E_PARAM.  A session that outlives 2^31 launches rebases: between two launches, with no
 *   refill in flight, subtract one offset from its tick and from every ready_at (clamped at 0) --
 *   routing_board_generation_amd.session.EpochStash does at tick 2^30.)
 *   depth       slots per env, 1 .. 12; slot s of env b is row b * depth + s of every array below
 *               (ABI 9; the slots of an env are adjacent so that an env's probes touch one or two
 *               cache lines per array; ABI 7-8 had row s * B + b.  Only the env side of the
 *               library maps (env, slot) to rows: refills work on rows)
 *   next        a full rbg_state of depth * B rows: next[r] = reset(split(tag[r])[0])
 *   tag         uint32 [depth*B][2]: the State.key next[r] was generated for (zero-initialised)
 *   ready_at    int32 [depth*B]: first launch tick that may consume the row (zero-initialised)
 *   refill_key  uint32 [depth*B][2]: the State.key whose successor the row should hold next
 *   hits        int32 [B], nullable: resets served from the stash (diagnostics; caller zeroes)
 *   misses      int32 [B], nullable: resets of a stash-backed launch that had to run the generator
 *               inline (caller zeroes; a host that sees them grow refills more often)
 *   inflight    int32 [depth*B], nullable (zero-initialised): non-zero while a listed refill is
 *               generating the row (the value is the claim_ready of the list that claimed it).
 *               Needed when refills overlap each other (rolling refills on side streams, below):
 *               rbg_stash_list skips such rows, so that nothing is generated twice and two
 *               producers never write one row at the same time
 *   far_key     uint32 [B][2], nullable: the farthest key env b holds a request for.  With it a
 *               restart that took its State from the stash makes its one new request --
 *               f(far) into the row it consumed -- without re-deriving the chain (deep stashes:
 *               up to 14 dependent Threefry evaluations otherwise, on the wave the launch waits for)
 *   scratch     int32 [depth*B + 1], nullable: work area of rbg_stash_refill_gen.  With it the
 *               refill first lists the rows that need a generator run and then packs them densely
 *               into wavefronts (SeedExtension runs four boards per wavefront: requests scattered
 *               over the rows would otherwise leave most of their lanes redoing a neighbour's board)
 * Protocol (host-driven, kernel boundaries only -- no in-kernel cross-stream flags):
 *   every reset/step launch has a tick t = 0, 1, 2, ...  A launch that (re)starts env b with
 *   State.key k writes requests (refill_key, ready_at = t + ready_delay), where the caller
 *   guarantees that a refill launched after launch t has completed before launch
 *   t + ready_delay starts.  With the refill on the launches' own stream every K launches that
 *   is the next multiple of K (stream order, no events; what
 *   routing_board_generation_amd.ops.EpochStash does by default); with the refill on a side
 *   stream it takes a stream event.  A row is consumed only if tick >= ready_at[r] and
 *   tag[r] == state.key[b]; otherwise the walk runs inline, so results never depend on timing.
 *   depth 1: the one row of env b is requested for k (the successor of the episode that starts).
 *   depth D > 1: the env wants the successors of its next D episodes, i.e. the rows requested for
 *   k, f(k), f(f(k)), ... with f(k) = split(split(k)[0])[0] (the wrapper's split, then the
 *   generator's: the State.key of the episode after the one with State.key k -- known without
 *   running any generator).  Rows that already carry one of these requests keep it; the others
 *   (just consumed, or never used) take the missing ones.  Every entry is then asked for D - 1
 *   whole episodes before it is needed, which lets a launch of many steps
 *   (rbg_connector_rollout, D = 2) serve all its resets from the stash, also those of an env
 *   that finishes two episodes within it, and lets a generator whose refill launch is long
 *   whatever its row count (SequentialRandomWalk, SeedExtension: as long as their slowest board)
 *   refill rarely.  The number of generator runs is the same as with depth 1; they only happen
 *   earlier. */
typedef struct rbg_stash {
  rbg_state next;
  uint32_t *tag;
  int32_t *ready_at;
  uint32_t *refill_key;
  int32_t *hits;
  int32_t *misses;
  int32_t *scratch;
  int32_t *inflight;
  uint32_t *far_key;
  int32_t depth;
} rbg_stash;

/* rbg_connector_reset / rbg_connector_step with a stash (NULL stash = the plain calls). */
int rbg_connector_reset_prefetch(const uint32_t *keys, int B, int m, int N, const rbg_state *state,
                                 const rbg_timestep *timestep, const rbg_stash *stash, int tick,
                                 int ready_delay, void *stream);
int rbg_connector_step_prefetch(int B, int m, int N, const rbg_state *state_in,
                                const rbg_state *state_out, const int32_t *action,
                                const rbg_env_params *params, const rbg_timestep *timestep,
                                const rbg_stash *stash, int tick, int ready_delay, void *stream);
/* Generate next[r] for every row with tag[r] != refill_key[r] (one sub-wave per row; the others
 * retire at once). */
int rbg_stash_refill(int B, int m, int N, const rbg_stash *stash, void *stream);

/* Masked-uniform random policy over the legal set (defined by this project; stands in for
 * [3P] jumanji.training.agents.random.RandomAgent in the rollout benchmark, BASELINE config 3):
 * at global step t, agent i of env b picks legal action number
 *   floor(threefry2x32(keys[b], counter (t, i)).y0 * #legal / 2^32)
 * (legal actions in ascending order; NOOP is always legal). */
int rbg_random_actions(const uint32_t *keys, int B, int N, uint32_t t, const uint8_t *action_mask,
                       int32_t *action, void *stream);

/* ---------------------------------------------------------------------------------------------
 * The walk taken apart: the methods of ParallelRandomWalkBoard that the reference's unit tests
 * call one by one (JAXI/board_generation/test_parallel_random_walk_board.py), on grids in the
 * reference's encoding (PATH 1 + 3i, POSITION 2 + 3i, TARGET 3 + 3i) and with the reference's
 * treatment of arbitrary arguments (flat cells go through python floor division, -1 = no cell,
 * out-of-range gathers wrap once and clamp).  generate_board does not go through these: it runs
 * the fused walk.  Agents are (target, position) [B][N][2]; Agent.id = arange(N).
 * ------------------------------------------------------------------------------------------- */

/* _initialise_agents(key, grid) (parallel_random_walk.py:147-186): start = the first N entries of
 * jax.random.permutation(key, m * m); grid_out = grid_in (NULL = empty) with the heads 2 + 3i
 * max-joined over the agents' private copies.  start doubles as Agent.position; target = -1. */
int rbg_prw_initialise_agents(const uint32_t *keys, int B, int m, int N, const int32_t *grid_in,
                              int32_t *grid_out, int32_t *start, void *stream);

/* _step((key, grid, agents)) (:92-145): one iteration of the walk.  key_out = split(key)[1];
 * action_out (nullable) = what _select_action drew for every agent.  target NULL = all -1. */
int rbg_prw_step(const uint32_t *keys, int B, int m, int N, const int32_t *grid,
                 const int32_t *target, const int32_t *position, uint32_t *key_out,
                 int32_t *grid_out, int32_t *position_out, int32_t *action_out, void *stream);

/* _select_action(key, grid, agent) (:205-230), one key per agent: agent_keys [B][N][2]. */
int rbg_prw_select_action(const uint32_t *agent_keys, int B, int m, int N, const int32_t *grid,
                          const int32_t *position, int32_t *action, void *stream);

/* Q flat cells per board, cells [B][Q] (any int32).  Outputs, all nullable:
 *   adjacent [B][Q][4]       _adjacent_cells(cell) (:262-291), [up, down, left, right], -1 = none
 *   available [B][Q][4]      _available_cells(grid, cell) (:293-319)
 *   is_free [B][Q]           _is_cell_free(grid, cell)[1] (:321-338)
 *   not_doubling_back [B][Q] _is_cell_doubling_back((grid, wire), cell)[1] (:340-374) with
 *                            wire = wire_id[b][q], or (grid[cell] - 1) // 3 when wire_id is NULL
 * grid may be NULL when only `adjacent` is asked for. */
int rbg_prw_cell_queries(const int32_t *grid, int B, int m, const int32_t *cells, int Q,
                         const int32_t *wire_id, int32_t *adjacent, int32_t *available,
                         uint8_t *is_free, uint8_t *not_doubling_back, void *stream);

/* _step_agent(agent, grid, action) (:376-429) for every agent on its OWN copy of the grid:
 * new_position [B][N][2]; new_grids [B][N][m][m] (nullable); moved [B][N] (nullable) =
 * _is_valid_position(grid, agent, move_position(position, action)) & (action != NOOP). */
int rbg_prw_step_agent(const int32_t *grid, int B, int m, int N, const int32_t *target,
                       const int32_t *position, const int32_t *action, int32_t *new_position,
                       int32_t *new_grids, uint8_t *moved, void *stream);

/* Per-env counters of a rollout (all members nullable, each int32 [B], overwritten per call).
 * Returns are integer counts so that they do not depend on a summation order:
 *   episode return summed over agents and steps
 *     = connected_reward * connections_made + timestep_reward * unconnected_agent_steps. */
typedef struct rbg_rollout_stats {
  int32_t *episodes;                /* timesteps with step_type LAST */
  int32_t *connections_made;        /* agent connection events */
  int32_t *unconnected_agent_steps; /* (agent, step) pairs that began the step unconnected */
  int32_t *final_connections;       /* sum over ended episodes of num_connections at LAST */
  int32_t *final_path_length;       /* sum over ended episodes of total_path_length at LAST */
} rbg_rollout_stats;

/* n_steps of VmapAutoResetWrapper(Connector).step under the rbg_random_actions policy (global
 * steps t0 .. t0 + n_steps - 1) in ONE launch: the rollout loop of
 * routing_board_generation/benchmarking/benchmarks/benchmark_on_random_agent.py:59-102 /
 * agent_training/training_script.py:70-79 with [3P] jumanji's RandomAgent.
 * Equivalent to n_steps x { rbg_random_actions; rbg_connector_step } with the same params, but
 * the env lives in LDS between steps: no per-step state or observation traffic.
 * state_out may alias state_in.  last_timestep (nullable, members nullable) receives the
 * timestep of the final step exactly as rbg_connector_step would have returned it. */
int rbg_connector_rollout(int B, int m, int N, const rbg_state *state_in,
                          const rbg_state *state_out, const uint32_t *policy_keys, uint32_t t0,
                          int n_steps, const rbg_env_params *params,
                          const rbg_rollout_stats *stats, const rbg_timestep *last_timestep,
                          void *stream);

/* rbg_connector_rollout with a stash: every reset of an env inside the launch takes its State
 * from the stash when a matching row is ready -- with depth 2 that includes the second reset of
 * an env that finishes two episodes within the launch; a reset without a ready row walks inline
 * (with n_steps = 1 and last_timestep set this is a fused "random policy + step" launch, one
 * kernel per env step). */
int rbg_connector_rollout_prefetch(int B, int m, int N, const rbg_state *state_in,
                                   const rbg_state *state_out, const uint32_t *policy_keys,
                                   uint32_t t0, int n_steps, const rbg_env_params *params,
                                   const rbg_rollout_stats *stats,
                                   const rbg_timestep *last_timestep, const rbg_stash *stash,
                                   int tick, int ready_delay, void *stream);

/* The same rollout that also RECORDS the trajectory an acting loop collects
 * (benchmarking/benchmarks/benchmark_on_random_agent.py:71-86, the lax.scan over env.step inside
 * run_epoch): `trajectory` is an rbg_timestep whose arrays have a leading n_steps axis -- member
 * x of step s and env b at row s * B + b -- and receives, for every step, exactly what
 * rbg_connector_step would have returned for it (observation of the next episode after an
 * auto-reset, reward / discount / step_type / extras of the step that ended); trajectory_action
 * (int32 [n_steps][B][N], nullable) receives the policy's actions.  All members nullable.  With
 * the observation recorded the launch is an HBM stream (11.7 KB per env-step at 16x16/10) whose
 * launch cost is paid once per n_steps.  stash may be NULL. */
int rbg_connector_rollout_record(int B, int m, int N, const rbg_state *state_in,
                                 const rbg_state *state_out, const uint32_t *policy_keys,
                                 uint32_t t0, int n_steps, const rbg_env_params *params,
                                 const rbg_rollout_stats *stats,
                                 const rbg_timestep *last_timestep,
                                 const rbg_timestep *trajectory, int32_t *trajectory_action,
                                 const rbg_stash *stash, int tick, int ready_delay,
                                 void *stream);

/* ---- the coupled env hot path for EVERY generator -------------------------------------------
 * The reference couples any `Generator` to `Connector` under VmapAutoResetWrapper
 * (routing_board_generation/rl_training/setup_train.py:112-161: generation_type offline_seed_extension,
 * offline_parallel_rw, online_uniform, online_seq_rw, online_random_seed, online_parallel_rw).
 * rbg_generator names the generator whose `__call__(key)` (re)starts episodes inside the env
 * launches below; NULL = ParallelRandomWalkGenerator (the calls above).  All of them start with
 * `key, _ = split(key)` and store that child as State.key, which is what the stash protocol needs.
 *   RBG_GEN_PARALLEL_RW     rl_training/online_generators/parallel_random_walk_generator.py:46-77
 *   RBG_GEN_UNIFORM         rl_training/online_generators/uniform_generator.py:70-109
 *   RBG_GEN_SEQUENTIAL_RW   rl_training/online_generators/sequential_random_walk_generator.py:34-66
 *   RBG_GEN_SEED_EXTENSION  rl_training/online_generators/random_seed_generator.py:28-57 with
 *                           randomness / two_sided / extension_iterations / max_extension_steps as in
 *                           rbg_seed_extension_generate (the generator's defaults: 0, 1, 1, INT32_MAX)
 *   RBG_GEN_PARALLEL_RW_PHILOX  nothing in the reference: ParallelRandomWalkPhiloxGenerator, whose
 *                           State.key is words 0, 1 of philox(key, (0, 0, 0, 3)) instead of a split
 *   RBG_GEN_DATASET         rl_training/offline_generation/dataset_generator_jax.py:112-141:
 *                           sample one of n_boards stored boards, heads / targets int32
 *                           [n_boards][N][2] (agent-major, device memory that outlives the launches) */
#define RBG_GEN_PARALLEL_RW 0
#define RBG_GEN_UNIFORM 1
#define RBG_GEN_SEQUENTIAL_RW 2
#define RBG_GEN_SEED_EXTENSION 3
#define RBG_GEN_DATASET 4
#define RBG_GEN_PARALLEL_RW_PHILOX 5 /* the opt-in stream of rbg_prw_generate_philox (not the reference's) */

typedef struct rbg_generator {
  int32_t kind;                 /* RBG_GEN_* */
  float randomness;             /* SEED_EXTENSION */
  int32_t two_sided;            /* SEED_EXTENSION */
  int32_t extension_iterations; /* SEED_EXTENSION, >= 0 */
  int32_t max_extension_steps;  /* SEED_EXTENSION */
  int32_t n_boards;             /* DATASET */
  const int32_t *heads;         /* DATASET */
  const int32_t *targets;       /* DATASET */
} rbg_generator;

/* vmap(Connector(generator).reset)(keys); stash may be NULL (then tick / ready_delay are unused). */
int rbg_connector_reset_gen(const rbg_generator *generator, const uint32_t *keys, int B, int m,
                            int N, const rbg_state *state, const rbg_timestep *timestep,
                            const rbg_stash *stash, int tick, int ready_delay, void *stream);

/* VmapAutoResetWrapper(Connector(generator)).step: rbg_connector_step_prefetch for any generator.
 * A finished episode takes its successor from the stash when a matching row is ready; otherwise
 * the generator runs inline in that env's wavefront (same result, a longer launch), so the
 * reference's "regenerate every env on every step" (vmap(lax.cond)) never happens.
 * policy_keys (nullable): non-NULL = the action comes from the rbg_random_actions policy of global
 * step policy_t inside the launch (action may then be NULL). */
int rbg_connector_step_gen(const rbg_generator *generator, int B, int m, int N,
                           const rbg_state *state_in, const rbg_state *state_out,
                           const int32_t *action, const uint32_t *policy_keys, uint32_t policy_t,
                           const rbg_env_params *params, const rbg_timestep *timestep,
                           const rbg_stash *stash, int tick, int ready_delay, void *stream);

/* rbg_connector_step_gen with its arguments in ONE struct, for hosts whose FFI pays per argument
 * and per temporary (ctypes: ~0.25 us per argument, ~1.5 us per struct built; at B = 4096 the step
 * launch itself takes 12 us): a caller keeps one rbg_step_call per env session, rewrites the
 * pointers that change from step to step -- state_out and timestep are adjacent, 17 consecutive
 * pointers -- and passes its address.  Same checks, same launch. */
typedef struct rbg_step_call {
  const rbg_generator *generator; /* NULL = ParallelRandomWalk */
  int32_t B, m, N;
  int32_t tick, ready_delay;
  uint32_t policy_t;
  rbg_state state_in;
  rbg_state state_out;
  rbg_timestep timestep;
  const int32_t *action;
  const uint32_t *policy_keys;
  rbg_env_params params;
  const rbg_stash *stash;
  void *stream;
} rbg_step_call;
int rbg_connector_step_call(const rbg_step_call *call);

/* The two halves of a refill, for hosts that keep SEVERAL refills in flight beside the stepping
 * stream ("rolling refills": a SeedExtension / SequentialRandomWalk refill lasts 0.4-1.8 ms
 * whatever its row count -- as long as its slowest board -- but leaves most of the chip idle, and
 * neither it nor the step launches are slowed much by the other):
 *   rbg_stash_list            lists the rows with a pending request (tag != refill_key) that no
 *                             earlier refill is still generating into stash->scratch ([0] = count)
 *                             and claims them in stash->inflight.  Calls must be stream-ordered
 *                             among themselves (one "listing" order); each in-flight refill
 *                             needs a scratch area of its own (pass a copy of the struct).
 *   rbg_stash_refill_listed   generates exactly the listed rows and releases them (tag, then the
 *                             claim, release-ordered; the list reads the claim first, acquire).
 * Ordering contract (host, events): a list call follows the launches whose requests it is to
 * see; its producer follows it; the launch that may consume a row (ready_at) follows the
 * producer.  Requests written while a refill runs are simply seen by a later list.
 * ready_at stays truthful under re-assignment: claim_ready = the ready_at carried by the requests
 * this list serves (tick + ready_delay of the launches since the previous list, >= 1), skip_ready
 * = the ready_at of the requests the NEXT list will serve (>= claim_ready).  A row that is skipped
 * because an earlier refill is still generating it, although its request is newer than that
 * refill's claim (ready_at > the claim's value: the env re-assigned the row while its producer
 * ran), can only be generated by the next list's producer, so its ready_at is raised to
 * skip_ready.  Without in-flight claims (stash->inflight NULL) both values are ignored. */
int rbg_stash_list(int B, int m, int N, const rbg_stash *stash, int claim_ready, int skip_ready,
                   void *stream);
int rbg_stash_refill_listed(const rbg_generator *generator, int B, int m, int N,
                            const rbg_stash *stash, void *stream);

/* One epoch boundary of the rolling form in ONE call, for hosts whose FFI and event API cost
 * microseconds per operation (Python: five event operations, two library calls and the miss
 * counter's snapshot came to ~80 us per refill).  In this order:
 *   record `launched` on step_stream; step_stream waits for `await_done` (nullable: the refill
 *   whose rows the NEXT launch may be the first to consume); gen_stream waits for `launched` and
 *   for `after_listed` (nullable: the previous refill's list); on gen_stream: rbg_stash_list
 *   (claim_ready, skip_ready), the miss snapshot (optional, below), record `listed`,
 *   rbg_stash_refill_listed, record `done`.
 * All events are hipEvent_t owned by the caller.  Miss snapshot (miss_word non-NULL): the list
 * launch also sums stash->misses through *miss_acc (a device work word, zero before the first
 * call and again after every call) and writes (sum << 20 | stamp & 0xFFFFF) to *miss_word
 * (device) and to *miss_host (pinned host memory: written by the launch itself when the device
 * can address it -- hipHostMalloc / hipHostRegister memory -- else by an 8-byte copy on
 * gen_stream) -- never a synchronisation; the host reads whatever has arrived
 * (schedule.MissWatch). */
typedef struct rbg_stash_roll_call {
  const rbg_generator *generator; /* NULL = ParallelRandomWalk */
  int32_t B, m, N;
  int32_t claim_ready, skip_ready;
  uint32_t stamp;
  const rbg_stash *stash; /* with the scratch area of THIS refill; inflight required */
  void *step_stream, *gen_stream;
  void *launched, *await_done, *after_listed, *listed, *done; /* hipEvent_t */
  int64_t *miss_acc, *miss_word; /* device, nullable together */
  int64_t *miss_host;            /* pinned host */
} rbg_stash_roll_call;
int rbg_stash_roll(const rbg_stash_roll_call *call);

/* rbg_stash_refill with the generator's own producer: next[r] = generator(split(refill_key[r])[0])
 * for every row with tag[r] != refill_key[r]. */
int rbg_stash_refill_gen(const rbg_generator *generator, int B, int m, int N,
                         const rbg_stash *stash, void *stream);

/* rbg_connector_rollout_record for any generator (the random-agent benchmark runs on whatever
 * generator the config names: benchmarking/benchmarks/benchmark_on_random_agent.py:59-102). */
int rbg_connector_rollout_gen(const rbg_generator *generator, int B, int m, int N,
                              const rbg_state *state_in, const rbg_state *state_out,
                              const uint32_t *policy_keys, uint32_t t0, int n_steps,
                              const rbg_env_params *params, const rbg_rollout_stats *stats,
                              const rbg_timestep *last_timestep, const rbg_timestep *trajectory,
                              int32_t *trajectory_action, const rbg_stash *stash, int tick,
                              int ready_delay, void *stream);


#ifdef __cplusplus
}
#endif
#endif /* RBG_PRW_H_ */
