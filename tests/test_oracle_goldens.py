"""Pins the CPU oracle (oracle/prw_oracle.c) against every golden the reference holds for the path.

Each test names the reference test / notebook cell it mirrors
(JAXI = routing_board_generation/board_generation_methods/jax_implementation/board_generation).
"""
import numpy as np
import pytest

WINDOW = [0, 1, 2, 7, 8, 9]          # numpy summarised print: 3 leading + 3 trailing items
WINDOW_BOARDS = [0, 1, 2, 97, 98, 99]


def _agents(d):
    return {k: np.asarray(v, np.int32) for k, v in d.items()}


# --------------------------------------------------------------------------- PRNG (G1, A.2)
def test_prngkey_and_split_known_answer(oracle):
    # State.key printed in profiling_generators.ipynb cells 4/7/10/13 == split(PRNGKey(0))[0]
    ks = oracle.split(oracle.prng_key(0))
    assert ks.tolist() == [[4146024105, 967050713], [2718843009, 1272950319]]
    assert oracle.prng_key(7).tolist() == [0, 7]
    assert oracle.prng_key((5 << 32) | 9).tolist() == [5, 9]


def test_batch_keys_match_notebook(oracle, goldens):
    # keys = jax.random.split(PRNGKey(0), 100) pass through generator: State.key = split(k)[0]
    b = goldens["notebook"]["batch100_10x10_5_split_PRNGKey0"]["parallel_random_walk"]
    keys = oracle.split(oracle.prng_key(0), 100)
    got = np.stack([oracle.split(k)[0] for k in keys])
    assert (got == np.asarray(b["key"], np.uint32)).all()


def test_random_bits_odd_padding(oracle):
    # threefry_2x32 pads an odd count with one 0 word: the first ceil(n/2) words use counters
    # (j, half+j); the last block of an odd request uses (half-1, 0).
    key = oracle.split(oracle.prng_key(3))[1]
    bits5 = oracle.random_bits(key, 5)
    for j in range(3):
        c1 = 3 + j if 3 + j < 5 else 0
        y0, y1 = oracle.threefry2x32(key, j, c1)
        assert bits5[j] == y0
        if 3 + j < 5:
            assert bits5[3 + j] == y1
    assert oracle.random_bits(key, 1)[0] == oracle.threefry2x32(key, 0, 0)[0]


def test_uniform_range(oracle):
    for s in range(50):
        u = oracle.uniform(oracle.prng_key(s))
        assert 0.0 <= float(u) < 1.0


# --------------------------------------------------------------------------- reference unit tests
def test_initialise_agents(oracle, goldens):  # JAXI/test_...py::test_initialise_agents
    tf = goldens["test_file"]
    grid, agents = oracle.initialise_agents(oracle.prng_key(0), 5, 3)
    assert (grid == np.asarray(tf["valid_starting_grid"])).all()
    exp = _agents(tf["agents_starting"])
    for k in exp:
        assert (agents[k] == exp[k]).all(), k


def test_step(oracle, goldens):  # ::test_step
    tf = goldens["test_file"]
    key = oracle.prng_key(0)
    new_key, grid, agents = oracle.prw_step(key, tf["valid_starting_grid"], tf["agents_starting"])
    assert (new_key == oracle.split(key)[1]).all()
    assert (grid == np.asarray(tf["valid_starting_grid_after_1_step"])).all()
    exp = _agents(tf["agents_starting_move_1_step_up"])
    for k in exp:
        assert (agents[k] == exp[k]).all(), k


def test_generate_board(oracle, goldens):  # ::test_generate_board
    tf = goldens["test_file"]
    heads, targets, grid, iters = oracle.generate_board(oracle.prng_key(0), 5, 3)
    exp = _agents(tf["agents_reshaped_for_generator"])
    assert (grid == np.asarray(tf["valid_end_grid2"])).all()
    assert (heads.T == exp["start"]).all()
    assert (targets.T == exp["position"]).all()
    assert iters == 6


def test_generate_board_for_various_keys(oracle):  # ::test_generate_board_for_various_keys
    boards = [oracle.generate_board(oracle.prng_key(i), 5, 3)[2] for i in range(10)]
    for i in range(10):
        for j in range(i):
            assert not (boards[i] == boards[j]).all()


@pytest.mark.parametrize("grid_name,agents_name,expected",
                         [("valid_end_grid", "agents_finished", False),
                          ("valid_starting_grid", "agents_starting", True)])
def test_continue_stepping(oracle, goldens, grid_name, agents_name, expected):
    tf = goldens["test_file"]
    assert oracle.continue_stepping(tf[grid_name], tf[agents_name]) == expected


@pytest.mark.parametrize("cell,expected", [(0, [-1, 5, -1, 1]), (6, [1, 11, 5, 7])])
def test_adjacent_cells(oracle, cell, expected):  # ::test_adjacent_cells
    assert oracle.adjacent_cells(cell, 5, 5).tolist() == expected


@pytest.mark.parametrize("grid_name,cell,expected",
                         [("valid_end_grid", 8, [-1, -1, -1, -1]),
                          ("grid_to_test_available_cells", 8, [-1, 13, -1, -1])])
def test_available_cells(oracle, goldens, grid_name, cell, expected):  # ::test_available_cells
    assert oracle.available_cells(goldens["test_file"][grid_name], cell).tolist() == expected


@pytest.mark.parametrize("cell,expected", [(21, True), (22, False), (32, False)])
def test_is_cell_free(oracle, goldens, cell, expected):  # ::test_is_cell_free
    assert oracle.is_cell_free(goldens["test_file"]["valid_starting_grid"], cell) == expected


@pytest.mark.parametrize("p1,p2,expected", [(0, 1, 2), (1, 0, 4), (0, 5, 3), (5, 0, 1)])
def test_action_from_position(oracle, p1, p2, expected):  # ::test_action_from_position
    assert oracle.action_from_positions(p1, p2, 5) == expected


def test_action_from_position_sentinel(oracle):
    # blocked agent -> choice returns -1 -> (-1, cols-1): NOOP except from (0, cols-1)
    assert oracle.action_from_positions(7, -1, 5) == 0
    assert oracle.action_from_positions(4, -1, 5) == 1  # UP, rejected later by is_valid_position


def test_step_agent(oracle, goldens):  # ::test_step_agent, actions [1,1,1]
    tf = goldens["test_file"]
    exp_agents = _agents(tf["agents_starting_move_1_step_up"])
    for i in range(3):
        pos, grid = oracle.step_agent(i, tf["agents_starting"], tf["valid_starting_grid"], 1)
        assert (grid == np.asarray(tf["grids_after_1_agent_step"][i])).all()
        assert pos == exp_agents["position"][i].tolist()


# --------------------------------------------------------------------------- notebook goldens
@pytest.mark.parametrize("name", ["parallel_random_walk", "uniform"])
def test_single_board_generators(oracle, goldens, name):  # profiling nb cells 4 and 13 (G5, G6)
    s = goldens["notebook"]["single_10x10_5_PRNGKey0"][name]
    st, _, iters = oracle.generate_batch(oracle.prng_key(0)[None], 10, 5, generator=name)
    assert (st["grid"][0] == np.asarray(s["grid"])).all()
    assert (st["start"][0] == np.asarray(s["start"])).all()
    assert (st["target"][0] == np.asarray(s["target"])).all()
    assert (st["position"][0] == np.asarray(s["position"])).all()
    assert (st["key"][0] == np.asarray(s["key"], np.uint32)).all()
    assert st["step_count"][0] == 0


@pytest.mark.parametrize("name", ["parallel_random_walk", "uniform"])
def test_batch100_generators(oracle, goldens, name):  # profiling nb cells 18 and 26
    """vmap(generator)(split(PRNGKey(0), 100)): start/target/position/key are printed in full
    (100x5x2 = 1000 elements, numpy's summarisation threshold), grids as 6x6x6 corner windows."""
    b = goldens["notebook"]["batch100_10x10_5_split_PRNGKey0"][name]
    keys = oracle.split(oracle.prng_key(0), 100)
    st, _, _ = oracle.generate_batch(keys, 10, 5, generator=name, threads=2)
    assert (st["start"] == np.asarray(b["start"])).all()
    assert (st["target"] == np.asarray(b["target"])).all()
    assert (st["position"] == np.asarray(b["position"])).all()
    assert (st["key"] == np.asarray(b["key"], np.uint32)).all()
    win = st["grid"][WINDOW_BOARDS][:, WINDOW][:, :, WINDOW]
    assert (win == np.asarray(b["grid_window"])).all()


# --------------------------------------------------------------------------- collision KAT
def _kat_state():
    grid = np.zeros((5, 5), np.int32)
    grid[2, 0], grid[2, 1] = 1, 2       # agent 0: path + head
    grid[2, 3] = 5                       # agent 1 head
    grid[1, 2] = 8                       # agent 2 head
    agents = {"id": [0, 1, 2], "start": [[2, 0], [2, 3], [1, 2]],
              "target": [[-1, -1]] * 3, "position": [[2, 1], [2, 3], [1, 2]]}
    return grid, agents


def test_collision_kat_two_way(oracle):
    """Hand-derived from JAXI/parallel_random_walk.py:104-107,128 (no reference golden has a
    contested cell in a *single* pinned step; the 100-board batch golden covers it end-to-end)."""
    grid, agents = _kat_state()
    g, a = oracle.step_agents_with_actions(grid, agents, [2, 4, 0])  # RIGHT, LEFT, NOOP
    assert g[2].tolist() == [1, 2, 5, 4, 0]
    assert a["position"].tolist() == [[2, 1], [2, 2], [1, 2]]


def test_collision_kat_three_way(oracle):
    grid, agents = _kat_state()
    g, a = oracle.step_agents_with_actions(grid, agents, [2, 4, 3])  # RIGHT, LEFT, DOWN
    assert g[1].tolist() == [0, 0, 7, 0, 0]
    assert g[2].tolist() == [1, 2, 8, 5, 0]
    assert a["position"].tolist() == [[2, 1], [2, 3], [2, 2]]


# --------------------------------------------------------------------------- structural invariants
def _same_wire_neighbours(grid, r, c, wire):
    m = grid.shape[0]
    cnt = 0
    for dr, dc in ((-1, 0), (1, 0), (0, -1), (0, 1)):
        rr, cc = r + dr, c + dc
        if 0 <= rr < m and 0 <= cc < m and grid[rr, cc] != 0 and (grid[rr, cc] - 1) // 3 == wire:
            cnt += 1
    return cnt


def check_solved_board(solved, start, target):
    """verify_wire_validity (NPI/utils/post_processor_utils_numpy.py:90-121): head/target cells
    have exactly one same-wire neighbour, path cells exactly two; a never-moved agent is a lone
    TARGET cell."""
    m = solved.shape[0]
    N = start.shape[0]
    for i in range(N):
        sr, sc = start[i]
        tr, tc = target[i]
        assert solved[tr, tc] == 3 + 3 * i
        if (sr, sc) == (tr, tc):
            assert _same_wire_neighbours(solved, sr, sc, i) == 0
        else:
            assert solved[sr, sc] == 2 + 3 * i
    for r in range(m):
        for c in range(m):
            v = solved[r, c]
            if v == 0:
                continue
            wire, kind = (v - 1) // 3, (v - 1) % 3
            n = _same_wire_neighbours(solved, r, c, wire)
            if kind == 0:
                assert n == 2, (r, c, v)
            elif not (start[wire] == target[wire]).all():
                assert n == 1, (r, c, v)


def test_validity_invariants_c1_and_c2_shapes(oracle):
    for (m, N, B) in ((8, 5, 64), (16, 10, 64)):
        keys = oracle.split(oracle.prng_key(0), B)
        st, solved, iters = oracle.generate_batch(keys, m, N, threads=2)
        assert (iters > 0).all()
        for b in range(B):
            check_solved_board(solved[b], st["start"][b], st["target"][b])
            # training grid holds heads+targets only
            assert (st["grid"][b] != 0).sum() == 2 * N - (st["start"][b] == st["target"][b]).all(1).sum()


# --------------------------------------------------------------------------- SequentialRandomWalk
def test_sequential_random_walk_generator_golden(oracle, goldens):  # profiling nb cell 7 (G7)
    s = goldens["notebook"]["single_10x10_5_PRNGKey0"]["sequential_random_walk"]
    st, solved, attempts = oracle.seq_rw_generator_batch(oracle.prng_key(0)[None], 10, 5)
    assert (st["grid"][0] == np.asarray(s["grid"])).all()
    assert (st["start"][0] == np.asarray(s["start"])).all()
    assert (st["target"][0] == np.asarray(s["target"])).all()
    assert (st["position"][0] == np.asarray(s["position"])).all()
    assert (st["key"][0] == np.asarray(s["key"], np.uint32)).all()
    assert attempts[0] == 1
    check_solved_board(solved[0], st["start"][0], st["target"][0])


def test_sequential_random_walk_retries_and_failure(oracle):
    """Dense boards: a wire that cannot start or move fails the attempt; the board is retried with
    max_length - 1 and the SAME key; after 2m failures the board is all zero."""
    st, solved, attempts = oracle.seq_rw_generator_batch(oracle.split(oracle.prng_key(1), 128), 4, 7,
                                                         threads=2)
    assert (attempts == 0).any() and (attempts > 1).any()
    failed = attempts == 0
    assert (solved[failed] == 0).all() and (st["start"][failed] == 0).all()
    assert (st["grid"][failed][:, 0, 0] == 21).all()
    for b in np.flatnonzero(~failed)[:20]:
        check_solved_board(solved[b], st["start"][b], st["target"][b])


# --------------------------------------------------------------------------- SeedExtension (G8)
def test_seed_extension_batch100_golden(oracle, goldens):  # profiling nb cell 23
    """vmap(SeedExtensionGenerator(10, 5).__call__) over split(PRNGKey(0), 100): every printed
    value of the reference's executed notebook is reproduced -- all 1000 start and 1000 target
    coordinates, the 100 State keys and the 6x6x6 corner windows of the training grids.  This pins
    _randint, the seeding, extend_wires_jax (flips, key threading, choice loop, stopping rule)
    for the defaults randomness=0, two_sided=True, extension_iterations=1."""
    b = goldens["notebook"]["batch100_10x10_5_split_PRNGKey0"]["seed_extension"]
    keys = oracle.split(oracle.prng_key(0), 100)
    st, solved, sweeps = oracle.seed_extension_batch(keys, 10, 5)
    assert (st["key"] == np.asarray(b["key"], np.uint32)).all()
    assert (st["start"] == np.asarray(b["start"], np.int32)).all()
    assert (st["target"] == np.asarray(b["target"], np.int32)).all()
    assert (st["position"] == np.asarray(b["position"], np.int32)).all()
    win = st["grid"][np.ix_(WINDOW_BOARDS, WINDOW, WINDOW)]
    assert (win == np.asarray(b["grid_window"], np.int32)).all()
    assert (st["step_count"] == 0).all()
    assert sweeps.min() >= 1
    _check_seed_extension_boards(solved, st["start"], st["target"])


def test_seed_extension_single_board_golden_is_stale(oracle, goldens):  # profiling nb cell 10
    """The single-board cell of the notebook was executed with an older reference (it imports the
    class as RandomSeedGenerator).  What today's code still shares with it: State.key and the two
    endpoints that never move (wire 0 head (6,5), wire 3 target (5,5)) -- they equal the SEEDED
    board of today's return_seeded_board.  The rest of that cell is 'parity unpinned, G8 stale'."""
    g = goldens["notebook"]["single_10x10_5_PRNGKey0"]["seed_extension"]
    st, _, _ = oracle.seed_extension_batch(oracle.prng_key(0)[None], 10, 5, extension_iterations=0)
    assert st["key"][0].tolist() == g["key"]
    assert st["start"][0, 0].tolist() == g["start"][0] == [6, 5]
    assert st["target"][0, 3].tolist() == g["target"][3] == [5, 5]


def _check_seed_extension_boards(solved, start, target):
    """Every wire is a simple chain from its POSITION cell to its TARGET cell (what
    NPI/utils/post_processor_utils_numpy.py:90-121 verifies), endpoints where State says."""
    from helpers import same_wire_neighbour_count
    B, m, _ = solved.shape
    N = start.shape[1]
    cnt = same_wire_neighbour_count(solved)
    kind = np.where(solved > 0, (solved - 1) % 3, -1)
    assert (solved >= 0).all() and (solved <= 3 * N).all()
    assert (cnt[kind == 0] == 2).all()
    assert (cnt[kind > 0] == 1).all()
    bb = np.arange(B)[:, None]
    i = np.arange(N)[None, :]
    assert (solved[bb, start[..., 0], start[..., 1]] == 2 + 3 * i).all()
    assert (solved[bb, target[..., 0], target[..., 1]] == 3 + 3 * i).all()
    for v in (2, 3):
        assert ((solved[:, :, :, None] == v + 3 * np.arange(N)).sum((1, 2)) == 1).all()


@pytest.mark.parametrize("m,N,kw", [
    (10, 5, dict(randomness=1.0, two_sided=False)),      # BoardDatasetGeneratorJAX defaults
    (10, 5, dict(randomness=0.85, extension_iterations=3)),
    (16, 10, dict()),
    (8, 16, dict(randomness=0.3)),                       # every candidate cell seeded
    (5, 2, dict(extension_iterations=2)), (7, 4, dict()), (2, 1, dict()), (3, 1, dict()),
    (16, 10, dict(extension_steps=2)),
])
def test_seed_extension_invariants(oracle, m, N, kw):
    """Unpinned parameter space (randomness > 0, one-sided, several extension iterations, odd
    sizes): structural invariants of the restatement + endpoints untouched by optimise_wire."""
    keys = oracle.split(oracle.prng_key(11), 64)
    st, solved, sweeps = oracle.seed_extension_batch(keys, m, N, **kw)
    _check_seed_extension_boards(solved, st["start"], st["target"])
    if kw.get("extension_steps"):
        assert (sweeps <= kw["extension_steps"] * kw.get("extension_iterations", 1)).all()
    if not kw.get("two_sided", True):
        # heads never move: they sit where the seeding put them
        seeded, _, _ = oracle.seed_extension_batch(keys, m, N, extension_iterations=0, **{
            k: v for k, v in kw.items() if k != "extension_iterations"})
        assert (seeded["start"] == st["start"]).all()
    # board-level entry (generate_starts_ends) = generator entry on split(key)[0]
    k0 = np.stack([oracle.split(k)[0] for k in keys])
    st2, solved2, _ = oracle.seed_extension_batch(k0, m, N, split_first=False, **kw)
    assert (solved2 == solved).all() and (st2["start"] == st["start"]).all()


def test_seed_extension_rejects_too_many_agents(oracle):
    with pytest.raises(ValueError):
        oracle.seed_extension_batch(oracle.prng_key(0)[None], 10, 26)  # 25 candidate cells


def test_randint_matches_scalar_form(oracle):
    # the vector form with n = 1 is the scalar jax.random.randint used by the dataset sampler
    for s in range(20):
        k = oracle.split(oracle.prng_key(s))[1]
        for span in (2, 4, 7, 100, 65537):
            assert int(oracle.randint_vec(k, 1, 0, span)[0]) == oracle.randint(k, 0, span)


def test_c1_boards_of_survey_appendix_b1(oracle):
    """SURVEY.md Appendix B.1 (BASELINE configs[0], 8x8 / 5 agents): the board for PRNGKey(0) and
    the B = 1 batch form, from the survey's independent emulation."""
    st, solved, iters = oracle.generate_batch(np.asarray([[0, 0]], np.uint32), 8, 5)
    assert st["key"][0].tolist() == [4146024105, 967050713] and int(iters[0]) == 9
    assert st["start"][0].tolist() == [[4, 3], [5, 7], [3, 5], [2, 1], [3, 0]]
    assert st["target"][0].tolist() == [[0, 5], [7, 3], [0, 6], [2, 3], [6, 0]]
    assert solved[0][0].tolist() == [10, 10, 10, 0, 0, 3, 9, 7]
    assert solved[0][7].tolist() == [13, 13, 13, 6, 0, 0, 0, 0]
    keys = oracle.split(oracle.prng_key(0), 1)
    assert keys[0].tolist() == [928981903, 3453687069]
    st, solved, iters = oracle.generate_batch(keys, 8, 5)
    assert st["key"][0].tolist() == [3152301319, 4230269698] and int(iters[0]) == 12
    assert st["start"][0].tolist() == [[2, 6], [6, 7], [4, 7], [0, 4], [3, 1]]
    assert st["target"][0].tolist() == [[3, 7], [6, 3], [6, 1], [0, 7], [7, 7]]
    assert solved[0][7].tolist() == [13, 13, 13, 13, 13, 13, 13, 15]


def test_philox4x32_10_known_answers(oracle):
    """Random123's published known-answer vectors for philox4x32-10 (kat_vectors): the opt-in
    stream's block function is the standard one."""
    kat = [((0, 0), (0, 0, 0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 2, (0xffffffff,) * 4, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0xa4093822, 0x299f31d0), (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for key, ctr, exp in kat:
        assert tuple(int(x) for x in oracle.philox4x32_10(key, ctr)) == exp


def test_philox_stream_boards_are_valid(oracle):
    from helpers import check_solved_boards
    keys = oracle.split(oracle.prng_key(0), 512)
    st, solved, iters = oracle.prw_philox_batch(keys, 16, 10, threads=4)
    check_solved_boards(solved, st["start"], st["target"])
    assert 20 < iters.mean() < 35  # the same walk: trip counts distributed like the jax stream's
