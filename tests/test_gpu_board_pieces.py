"""GPU tests of the "private" methods of ParallelRandomWalkBoard: the
reference's own unit tests (JAXI/board_generation/test_parallel_random_walk_board.py), call for
call, through the HIP kernels of csrc/rbg_pieces.h -- expected values are the reference's
(tests/golden/reference_goldens.json::test_file, or the literals of its parametrize lists) --
followed by oracle <-> HIP comparisons on random mid-walk states and arbitrary arguments, and by
the check that the pieces compose to what the fused generate_board kernel returns.
"""
import numpy as np
import pytest

from helpers import to_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rbg():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (there is no CPU fallback to hide behind)")
    import routing_board_generation_amd as pkg

    pkg._native.lib()
    return pkg


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    return torch


@pytest.fixture(scope="module")
def board(rbg):  # the reference's fixture: ParallelRandomWalkBoard(rows=5, cols=5, num_agents=3)
    return rbg.ParallelRandomWalkBoard(5, 5, 3)


@pytest.fixture(scope="module")
def tf(goldens):
    return goldens["test_file"]


def _grid(torch, g):
    return torch.tensor(np.asarray(g, np.int32), device="cuda")


def _agent(rbg, torch, d):
    i32 = dict(dtype=torch.int32, device="cuda")
    return rbg.Agent(id=torch.tensor(d["id"], **i32), start=torch.tensor(d["start"], **i32),
                     target=torch.tensor(d["target"], **i32),
                     position=torch.tensor(d["position"], **i32))


def _agents_equal(a, d):
    return all((to_np(getattr(a, k)) == np.asarray(d[k])).all()
               for k in ("id", "start", "target", "position"))


# ------------------------------------------------------------ the reference's unit tests, on HIP
def test_step(rbg, torch_mod, board, tf, oracle):  # ::test_step
    key = rbg.PRNGKey(0)
    new_key, new_grid, new_agents = board._step(
        (key, _grid(torch_mod, tf["valid_starting_grid"]),
         _agent(rbg, torch_mod, tf["agents_starting"])))
    assert _agents_equal(new_agents, tf["agents_starting_move_1_step_up"])
    assert (to_np(new_grid) == np.asarray(tf["valid_starting_grid_after_1_step"])).all()
    assert (to_np(new_key) == to_np(rbg.split(key))[1]).all()  # key_2 of the reference's test


def test_initialise_agents(rbg, torch_mod, board, tf):  # ::test_initialise_agents
    empty_grid = torch_mod.zeros((5, 5), device="cuda")  # float zeros, as in the reference's test
    grid, agents = board._initialise_agents(rbg.PRNGKey(0), empty_grid)
    assert _agents_equal(agents, tf["agents_starting"])
    assert (to_np(grid) == np.asarray(tf["valid_starting_grid"])).all()


def test_place_agent_heads_on_grid(rbg, torch_mod, board, tf):  # ::test_place_agent_heads_on_grid
    grids = board._place_agent_heads_on_grid(torch_mod.zeros((5, 5), device="cuda"),
                                             _agent(rbg, torch_mod, tf["agents_starting"]))
    exp = np.zeros((3, 5, 5))
    exp[0, 2, 0], exp[1, 2, 4], exp[2, 4, 2] = 2, 5, 8
    assert (to_np(grids) == exp).all()
    one = board._place_agent_heads_on_grid(
        torch_mod.zeros((5, 5), device="cuda"),
        rbg.Agent(*(x[1] for x in _agent(rbg, torch_mod, tf["agents_starting"]))))
    assert (to_np(one) == exp[1]).all()


@pytest.mark.parametrize("grid_name,agents_name,expected",
                         [("valid_end_grid", "agents_finished", False),
                          ("valid_starting_grid", "agents_starting", True)])
def test_continue_stepping(rbg, torch_mod, board, tf, grid_name, agents_name, expected):
    out = board._continue_stepping((rbg.PRNGKey(0), _grid(torch_mod, tf[grid_name]),
                                    _agent(rbg, torch_mod, tf[agents_name])))
    assert bool(out) == expected


@pytest.mark.parametrize("grid_name,agents_name,expected",
                         [("valid_end_grid", "agents_finished", [True, True, True]),
                          ("valid_starting_grid", "agents_starting", [False, False, False])])
def test_is_any_step_possible(rbg, torch_mod, board, tf, grid_name, agents_name, expected):
    dones = board._is_any_step_possible(_grid(torch_mod, tf[grid_name]),
                                        _agent(rbg, torch_mod, tf[agents_name]))
    assert to_np(dones).tolist() == expected
    agents = _agent(rbg, torch_mod, tf[agents_name])
    one = board._is_any_step_possible(_grid(torch_mod, tf[grid_name]),
                                      rbg.Agent(*(x[2] for x in agents)))
    assert bool(one) == expected[2]


@pytest.mark.parametrize("flat,tup", [(7, [1, 2]), (24, [4, 4]), (1, [0, 1])])
def test_convert_positions(torch_mod, board, flat, tup):  # ::test_convert_*_position*
    assert to_np(board._convert_flat_position_to_tuple(torch_mod.tensor(flat))).tolist() == tup
    assert int(board._convert_tuple_to_flat_position(torch_mod.tensor(tup))) == flat


@pytest.mark.parametrize("p1,p2,expected", [(0, 1, 2), (1, 0, 4), (0, 5, 3), (5, 0, 1)])
def test_action_from_position(board, p1, p2, expected):  # ::test_action_from_position
    assert int(board._action_from_positions(p1, p2)) == expected


@pytest.mark.parametrize("tup,expected", [([1, 0], 3), ([-1, 0], 1), ([0, -1], 4), ([0, 1], 2),
                                          ([0, 0], 0)])
def test_action_from_tuple(torch_mod, board, tup, expected):  # ::test_action_from_tuple
    assert int(board._action_from_tuple(torch_mod.tensor(tup))) == expected


@pytest.mark.parametrize("cell,expected", [(0, [-1, 5, -1, 1]), (6, [1, 11, 5, 7])])
def test_adjacent_cells(board, cell, expected):  # ::test_adjacent_cells
    assert to_np(board._adjacent_cells(cell)).tolist() == expected


@pytest.mark.parametrize("grid_name,cell,expected",
                         [("valid_end_grid", 8, [-1, -1, -1, -1]),
                          ("grid_to_test_available_cells", 8, [-1, 13, -1, -1])])
def test_available_cells(torch_mod, board, tf, grid_name, cell, expected):  # ::test_available_cells
    assert to_np(board._available_cells(_grid(torch_mod, tf[grid_name]), cell)).tolist() == expected


@pytest.mark.parametrize("cell,expected", [(21, True), (22, False), (32, False)])
def test_is_cell_free(torch_mod, board, tf, cell, expected):  # ::test_is_cell_free
    grid, free = board._is_cell_free(_grid(torch_mod, tf["valid_starting_grid"]), cell)
    assert bool(free) == expected and grid.shape == (5, 5)


def test_step_agent(rbg, torch_mod, board, tf):  # ::test_step_agent, vmapped over the agents
    new_agents, new_grids = board._step_agent(_agent(rbg, torch_mod, tf["agents_starting"]),
                                              _grid(torch_mod, tf["valid_starting_grid"]),
                                              torch_mod.tensor([1, 1, 1]))
    assert _agents_equal(new_agents, tf["agents_starting_move_1_step_up"])
    assert (to_np(new_grids) == np.asarray(tf["grids_after_1_agent_step"])).all()
    agents = _agent(rbg, torch_mod, tf["agents_starting"])
    one, grid = board._step_agent(rbg.Agent(*(x[1] for x in agents)),
                                  _grid(torch_mod, tf["valid_starting_grid"]), 1)
    assert to_np(one.position).tolist() == tf["agents_starting_move_1_step_up"]["position"][1]
    assert (to_np(grid) == np.asarray(tf["grids_after_1_agent_step"][1])).all()


@pytest.mark.parametrize("position,expected", [([1, 2], True), ([1, 1], True), ([-1, 1], False)])
def test_is_valid_position_rw(rbg, torch_mod, board, tf, position, expected):
    valid = board._is_valid_position(_grid(torch_mod, tf["valid_starting_grid"]),
                                     _agent(rbg, torch_mod, tf["agents_starting"]),
                                     torch_mod.tensor(position))
    assert bool(valid.all()) == expected


def test_generate_board_from_the_pieces(rbg, torch_mod, board, tf):
    """generate_board (:60-90) written out with the pieces, as the reference does, equals the
    reference's golden board and the fused kernel's."""
    key, step_key = rbg.split(rbg.PRNGKey(0))
    grid, agents = board._initialise_agents(key, board._return_blank_board())
    carry, trips = (step_key, grid, agents), 0
    while bool(board._continue_stepping(carry)):
        carry = board._step(carry)
        trips += 1
    _, grid, agents = carry
    heads, targets = agents.start.T, agents.position.T
    solved = board.update_solved_board_with_head_target_encodings(grid, tuple(heads), tuple(targets))
    assert trips == 6
    assert (to_np(solved) == np.asarray(tf["valid_end_grid2"])).all()
    fh, ft, fg = board.generate_board(rbg.PRNGKey(0))
    assert (to_np(fg) == to_np(solved)).all()
    assert (to_np(fh) == to_np(heads)).all() and (to_np(ft) == to_np(targets)).all()


# ------------------------------------------------------------------ oracle <-> HIP, random inputs
def _mid_walk_states(oracle, m, N, B, seed, max_steps):
    """B boards stopped after a random number of iterations of the oracle's walk."""
    rng = np.random.default_rng(seed)
    keys = oracle.split(oracle.prng_key(seed), B)
    out = []
    for b in range(B):
        k = oracle.split(keys[b])
        grid, agents = oracle.initialise_agents(k[0], m, N)
        step_key = k[1]
        for _ in range(int(rng.integers(0, max_steps + 1))):
            if not oracle.continue_stepping(grid, agents):
                break
            step_key, grid, agents = oracle.prw_step(step_key, grid, agents)
        out.append((step_key, grid, agents))
    return out


@pytest.mark.parametrize("m,N,B,seed", [(5, 3, 64, 0), (8, 5, 64, 1), (16, 10, 48, 2),
                                        (10, 5, 64, 3), (7, 4, 64, 4), (32, 20, 8, 5),
                                        (12, 40, 8, 6), (16, 64, 4, 7), (3, 2, 32, 8)])
def test_step_matches_oracle_on_mid_walk_states(rbg, torch_mod, oracle, m, N, B, seed):
    from routing_board_generation_amd import ops

    torch = torch_mod
    states = _mid_walk_states(oracle, m, N, B, seed, 3 * m)
    keys = torch.from_numpy(np.stack([s[0] for s in states]).view(np.int32)).cuda().view(torch.uint32)
    grid = torch.from_numpy(np.stack([s[1] for s in states])).cuda()
    pos = torch.from_numpy(np.stack([s[2]["position"] for s in states])).cuda()
    tgt = torch.from_numpy(np.stack([s[2]["target"] for s in states])).cuda()
    nk, ng, npos, act = ops.prw_step(keys, grid, tgt, pos, want_actions=True)
    # the same through one key per agent
    agent_keys = rbg.split(keys, N)
    act2 = ops.prw_select_action(agent_keys.contiguous(), grid, pos)
    assert (to_np(act) == to_np(act2)).all()
    # _step_agent on private grids, for the drawn actions
    p1, grids1, moved = ops.prw_step_agent(grid, tgt, pos, act)
    for b, (k, g, a) in enumerate(states):
        ek, eg, ea = oracle.prw_step(k, g, a)
        assert (to_np(nk[b]) == ek).all()
        assert (to_np(ng[b]) == eg).all(), f"board {b}"
        assert (to_np(npos[b]) == ea["position"]).all()
        sub = oracle.split(k, N)
        for i in range(N):
            assert int(act[b, i]) == oracle.select_action(sub[i], g, a, i)
        for i in range(min(N, 6)):
            epos, egrid = oracle.step_agent(i, a, g, int(act[b, i]))
            assert to_np(p1[b, i]).tolist() == epos and (to_np(grids1[b, i]) == egrid).all()
            assert bool(moved[b, i]) == (epos != a["position"][i].tolist())


@pytest.mark.parametrize("m,N,seed", [(5, 3, 0), (8, 5, 1), (16, 10, 2), (9, 7, 3)])
def test_cell_queries_match_oracle_for_arbitrary_cells(torch_mod, oracle, m, N, seed):
    """Cells off the grid, negative cells, cells of empty squares: the reference's floor
    division / wrap-and-clamp gather decide, and the kernels follow."""
    from routing_board_generation_amd import ops

    torch = torch_mod
    rng = np.random.default_rng(seed)
    states = _mid_walk_states(oracle, m, N, 8, 100 + seed, 2 * m)
    grid = torch.from_numpy(np.stack([s[1] for s in states])).cuda()
    Q = 48
    cells = rng.integers(-m - 3, m * m + m + 3, size=(8, Q)).astype(np.int32)
    wires = rng.integers(-1, N + 1, size=(8, Q)).astype(np.int32)
    out = ops.prw_cell_queries(grid, torch.from_numpy(cells).cuda(), m,
                               wire_id=torch.from_numpy(wires).cuda(),
                               want=("adjacent", "available", "is_free", "not_doubling_back"))
    own = ops.prw_cell_queries(grid, torch.from_numpy(cells).cuda(), m,
                               want=("not_doubling_back",))["not_doubling_back"]
    for b in range(8):
        g = states[b][1]
        for q in range(Q):
            c = int(cells[b, q])
            assert to_np(out["adjacent"][b, q]).tolist() == oracle.adjacent_cells(c, m, m).tolist()
            assert to_np(out["available"][b, q]).tolist() == oracle.available_cells(g, c).tolist()
            assert bool(out["is_free"][b, q]) == oracle.is_cell_free(g, c)
            assert bool(out["not_doubling_back"][b, q]) == \
                oracle.not_doubling_back(g, int(wires[b, q]), c)
    assert own.shape == (8, Q)


@pytest.mark.parametrize("m,N,B,seed", [(5, 3, 100, 0), (16, 10, 256, 1), (10, 5, 100, 2),
                                        (40, 64, 8, 3), (2, 1, 16, 4), (7, 49, 16, 5)])
def test_initialise_agents_matches_oracle(rbg, torch_mod, oracle, m, N, B, seed):
    from routing_board_generation_amd import ops

    torch = torch_mod
    keys = rbg.split(rbg.PRNGKey(seed), B)
    grid, start = ops.prw_initialise_agents(keys, m, N)
    rng = np.random.default_rng(seed)
    busy = (rng.integers(0, 4, size=(B, m, m)) == 0) * rng.integers(1, 3 * N + 1, size=(B, m, m))
    busy = busy.astype(np.int32)
    grid2, start2 = ops.prw_initialise_agents(keys, m, N, torch.from_numpy(busy).cuda())
    assert (to_np(start2) == to_np(start)).all()
    kn = to_np(keys)
    for b in range(min(B, 64)):
        eg, ea = oracle.initialise_agents(kn[b], m, N)
        assert (to_np(grid[b]) == eg).all() and (to_np(start[b]) == ea["start"]).all()
        eg2, _ = oracle.initialise_agents(kn[b], m, N, grid=busy[b])
        if N > 1:  # (with one agent there is no other copy to max-join with: plain set)
            assert (to_np(grid2[b]) == eg2).all()
        else:
            r, c = ea["start"][0]
            assert int(grid2[b, r, c]) == 2


def test_step_agent_clamps_actions_and_respects_targets(rbg, torch_mod, oracle):
    """Out-of-range actions clamp like lax.switch; a connected agent does not move; an agent may
    enter its own target cell only."""
    from routing_board_generation_amd import ops

    torch = torch_mod
    g = np.zeros((1, 5, 5), np.int32)
    g[0, 2, 2], g[0, 1, 2], g[0, 2, 3], g[0, 0, 0] = 2, 3, 6, 5  # agent 0 + its target, 1's target
    pos = np.array([[[2, 2], [0, 0]]], np.int32)
    tgt = np.array([[[1, 2], [0, 0]]], np.int32)  # agent 1 is connected
    agents = {"id": [0, 1], "start": pos[0], "target": tgt[0], "position": pos[0]}
    for acts in ([1, 1], [2, 3], [7, -4], [0, 2], [3, 3], [4, 4]):
        p, grids, moved = ops.prw_step_agent(torch.from_numpy(g).cuda(), torch.from_numpy(tgt).cuda(),
                                             torch.from_numpy(pos).cuda(),
                                             torch.tensor([acts], dtype=torch.int32, device="cuda"))
        for i in range(2):
            epos, egrid = oracle.step_agent(i, agents, g[0], acts[i])
            assert to_np(p[0, i]).tolist() == epos, (acts, i)
            assert (to_np(grids[0, i]) == egrid).all(), (acts, i)


def test_pieces_reject_bad_arguments(rbg, torch_mod, board):
    from routing_board_generation_amd import ops

    torch = torch_mod
    with pytest.raises(ValueError):
        board._available_cells(torch.zeros((4, 4), device="cuda"), 3)  # not this board's shape
    with pytest.raises(ValueError):
        ops.prw_cell_queries(None, torch.zeros((1, 2), dtype=torch.int32), 5)  # CPU tensor
    with pytest.raises(rbg._native.NativeLibraryError):
        ops.prw_cell_queries(None, torch.zeros((1, 2), dtype=torch.int32, device="cuda"), 5,
                             want=("available",))  # needs a grid
