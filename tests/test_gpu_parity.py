"""GPU parity tests: the HIP path (through the C-ABI) against the reference's goldens and against
the CPU oracle on the same seeded inputs.  Bit-exact everywhere: grids, indices, keys and trip
counts are integers; rewards/discounts/ratios are single IEEE float32 operations.

Run on the GPU box with `python -m pytest tests -m gpu`.
"""
import hashlib

import numpy as np
import pytest

from helpers import (assert_dicts_equal, check_solved_boards, state_to_np, timestep_to_np, to_np)

pytestmark = pytest.mark.gpu

WINDOW = [0, 1, 2, 7, 8, 9]
WINDOW_BOARDS = [0, 1, 2, 97, 98, 99]


@pytest.fixture(scope="module")
def rbg():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (there is no CPU fallback to hide behind)")
    import routing_board_generation_amd as pkg

    pkg._native.lib()  # raises if the in-tree .so is missing
    return pkg


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    return torch


def _keys(rbg, seed, B):
    return rbg.split(rbg.PRNGKey(seed), B)


# ------------------------------------------------------------------------------ PRNG
def test_split_matches_oracle_and_known_answer(rbg, oracle):
    ks = to_np(rbg.split(rbg.PRNGKey(0)))
    assert ks.tolist() == [[4146024105, 967050713], [2718843009, 1272950319]]
    for num in (1, 3, 7, 100, 4096):
        assert (to_np(rbg.split(rbg.PRNGKey(5), num)) == oracle.split(oracle.prng_key(5), num)).all()
    big = rbg.PRNGKey((123 << 32) | 77)
    assert to_np(big).tolist() == [123, 77]


def test_split_batch_matches_oracle(rbg, oracle):
    keys = _keys(rbg, 1, 33)
    got = to_np(rbg.split(keys, 3))
    exp = np.stack([oracle.split(k, 3) for k in to_np(keys)])
    assert (got == exp).all()


# ------------------------------------------------------------------------------ reference goldens
def test_generate_board_golden_5x5(rbg, goldens):
    """JAXI/test_parallel_random_walk_board.py::test_generate_board (key = PRNGKey(0))."""
    tf = goldens["test_file"]
    board = rbg.ParallelRandomWalkBoard(rows=5, cols=5, num_agents=3)
    heads, targets, grid = board.generate_board(rbg.PRNGKey(0))
    exp = tf["agents_reshaped_for_generator"]
    assert (to_np(grid) == np.asarray(tf["valid_end_grid2"])).all()
    assert (to_np(heads) == np.asarray(exp["start"])).all()
    assert (to_np(targets) == np.asarray(exp["position"])).all()


def test_generate_board_for_various_keys(rbg):
    """::test_generate_board_for_various_keys -- ten seeds give ten different boards."""
    board = rbg.ParallelRandomWalkBoard(5, 5, 3)
    boards = [to_np(board.generate_board(rbg.PRNGKey(i))[2]) for i in range(10)]
    for i in range(10):
        for j in range(i):
            assert not (boards[i] == boards[j]).all()


@pytest.mark.parametrize("name", ["parallel_random_walk", "uniform"])
def test_single_board_generators_golden(rbg, goldens, name):
    """profiling_generators.ipynb cells 13 / 4: generator(10, 5)(PRNGKey(0))."""
    s = goldens["notebook"]["single_10x10_5_PRNGKey0"][name]
    gen = {"parallel_random_walk": rbg.ParallelRandomWalkGenerator,
           "uniform": rbg.UniformRandomGenerator}[name](10, 5)
    state = gen(rbg.PRNGKey(0))
    assert (to_np(state.grid) == np.asarray(s["grid"])).all()
    assert (to_np(state.agents.start) == np.asarray(s["start"])).all()
    assert (to_np(state.agents.target) == np.asarray(s["target"])).all()
    assert (to_np(state.agents.position) == np.asarray(s["position"])).all()
    assert (to_np(state.agents.id) == np.arange(5)).all()
    assert (to_np(state.key) == np.asarray(s["key"], np.uint32)).all()
    assert int(state.step_count) == 0


@pytest.mark.parametrize("name", ["parallel_random_walk", "uniform"])
def test_batch100_generators_golden(rbg, goldens, name):
    """profiling_generators.ipynb cells 26 / 18: vmap(generator)(split(PRNGKey(0), 100))."""
    b = goldens["notebook"]["batch100_10x10_5_split_PRNGKey0"][name]
    gen = {"parallel_random_walk": rbg.ParallelRandomWalkGenerator,
           "uniform": rbg.UniformRandomGenerator}[name](10, 5)
    state = gen(_keys(rbg, 0, 100))
    assert (to_np(state.agents.start) == np.asarray(b["start"])).all()
    assert (to_np(state.agents.target) == np.asarray(b["target"])).all()
    assert (to_np(state.agents.position) == np.asarray(b["position"])).all()
    assert (to_np(state.key) == np.asarray(b["key"], np.uint32)).all()
    win = to_np(state.grid)[WINDOW_BOARDS][:, WINDOW][:, :, WINDOW]
    assert (win == np.asarray(b["grid_window"])).all()


# ------------------------------------------------------------------------------ C1 vectors
# the two 8x8 / 5-agent boards of BASELINE configs[0], derived by the
# survey's independent emulation (which reproduces the reference goldens G1-G7).
B1_SINGLE = dict(
    key_in=(0, 0), state_key=(4146024105, 967050713), T=9,
    start=[[4, 3], [5, 7], [3, 5], [2, 1], [3, 0]], target=[[0, 5], [7, 3], [0, 6], [2, 3], [6, 0]],
    solved=[[10, 10, 10, 0, 0, 3, 9, 7], [10, 0, 10, 10, 1, 1, 0, 7], [10, 11, 0, 12, 1, 0, 7, 7],
            [14, 0, 0, 1, 1, 8, 7, 0], [13, 13, 0, 2, 0, 0, 0, 0], [0, 13, 13, 0, 0, 0, 0, 5],
            [15, 0, 13, 4, 4, 4, 4, 4], [13, 13, 13, 6, 0, 0, 0, 0]])
B1_BATCH1 = dict(
    key_in=(928981903, 3453687069), state_key=(3152301319, 4230269698), T=12,
    start=[[2, 6], [6, 7], [4, 7], [0, 4], [3, 1]], target=[[3, 7], [6, 3], [6, 1], [0, 7], [7, 7]],
    solved=[[0, 0, 0, 0, 11, 10, 0, 12], [0, 0, 0, 0, 0, 10, 10, 10], [0, 0, 0, 0, 0, 0, 2, 1],
            [0, 14, 0, 0, 7, 7, 7, 3], [13, 13, 0, 0, 7, 0, 7, 8], [13, 0, 7, 7, 7, 4, 4, 4],
            [13, 9, 7, 6, 4, 4, 0, 5], [13, 13, 13, 13, 13, 13, 13, 15]])


@pytest.mark.parametrize("vec", [B1_SINGLE, B1_BATCH1], ids=["PRNGKey0", "split_PRNGKey0_1"])
def test_c1_boards_of_survey_appendix_b1(rbg, oracle, torch_mod, vec):
    from routing_board_generation_amd import ops

    if vec is B1_BATCH1:  # keys = split(PRNGKey(0), 1)
        assert tuple(to_np(_keys(rbg, 0, 1))[0].tolist()) == vec["key_in"]
    key = torch_mod.tensor([vec["key_in"]], dtype=torch_mod.uint32, device="cuda")
    out = ops.prw_generate(key, 8, 5, split_first=True, want_solved=True, want_train=True,
                           want_iters=True)
    st, solved, iters = oracle.generate_batch(np.asarray([vec["key_in"]], np.uint32), 8, 5)
    for got_solved, got_start, got_target, got_key, got_T in (
            (to_np(out["solved_grid"])[0], to_np(out["start"])[0], to_np(out["target"])[0],
             to_np(out["key_out"])[0], int(to_np(out["iters"])[0])),
            (solved[0], st["start"][0], st["target"][0], st["key"][0], int(iters[0]))):
        assert (got_solved == np.asarray(vec["solved"])).all()
        assert got_start.tolist() == vec["start"] and got_target.tolist() == vec["target"]
        assert tuple(got_key.tolist()) == vec["state_key"] and got_T == vec["T"]
    train = np.asarray(vec["solved"]).copy()
    train[train % 3 == 1] = 0
    assert (to_np(out["train_grid"])[0] == train).all()


# ------------------------------------------------------------------------------ oracle parity
CONFIGS = [
    # (m, N, B, seed)          BASELINE configs: C1 8x8/5/B=1, C2 16x16/10/4096, C4 shape 32x32/20
    (8, 5, 1, 0), (16, 10, 4096, 0), (32, 20, 256, 0),
    (5, 3, 64, 1), (10, 5, 100, 0), (7, 4, 37, 2), (9, 9, 50, 3), (2, 1, 5, 4), (3, 9, 6, 5),
    (12, 64, 16, 6), (40, 64, 8, 7), (33, 17, 9, 8), (16, 1, 7, 9), (4, 16, 3, 10),
]


@pytest.mark.parametrize("m,N,B,seed", CONFIGS)
def test_prw_generator_matches_oracle(rbg, oracle, m, N, B, seed):
    from routing_board_generation_amd import ops

    keys = _keys(rbg, seed, B)
    out = ops.prw_generate(keys, m, N, split_first=True, want_solved=True, want_train=True,
                           want_iters=True)
    st, solved, iters = oracle.generate_batch(to_np(keys), m, N, threads=8)
    got = {"grid": to_np(out["train_grid"]), "start": to_np(out["start"]),
           "target": to_np(out["target"]), "key": to_np(out["key_out"])}
    assert_dicts_equal(got, {k: st[k] for k in got}, f"{m}x{m}/{N}: ")
    assert (to_np(out["solved_grid"]) == solved).all()
    assert (to_np(out["iters"]) == iters).all()


@pytest.mark.parametrize("m,N,B,seed", [(8, 5, 1, 0), (16, 10, 512, 3), (10, 5, 100, 0)])
def test_generate_board_matches_oracle(rbg, oracle, m, N, B, seed):
    keys = _keys(rbg, seed, B)
    heads, targets, grid, iters = rbg.ParallelRandomWalkBoard(m, m, N).generate_board_with_stats(keys)
    eh, et, es, ei = oracle.generate_board_batch(to_np(keys), m, N, threads=8)
    assert (to_np(heads) == eh.transpose(0, 2, 1)).all()
    assert (to_np(targets) == et.transpose(0, 2, 1)).all()
    assert (to_np(grid) == es).all()
    assert (to_np(iters) == ei).all()


def test_c2_batch_digests(rbg):
    """SHA-256 digests of the full BASELINE config-2 batch (derived
    cross-check data reproduced independently by the survey's emulation and by oracle/)."""
    from routing_board_generation_amd import ops

    out = ops.prw_generate(_keys(rbg, 0, 4096), 16, 10, split_first=True, want_iters=True)
    h = lambda t: hashlib.sha256(np.ascontiguousarray(to_np(t)).tobytes()).hexdigest()  # noqa
    assert h(out["train_grid"]) == "0a4e49ce3c80f710914af2143a28d6a25b32add93a6627e6a35fc0bd58204dfb"
    assert h(out["solved_grid"]) == "1b3383bf42acab1b1c4d3a43f5de126d7fe82b2e3778ea4fb93ce17b39f528ca"
    assert h(out["start"]) == "abdb3d790655b09f4e241936096ba476c6ac7b03be0a9c551137d166f7805ee6"
    assert h(out["target"]) == "34c740bd17c08aa8ef3893431e025b5c6563a679ae440185a148cfeb90802405"
    assert h(out["key_out"]) == "44f6f43143cc8f114be60dbb5ac4356f9a95790877396ef645e6fe8efe2444b5"
    assert h(out["iters"]) == "a847c96372f12e9b11e4c1c67ad5d98192c26d18c0a1e979df0889acb08ea5dc"


@pytest.mark.parametrize("m,N,B", [(16, 10, 4096), (32, 20, 4096)])
def test_full_size_properties(rbg, m, N, B):
    """Size-independent properties at BASELINE's full per-GPU sizes (C2, C4's per-GPU shard):
    wire validity of every solved board, training grid = heads + targets only, determinism,
    and shard invariance (a key range gives the same boards whichever batch it is part of)."""
    from routing_board_generation_amd import ops

    keys = _keys(rbg, 0, B)
    out = ops.prw_generate(keys, m, N, split_first=True, want_iters=True)
    solved, train = to_np(out["solved_grid"]), to_np(out["train_grid"])
    start, target, iters = to_np(out["start"]), to_np(out["target"]), to_np(out["iters"])
    check_solved_boards(solved, start, target)
    assert (train == np.where((solved % 3) == 1, 0, solved)).all()
    assert (iters >= 1).all() and (iters <= m * m - N).all()
    again = ops.prw_generate(keys, m, N, split_first=True)
    assert (to_np(again["solved_grid"]) == solved).all()
    lo, hi = B // 8 * 3, B // 8 * 4  # "rank 3 of 8"
    shard = ops.prw_generate(keys[lo:hi].contiguous(), m, N, split_first=True)
    assert (to_np(shard["solved_grid"]) == solved[lo:hi]).all()
    assert (to_np(shard["target"]) == target[lo:hi]).all()


@pytest.mark.parametrize("B", [0, 1, 2, 3, 5, 63, 65])
def test_ragged_batch_sizes(rbg, oracle, B, torch_mod):
    from routing_board_generation_amd import ops

    if B == 0:
        keys = torch_mod.empty((0, 2), dtype=torch_mod.uint32, device="cuda")
        out = ops.prw_generate(keys, 16, 10, split_first=True)
        assert out["solved_grid"].shape == (0, 16, 16)
        return
    keys = _keys(rbg, 11, B)
    out = ops.prw_generate(keys, 16, 10, split_first=True)
    st, solved, _ = oracle.generate_batch(to_np(keys), 16, 10)
    assert (to_np(out["solved_grid"]) == solved).all()
    assert (to_np(out["target"]) == st["target"]).all()


# ------------------------------------------------------------------------------ Connector
@pytest.mark.parametrize("m,N,B", [(16, 10, 256), (10, 5, 100), (7, 4, 19), (32, 20, 32),
                                   (12, 33, 24), (16, 64, 12), (10, 32, 16)])
def test_connector_reset_matches_oracle(rbg, oracle, m, N, B):
    env = rbg.Connector(generator=rbg.ParallelRandomWalkGenerator(m, N))
    keys = _keys(rbg, 2, B)
    state, ts = env.reset(keys)
    est, ets = oracle.connector_reset_batch(to_np(keys), m, N, threads=8)
    assert_dicts_equal(state_to_np(state), est, "state.")
    assert_dicts_equal(timestep_to_np(ts), ets, "timestep.")
    assert (to_np(ts.first())).all()


def test_connector_reset_single_env_shapes(rbg, oracle):
    env = rbg.Connector(generator=rbg.ParallelRandomWalkGenerator(10, 5))
    state, ts = env.reset(rbg.PRNGKey(0))
    assert state.grid.shape == (10, 10) and state.agents.start.shape == (5, 2)
    assert ts.observation.grid.shape == (5, 10, 10) and ts.observation.action_mask.shape == (5, 5)
    assert ts.reward.shape == (5,) and ts.step_type.shape == ()
    est, ets = oracle.connector_reset_batch(oracle.prng_key(0)[None], 10, 5)
    assert (to_np(ts.observation.grid) == ets["obs_grid"][0]).all()
    assert (to_np(state.grid) == est["grid"][0]).all()


@pytest.mark.parametrize("auto_reset", [False, True])
@pytest.mark.parametrize("m,N,B,steps", [(16, 10, 128, 60), (10, 5, 64, 60), (7, 4, 21, 25),
                                         # N >= 32: regenerate() takes the two- and three-
                                         # evaluation Threefry paths inside the env kernels
                                         (12, 33, 24, 45), (16, 64, 12, 45), (14, 40, 16, 45)])
def test_connector_rollout_matches_oracle(rbg, oracle, torch_mod, m, N, B, steps, auto_reset):
    """Random-legal-action rollout: state and timestep are compared after EVERY step, through
    connection events, collisions, blocked agents, the time limit and (auto_reset) fused resets."""
    from routing_board_generation_amd import ops

    env = rbg.Connector(generator=rbg.ParallelRandomWalkGenerator(m, N), time_limit=20)
    if auto_reset:
        env = rbg.VmapAutoResetWrapper(env, prefetch=False)  # the fused (inline) reset path
    keys = _keys(rbg, 4, B)
    akeys = _keys(rbg, 1, B)
    state, ts = env.reset(keys)
    est, ets = oracle.connector_reset_batch(to_np(keys), m, N, threads=8)
    n_last = 0
    for t in range(steps):
        action = ops.random_actions(akeys, t, ts.observation.action_mask)
        eaction = oracle.random_actions_batch(to_np(akeys), t, ets["action_mask"])
        assert (to_np(action) == eaction).all(), f"policy diverged at step {t}"
        state, ts = env.step(state, action)
        ets = oracle.connector_step_batch(est, eaction, time_limit=20, auto_reset=auto_reset,
                                          threads=8)
        assert_dicts_equal(state_to_np(state), est, f"step {t} state.")
        assert_dicts_equal(timestep_to_np(ts), ets, f"step {t} timestep.")
        n_last += int(to_np(ts.last()).sum())
    assert n_last > 0, "the rollout never reached a terminal step"
    if auto_reset:
        assert int(to_np(state.step_count).max()) <= 20


def test_connector_step_in_place_and_out_of_place_agree(rbg, torch_mod):
    from routing_board_generation_amd import ops

    keys = _keys(rbg, 8, 64)
    st, ts = ops.connector_reset(keys, 16, 10)
    action = ops.random_actions(keys, 0, ts["action_mask"])
    out_state, out_ts = ops.connector_step(st, action)
    inplace = {k: v.clone() for k, v in st.items()}
    _, in_ts = ops.connector_step(inplace, action, state_out=inplace)
    for k in out_state:
        assert torch_mod.equal(out_state[k].view(torch_mod.int32) if k == "key" else out_state[k],
                               inplace[k].view(torch_mod.int32) if k == "key" else inplace[k]), k
    assert torch_mod.equal(out_ts["obs_grid"], in_ts["obs_grid"])


def test_collision_kat_through_connector_step(rbg, torch_mod):
    """SURVEY.md 4.2 collision KAT (hand-derived from JAXI/parallel_random_walk.py:104-107,128):
    two and three agents contest one cell; the highest id wins, the others stay untouched."""
    from routing_board_generation_amd import ops

    dev = "cuda"
    grid = torch_mod.zeros((2, 5, 5), dtype=torch_mod.int32, device=dev)
    for b in range(2):
        grid[b, 2, 0], grid[b, 2, 1], grid[b, 2, 3], grid[b, 1, 2] = 1, 2, 5, 8
    pos = torch_mod.tensor([[[2, 1], [2, 3], [1, 2]]] * 2, dtype=torch_mod.int32, device=dev)
    st = {"grid": grid, "step_count": torch_mod.zeros(2, dtype=torch_mod.int32, device=dev),
          "start": pos.clone(), "target": torch_mod.full_like(pos, -1), "position": pos.clone(),
          "key": torch_mod.zeros((2, 2), dtype=torch_mod.uint32, device=dev)}
    action = torch_mod.tensor([[2, 4, 0], [2, 4, 3]], dtype=torch_mod.int32, device=dev)
    so, _ = ops.connector_step(st, action)
    g = to_np(so["grid"])
    assert g[0, 2].tolist() == [1, 2, 5, 4, 0]
    assert to_np(so["position"])[0].tolist() == [[2, 1], [2, 2], [1, 2]]
    assert g[1, 1].tolist() == [0, 0, 7, 0, 0]
    assert g[1, 2].tolist() == [1, 2, 8, 5, 0]
    assert to_np(so["position"])[1].tolist() == [[2, 1], [2, 3], [2, 2]]


def test_private_env_methods_used_by_the_reference(rbg, oracle):
    """demos/board_generator_demo.py:80-96 builds a TimeStep by hand from _get_action_mask,
    _obs_from_grid and _get_extras."""
    env = rbg.Connector(generator=rbg.UniformRandomGenerator(10, 5))
    state = rbg.UniformRandomGenerator(10, 5)(rbg.PRNGKey(3))
    mask = env._get_action_mask(state.agents, state.grid)
    obs = env._obs_from_grid(state.grid)
    extras = env._get_extras(state)
    est, _, _ = oracle.generate_batch(oracle.prng_key(3)[None], 10, 5, generator="uniform")
    ets = oracle.connector_observe_batch(est)
    assert (to_np(mask).astype(np.uint8) == ets["action_mask"][0]).all()
    assert (to_np(obs) == ets["obs_grid"][0]).all()
    assert int(extras["num_connections"]) == int(ets["num_connections"][0])
    assert int(extras["total_path_length"]) == 0
    # the reference's own call shape: one agent at a time (vmap over agents, :85-87)
    for i in range(5):
        one = rbg.Agent(id=state.agents.id[i], start=state.agents.start[i],
                        target=state.agents.target[i], position=state.agents.position[i])
        m1 = env._get_action_mask(one, state.grid)
        assert m1.shape == (5,) and (to_np(m1).astype(np.uint8) == ets["action_mask"][0, i]).all()
    # ... and a batch of grids with one agent each
    bstate = rbg.UniformRandomGenerator(10, 5)(_keys(rbg, 3, 7))
    est7, _, _ = oracle.generate_batch(to_np(_keys(rbg, 3, 7)), 10, 5, generator="uniform")
    ets7 = oracle.connector_observe_batch(est7)
    one = rbg.Agent(id=bstate.agents.id[:, 2], start=bstate.agents.start[:, 2],
                    target=bstate.agents.target[:, 2], position=bstate.agents.position[:, 2])
    mb = env._get_action_mask(one, bstate.grid)
    assert mb.shape == (7, 5) and (to_np(mb).astype(np.uint8) == ets7["action_mask"][:, 2]).all()


def test_uniform_generator_env_with_generic_auto_reset(rbg, oracle):
    """A Generator without a fused kernel goes through generator(key) + rbg_connector_observe."""
    env = rbg.VmapAutoResetWrapper(rbg.MultiToSingleWrapper(
        rbg.Connector(generator=rbg.UniformRandomGenerator(8, 4), time_limit=3)))
    keys = _keys(rbg, 6, 32)
    state, ts = env.reset(keys)
    assert ts.reward.shape == (32,)
    import torch

    for _ in range(3):
        state, ts = env.step(state, torch.zeros((32, 4), dtype=torch.int32, device="cuda"))
    assert to_np(ts.last()).all()           # time limit reached with NOOPs
    assert (to_np(state.step_count) == 0).all()  # and every env was regenerated
    exp_keys = np.stack([oracle.split(oracle.split(oracle.split(k)[0])[0])[0] for k in to_np(keys)])
    assert (to_np(state.key) == exp_keys).all()


def test_registry(rbg):
    from routing_board_generation_amd.interface.board_generator_interface import make_online_generator

    cls = rbg.BoardGenerator.get_board_generator(rbg.BoardName("offline_parallel_rw"))
    assert cls is rbg.ParallelRandomWalkBoard
    with pytest.raises(NotImplementedError):
        rbg.BoardGenerator.get_board_generator(rbg.BoardName("wfc"))
    gen = make_online_generator("online_parallel_rw", 16, 10)
    assert isinstance(gen, rbg.ParallelRandomWalkGenerator) and gen.grid_size == 16


# ------------------------------------------------------------------------------ fused rollout
@pytest.mark.parametrize("auto_reset", [False, True])
@pytest.mark.parametrize("m,N,B,steps", [(16, 10, 128, 45), (10, 5, 64, 45), (7, 4, 21, 30),
                                         (32, 20, 16, 25), (12, 33, 24, 40), (16, 64, 12, 40),
                                         (14, 40, 16, 40)])
def test_random_rollout_matches_oracle(rbg, oracle, m, N, B, steps, auto_reset):
    """rbg_connector_rollout == steps x { rbg_random_actions; rbg_connector_step }: final state,
    per-env counters and the timestep of the last step, against the oracle's step-by-step loop."""
    env = rbg.Connector(generator=rbg.ParallelRandomWalkGenerator(m, N), time_limit=15)
    if auto_reset:
        env = rbg.VmapAutoResetWrapper(env)
    keys, pkeys = _keys(rbg, 21, B), _keys(rbg, 22, B)
    state, _ = env.reset(keys)
    est, _ = oracle.connector_reset_batch(to_np(keys), m, N, threads=8)
    t0 = 7
    state, stats, ts = env.unwrapped.random_rollout(state, pkeys, t0, steps, want_timestep=True)
    estats, ets = oracle.connector_rollout_batch(est, to_np(pkeys), t0, steps, time_limit=15,
                                                 auto_reset=auto_reset, threads=8)
    assert_dicts_equal(state_to_np(state), est, "rollout state.")
    assert_dicts_equal({k: to_np(v) for k, v in stats.items()}, estats, "rollout stats.")
    assert_dicts_equal(timestep_to_np(ts), ets, "rollout last timestep.")
    assert int(to_np(stats["episodes"]).sum()) > 0


def test_random_rollout_equals_stepwise_gpu(rbg, torch_mod):
    """The fused rollout and the per-step API are the same function (chunking invariance too)."""
    from routing_board_generation_amd import ops

    B, m, N = 256, 16, 10
    keys, pkeys = _keys(rbg, 31, B), _keys(rbg, 32, B)
    st, ts = ops.connector_reset(keys, m, N)
    stepwise = {k: v.clone() for k, v in st.items()}
    mask = ts["action_mask"]
    for t in range(24):
        act = ops.random_actions(pkeys, t, mask)
        stepwise, sts = ops.connector_step(stepwise, act, auto_reset=True)
        mask = sts["action_mask"]
    fused, _, _ = ops.connector_rollout(st, pkeys, 0, 24, auto_reset=True)
    half, _, _ = ops.connector_rollout(st, pkeys, 0, 10, auto_reset=True)
    chunked, _, fts = ops.connector_rollout(half, pkeys, 10, 14, auto_reset=True, want_timestep=True)
    for k in ("grid", "step_count", "start", "target", "position"):
        assert torch_mod.equal(fused[k], stepwise[k]), k
        assert torch_mod.equal(chunked[k], stepwise[k]), k
    assert torch_mod.equal(fused["key"].view(torch_mod.int32), stepwise["key"].view(torch_mod.int32))
    assert torch_mod.equal(fts["obs_grid"], sts["obs_grid"])
    assert torch_mod.equal(fts["reward"], sts["reward"])


@pytest.mark.parametrize("auto_reset,prefetch", [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize("m,N,B,steps", [(16, 10, 96, 40), (10, 5, 64, 45), (7, 4, 21, 30),
                                         (32, 20, 8, 20), (12, 33, 12, 30), (16, 64, 6, 24)])
def test_recorded_rollout_matches_oracle_step_by_step(rbg, oracle, m, N, B, steps, auto_reset,
                                                      prefetch):
    """rbg_connector_rollout_record: entry s of the trajectory == what the s-th
    { rbg_random_actions; rbg_connector_step } of the loop returns, against the oracle's loop."""
    env = rbg.Connector(generator=rbg.ParallelRandomWalkGenerator(m, N), time_limit=15)
    if auto_reset:
        env = rbg.VmapAutoResetWrapper(env, prefetch=prefetch)
    keys, pkeys = _keys(rbg, 41, B), _keys(rbg, 42, B)
    state, ts0 = env.reset(keys)
    est, ets = oracle.connector_reset_batch(to_np(keys), m, N, threads=8)
    t0 = 3
    state, _, actions, traj = env.unwrapped.random_rollout_record(state, pkeys, t0, steps)
    assert actions.shape == (steps, B, N) and traj.observation.grid.shape == (steps, B, N, m, m)
    got = timestep_to_np(traj)
    mask = ets["action_mask"]
    for s in range(steps):
        a = oracle.random_actions_batch(to_np(pkeys), t0 + s, mask)
        assert (to_np(actions[s]) == a).all(), f"step {s}: actions"
        ets = oracle.connector_step_batch(est, a, time_limit=15, auto_reset=auto_reset, threads=8)
        assert_dicts_equal({k: v[s] for k, v in got.items()}, ets, f"step {s}: ")
        mask = ets["action_mask"]
    assert_dicts_equal(state_to_np(state), est, "final state.")
    assert int((got["step_type"] == 2).sum()) > 0


def test_recorded_rollout_at_full_size_equals_the_per_step_api(rbg, torch_mod):
    """BASELINE configs[2] shape: 4096 envs x 50 steps recorded in one launch == 50 launches of
    the per-step API (itself compared with the oracle at this size elsewhere)."""
    from routing_board_generation_amd import ops

    torch = torch_mod
    B, m, N, T = 4096, 16, 10, 50
    keys, pkeys = _keys(rbg, 51, B), _keys(rbg, 52, B)
    st, ts = ops.connector_reset(keys, m, N)
    so, _, traj = ops.connector_rollout_record(st, pkeys, 0, T)
    cur, mask = {k: v.clone() for k, v in st.items()}, ts["action_mask"]
    for t in range(T):
        act = ops.random_actions(pkeys, t, mask)
        cur, sts = ops.connector_step(cur, act, auto_reset=True)
        mask = sts["action_mask"]
        assert torch.equal(traj["action"][t], act), t
        for k in ("obs_grid", "action_mask", "reward", "discount", "step_type", "obs_step_count",
                  "num_connections", "ratio_connections", "total_path_length"):
            assert torch.equal(traj[k][t], sts[k]), (t, k)
    for k in ("grid", "step_count", "start", "target", "position"):
        assert torch.equal(so[k], cur[k]), k


# ------------------------------------------------------------------------------ offline dataset
def test_board_dataset_generator(rbg, oracle):
    """BoardDatasetGeneratorJAX (dataset_generator_jax.py:20-141): dataset = generate_board over
    split(PRNGKey(0), n) (pinned through the walk); sampling follows jax.random.randint
    (parity unpinned: no reference golden exercises it)."""
    from routing_board_generation_amd import ops

    n, m, N = 300, 10, 5
    # the reference's constructor call, setup_train.py:121-130 (seed-extension keywords ignored)
    gen = rbg.BoardDatasetGeneratorJAX(grid_size=m, num_agents=N, randomness=0, two_sided=True,
                                       extension_iterations=1, extension_steps=1e23,
                                       board_name="offline_parallel_rw", number_of_boards=n)
    assert rbg.BoardDatasetGenerator is rbg.BoardDatasetGeneratorJAX and len(gen) == n
    eh, et, es, _ = oracle.generate_board_batch(oracle.split(oracle.prng_key(0), n), m, N, threads=8)
    # heads/targets as the reference holds them: (n_boards, 2, N)  (dataset_generator_jax.py:61-62)
    assert gen.heads.shape == (n, 2, N) and gen.targets.shape == (n, 2, N)
    assert (to_np(gen.heads) == eh.transpose(0, 2, 1)).all()
    assert (to_np(gen.targets) == et.transpose(0, 2, 1)).all()
    assert (to_np(gen.solved_boards) == es).all()
    keys = _keys(rbg, 12, 257)
    st, which = ops.dataset_sample(keys, gen._heads, gen._targets, m, want_which=True)
    est, ewhich = oracle.dataset_sample_batch(to_np(keys), eh, et, m)
    assert (to_np(which) == ewhich).all()
    assert len(set(ewhich.tolist())) > 100
    got = {k: to_np(v) for k, v in st.items()}
    assert_dicts_equal(got, est, "dataset state.")
    single = gen(keys[3])
    assert single.grid.shape == (m, m) and (to_np(single.grid) == est["grid"][3]).all()
    env = rbg.Connector(generator=gen)
    state, ts = env.reset(keys[:8])
    assert (to_np(state.grid) == est["grid"][:8]).all() and ts.observation.grid.shape == (8, N, m, m)


def test_dataset_generate_n_boards(rbg, oracle):
    """BoardDatasetGeneratorJAX.generate_n_boards(key, n_boards, generate_solved_boards)
    (dataset_generator_jax.py:65-110) for a key other than PRNGKey(0): (heads, targets,
    solved_boards) of split(key, n) in the reference's (n, 2, N) layout, from one batched launch;
    the constructor's dataset is the same call on PRNGKey(0)."""
    m, N, n = 10, 5, 37
    key = rbg.PRNGKey(1234)
    ekeys = oracle.split(oracle.prng_key(1234), n)
    prw = rbg.BoardDatasetGeneratorJAX(m, N, board_name="offline_parallel_rw", number_of_boards=8)
    heads, targets, solved = prw.generate_n_boards(key, n)
    eh, et, es, _ = oracle.generate_board_batch(ekeys, m, N, threads=8)
    assert heads.shape == (n, 2, N) and targets.shape == (n, 2, N)
    assert (to_np(heads) == eh.transpose(0, 2, 1)).all()
    assert (to_np(targets) == et.transpose(0, 2, 1)).all()
    assert (to_np(solved) == es).all()
    assert len(prw) == 8 and prw.generate_n_boards(key)[0].shape == (10, 2, N)  # default n_boards
    d0 = prw.generate_n_boards(rbg.PRNGKey(0), 8)
    assert (to_np(d0[0]) == to_np(prw.heads)).all() and (to_np(d0[2]) == to_np(prw.solved_boards)).all()
    # the reference's default board_name with the config's keywords (connector.yaml:10-15)
    se = rbg.BoardDatasetGeneratorJAX(m, N, randomness=0.0, two_sided=True, extension_iterations=1,
                                      extension_steps=1e23, number_of_boards=4)
    heads, targets, solved = se.generate_n_boards(key, n)
    est = oracle.seed_extension_batch(ekeys, m, N, split_first=False, threads=8)[0]
    assert solved == []  # (:96-104: no solved boards unless asked for)
    assert (to_np(heads) == est["start"].transpose(0, 2, 1)).all()
    assert (to_np(targets) == est["target"].transpose(0, 2, 1)).all()
    _, _, solved = se.generate_n_boards(key, n, generate_solved_boards=True)
    esolved = oracle.seed_extension_batch(ekeys, m, N, split_first=False, threads=8)[1]
    assert (to_np(solved) == esolved).all()
    with pytest.raises(ValueError):
        se.generate_n_boards(rbg.split(key, 3), 4)  # one key, not a batch


# ------------------------------------------------------------------------------ SequentialRandomWalk
def test_sequential_random_walk_golden(rbg, goldens):
    """profiling_generators.ipynb cell 7 (G7): SequentialRandomWalkGenerator(10, 5)(PRNGKey(0))."""
    s = goldens["notebook"]["single_10x10_5_PRNGKey0"]["sequential_random_walk"]
    state = rbg.SequentialRandomWalkGenerator(10, 5)(rbg.PRNGKey(0))
    assert (to_np(state.grid) == np.asarray(s["grid"])).all()
    assert (to_np(state.agents.start) == np.asarray(s["start"])).all()
    assert (to_np(state.agents.target) == np.asarray(s["target"])).all()
    assert (to_np(state.key) == np.asarray(s["key"], np.uint32)).all()


@pytest.mark.parametrize("m,N,B,seed", [(16, 10, 4096, 0),  # BASELINE configs[4], full size
                                        (10, 5, 100, 0), (7, 4, 64, 1),
                                        (4, 7, 128, 1), (3, 2, 33, 2), (5, 6, 40, 3),
                                        (32, 20, 32, 4), (6, 3, 17, 5)])
def test_sequential_random_walk_matches_oracle(rbg, oracle, m, N, B, seed):
    """BASELINE configs[4] shape and edge cases (dense boards exercise retries and total failure,
    tiny grids the odd/even random_bits layouts)."""
    from routing_board_generation_amd import ops

    keys = _keys(rbg, seed, B)
    out = ops.seq_rw_generate(keys, m, N, split_first=True)
    est, esolved, eatt = oracle.seq_rw_generator_batch(to_np(keys), m, N, threads=8)
    assert (to_np(out["attempts"]) == eatt).all()
    assert (to_np(out["solved_grid"]) == esolved).all()
    got = {"grid": to_np(out["train_grid"]), "start": to_np(out["start"]),
           "target": to_np(out["target"]), "position": to_np(out["position"]),
           "key": to_np(out["key_out"]), "step_count": to_np(out["step_count"])}
    assert_dicts_equal(got, est, f"seq_rw {m}x{m}/{N}: ")


SEQ_SWEEP = [  # (m, N, log2 B at scale 1): sparse boards (one attempt), dense ones (most retries
    # resume from the cut wire, some replay, some boards fail for good), odd cell counts
    (16, 10, 14), (10, 5, 13), (8, 8, 13), (6, 5, 13), (10, 9, 12), (16, 14, 12), (5, 6, 12),
    (7, 7, 12), (12, 12, 11), (24, 20, 10), (9, 4, 12), (40, 30, 7),
]


@pytest.mark.parametrize("m,N,log2b", SEQ_SWEEP)
def test_many_sequential_boards_match_oracle(rbg, oracle, m, N, log2b):
    """The retry paths of the kernel (resume from the cut wire / replay when that wire was cut
    before / give up) are taken by a few boards in a hundred: large samples, dense shapes.
    RBG_SWEEP_SCALE multiplies the batch sizes."""
    import os

    from routing_board_generation_amd import ops

    B = (1 << log2b) * int(os.environ.get("RBG_SWEEP_SCALE", "1"))
    for lo in range(0, B, 32768):
        keys = _keys(rbg, 2000 + m * 64 + N + lo, min(32768, B - lo))
        out = ops.seq_rw_generate(keys, m, N, split_first=True)
        est, esolved, eatt = oracle.seq_rw_generator_batch(to_np(keys), m, N, threads=16)
        assert (to_np(out["attempts"]) == eatt).all()
        assert (to_np(out["solved_grid"]) == esolved).all()
        got = {"grid": to_np(out["train_grid"]), "start": to_np(out["start"]),
               "target": to_np(out["target"]), "position": to_np(out["position"]),
               "key": to_np(out["key_out"]), "step_count": to_np(out["step_count"])}
        assert_dicts_equal(got, est, f"seq_rw {m}x{m}/{N}: ")


def test_sequential_retries_without_step_history(rbg, oracle, torch_mod, slow_path_lib):
    """Shapes whose step history does not fit the LDS budget keep one record per wire and replay
    an attempt from the start when a wire has to be cut twice; a build without any history takes
    that path on every shape."""
    torch = torch_mod
    for (m, N, B, seed) in ((16, 10, 2048, 41), (8, 8, 2048, 42), (6, 5, 1024, 43), (10, 9, 1024, 44)):
        keys = _keys(rbg, seed, B)
        i32 = dict(dtype=torch.int32, device="cuda")
        solved, train = torch.empty((B, m, m), **i32), torch.empty((B, m, m), **i32)
        start, target = torch.empty((B, N, 2), **i32), torch.empty((B, N, 2), **i32)
        attempts = torch.empty((B,), **i32)
        rc = slow_path_lib.rbg_seq_rw_generate(
            keys.data_ptr(), B, m, N, 1, solved.data_ptr(), train.data_ptr(), start.data_ptr(),
            target.data_ptr(), None, None, None, attempts.data_ptr(),
            torch.cuda.current_stream().cuda_stream)
        assert rc == 0, slow_path_lib.rbg_last_error()
        est, esolved, eatt = oracle.seq_rw_generator_batch(to_np(keys), m, N, threads=16)
        assert (to_np(attempts) == eatt).all() and (eatt > 2).any()
        assert (to_np(solved) == esolved).all() and (to_np(train) == est["grid"]).all()
        assert (to_np(start) == est["start"]).all() and (to_np(target) == est["target"]).all()


def test_sequential_random_walk_board_api(rbg, oracle):
    board = rbg.SequentialRandomWalkBoard(10, 10, 5)
    keys = _keys(rbg, 9, 16)
    grids = board.generate(keys)
    (sr, sc), (er, ec) = board.generate_starts_ends(keys)
    assert grids.shape == (16, 10, 10) and sr.shape == (16, 5)
    g = to_np(grids)
    b = np.arange(16)[:, None]
    i = np.arange(5)[None, :]
    ok = to_np(board.generate(keys).sum((1, 2)) > 0)[:, None]
    assert ((g[b, to_np(sr), to_np(sc)] == 2 + 3 * i) | ~ok).all()
    assert ((g[b, to_np(er), to_np(ec)] == 3 + 3 * i) | ~ok).all()
    env = rbg.Connector(generator=rbg.SequentialRandomWalkGenerator(10, 5))
    state, ts = env.reset(keys)
    assert ts.observation.action_mask.shape == (16, 5, 5)


# ------------------------------------------------------------------------------ streams / graphs
def test_launches_follow_the_current_stream_and_capture_into_a_graph(rbg, torch_mod):
    """The C-ABI is asynchronous on the caller's stream with no hidden allocation or sync, so a
    rollout step can be captured into a hipGraph (torch.cuda.CUDAGraph) and replayed."""
    from routing_board_generation_amd import ops

    B, m, N = 512, 16, 10
    keys, pkeys = _keys(rbg, 41, B), _keys(rbg, 42, B)
    ref_state, ref_ts = ops.connector_reset(keys, m, N)
    expect, _, _ = ops.connector_rollout(ref_state, pkeys, 0, 12)

    side = torch_mod.cuda.Stream()
    side.wait_stream(torch_mod.cuda.current_stream())
    with torch_mod.cuda.stream(side):
        st, ts = ops.connector_reset(keys, m, N)
        act = torch_mod.empty((B, N), dtype=torch_mod.int32, device="cuda")
        # warm-up on the side stream, then capture one policy + step pair
        ops.random_actions(pkeys, 0, ts["action_mask"], out=act)
        ops.connector_step(st, act, auto_reset=True, state_out=st, timestep=ts)
        t_dev = 1
        graph = torch_mod.cuda.CUDAGraph()
        with torch_mod.cuda.graph(graph, stream=side):
            ops.random_actions(pkeys, t_dev, ts["action_mask"], out=act)
            ops.connector_step(st, act, auto_reset=True, state_out=st, timestep=ts)
    torch_mod.cuda.current_stream().wait_stream(side)
    torch_mod.cuda.synchronize()
    # the captured graph hard-codes t = 1, so replaying it is only the step at t = 1; run the
    # remaining steps eagerly and compare with the fused rollout
    graph.replay()
    for t in range(2, 12):
        ops.random_actions(pkeys, t, ts["action_mask"], out=act)
        ops.connector_step(st, act, auto_reset=True, state_out=st, timestep=ts)
    torch_mod.cuda.synchronize()
    for k in ("grid", "step_count", "position", "target"):
        assert torch_mod.equal(st[k], expect[k]), k


def test_large_batch_indexing(rbg, oracle):
    """65536 boards in one launch (8192 workgroups): spot-check 3 x 512 boards against the oracle
    and the whole batch for determinism of the per-board result (independent of batch position)."""
    from routing_board_generation_amd import ops

    B, m, N = 65536, 16, 10
    keys = _keys(rbg, 77, B)
    out = ops.prw_generate(keys, m, N, split_first=True, want_iters=True)
    solved = to_np(out["solved_grid"])
    for lo in (0, 30000, B - 512):
        st, esolved, eit = oracle.generate_batch(to_np(keys[lo:lo + 512]), m, N, threads=8)
        assert (solved[lo:lo + 512] == esolved).all()
        assert (to_np(out["target"])[lo:lo + 512] == st["target"]).all()
        assert (to_np(out["iters"])[lo:lo + 512] == eit).all()
    tail = ops.prw_generate(keys[B - 1000:].contiguous(), m, N, split_first=True)
    assert (to_np(tail["solved_grid"]) == solved[B - 1000:]).all()


# ------------------------------------------------------------------------------ C-ABI without Python
def test_c_api_demo_program(oracle, tmp_path):
    """examples/c_api_demo.cpp drives the library from plain C++ (no torch): same boards."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_api_demo")
    libdir = os.path.join(root, "routing_board_generation_amd", "lib")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-w", "-I",
                           os.path.join(root, "include"),
                           os.path.join(root, "examples", "c_api_demo.cpp"), "-L", libdir,
                           "-lrbg_prw", f"-Wl,-rpath,{libdir}", "-o", exe])
    out = subprocess.check_output([exe, "1000"], text=True)
    keys = oracle.split(oracle.prng_key(0), 1000)
    _, solved, iters = oracle.generate_batch(keys, 16, 10, threads=8)
    fnv = 1469598103934665603
    for v in solved.astype(np.uint32).ravel().tolist():
        fnv = ((fnv ^ v) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert f"walk_iters_sum {int(iters.sum())} " in out, out
    assert f"solved_fnv1a {fnv:016x}" in out, out
    assert "rollout 50 steps: episodes" in out
    # the second half of the demo: Connector(SequentialRandomWalkGenerator) under auto-reset, 20
    # NOOP steps at time_limit 6 -> three restarts per env, served by the generator's own stash
    st, _, _ = oracle.seq_rw_generator_batch(keys, 16, 10, threads=8)
    noop = np.zeros((1000, 10), np.int32)
    for _ in range(20):
        ts = oracle.connector_step_batch(st, noop, time_limit=6, auto_reset=False, threads=8)
        idx = np.nonzero(ts["step_type"] == 2)[0]
        if idx.size:
            fresh, _, _ = oracle.seq_rw_generator_batch(
                np.stack([oracle.split(k)[0] for k in st["key"][idx]]), 16, 10, threads=8)
            for f in st:
                st[f][idx] = fresh[f]
    fnv = 1469598103934665603
    for v in st["grid"].astype(np.uint32).ravel().tolist():
        fnv = ((fnv ^ v) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert f"grid_fnv1a {fnv:016x}" in out, out
    hits, misses = (int(x) for x in
                    out.split("stash hits ")[1].splitlines()[0].replace("misses", "").split())
    assert hits + misses == 3000 and hits >= 2000, out
    # the third part: the same env from a fresh reset with ROLLING refills driven from C++ (one
    # rbg_stash_roll per epoch, hipEvents of the program): 40 NOOP steps -> six restarts per env
    st, _, _ = oracle.seq_rw_generator_batch(keys, 16, 10, threads=8)
    for _ in range(40):
        ts = oracle.connector_step_batch(st, noop, time_limit=6, auto_reset=False, threads=8)
        idx = np.nonzero(ts["step_type"] == 2)[0]
        if idx.size:
            fresh, _, _ = oracle.seq_rw_generator_batch(
                np.stack([oracle.split(k)[0] for k in st["key"][idx]]), 16, 10, threads=8)
            for f in st:
                st[f][idx] = fresh[f]
    fnv = 1469598103934665603
    for v in st["grid"].astype(np.uint32).ravel().tolist():
        fnv = ((fnv ^ v) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert f"rolling_fnv1a {fnv:016x}" in out, out
    rhits = int(out.split("rolling hits ")[1].split()[0])
    rmiss = int(out.split("rolling misses ")[1].split()[0])
    assert rhits + rmiss == 6000 and rhits >= 4000, out


# ------------------------------------------------------------------------------ prefetched resets
@pytest.mark.parametrize("delay", [1, 4])
def test_prefetched_auto_reset_matches_oracle(rbg, oracle, delay):
    """VmapAutoResetWrapper(prefetch=True): next-episode States come from the stash;
    every state and timestep must equal the inline path (= the oracle), and the stash must
    actually be hit once it is warm."""
    from routing_board_generation_amd import ops

    m, N, B, steps, tl = 16, 10, 256, 70, 12
    env = rbg.VmapAutoResetWrapper(
        rbg.Connector(generator=rbg.ParallelRandomWalkGenerator(m, N), time_limit=tl),
        prefetch=True, prefetch_delay=delay)
    keys, akeys = _keys(rbg, 51, B), _keys(rbg, 52, B)
    state, ts = env.reset(keys)
    est, ets = oracle.connector_reset_batch(to_np(keys), m, N, threads=8)
    n_last = 0
    for t in range(steps):
        action = ops.random_actions(akeys, t, ts.observation.action_mask)
        eaction = oracle.random_actions_batch(to_np(akeys), t, ets["action_mask"])
        state, ts = env.step(state, action)
        ets = oracle.connector_step_batch(est, eaction, time_limit=tl, auto_reset=True, threads=8)
        assert_dicts_equal(state_to_np(state), est, f"step {t} state.")
        assert_dicts_equal(timestep_to_np(ts), ets, f"step {t} timestep.")
        n_last += int(to_np(ts.last()).sum())
    (sess,) = env.unwrapped._sessions.values()
    hits = int(to_np(sess.epochs.stash["hits"]).sum())
    assert n_last > 2 * B
    # an entry is usable at most 2 * delay launches after it was requested and nearly every
    # episode runs to the time limit of 12 steps: nearly every reset is a stash hit
    assert hits >= 0.9 * n_last, (hits, n_last)
    sess.drain()


def test_step_accepts_states_it_did_not_return(rbg, torch_mod):
    """Connector.step recognises the State it returned last by tensor identity and skips
    re-validating it; any other State (here: clones, and a State from two steps ago) takes the
    checked path.  Same function either way."""
    from routing_board_generation_amd import ops
    from routing_board_generation_amd.types import Agent, State

    m, N, B = 10, 5, 300
    env = rbg.VmapAutoResetWrapper(rbg.Connector(generator=rbg.ParallelRandomWalkGenerator(m, N),
                                                 time_limit=9))
    keys, akeys = _keys(rbg, 81, B), _keys(rbg, 82, B)
    s0, ts0 = env.reset(keys)
    a0 = ops.random_actions(akeys, 0, ts0.observation.action_mask)
    s1, ts1 = env.step(s0, a0)                      # fast path (s0 was returned last)
    a1 = ops.random_actions(akeys, 1, ts1.observation.action_mask)
    s2, ts2 = env.step(s1, a1)                      # fast path
    clone = State(grid=s1.grid.clone(), step_count=s1.step_count.clone(),
                  agents=Agent(*[x.clone() for x in s1.agents]), key=s1.key.clone())
    s2c, ts2c = env.step(clone, a1.to(torch_mod.int64))   # checked path, action cast to int32
    s1b, ts1b = env.step(s0, a0)                    # checked path: s0 is no longer the last one
    for a, b in ((s2, s2c), (s1, s1b)):
        assert_dicts_equal(state_to_np(a), state_to_np(b), "state.")
    for a, b in ((ts2, ts2c), (ts1, ts1b)):
        assert_dicts_equal(timestep_to_np(a), timestep_to_np(b), "timestep.")
    with pytest.raises(ValueError):
        env.step(s1, a1[:, :3])


@pytest.mark.parametrize("serial", [True, False])
def test_prefetched_auto_reset_async_stress(rbg, torch_mod, serial):
    """400 steps of 4096 envs without any host synchronisation, through the validated
    ops.connector_*_prefetch calls driven by ops.EpochStash (refill on the same stream, or on a
    side stream where it genuinely overlaps later steps): the stash path must end in exactly the
    state of the inline path."""
    from routing_board_generation_amd import ops

    m, N, B, steps, tl = 16, 10, 4096, 400, 9
    keys, akeys = _keys(rbg, 61, B), _keys(rbg, 62, B)
    st_a, ts_a = ops.connector_reset(keys, m, N)
    act = torch_mod.empty((B, N), dtype=torch_mod.int32, device="cuda")
    for t in range(steps):
        ops.random_actions(akeys, t, ts_a["action_mask"], out=act)
        ops.connector_step(st_a, act, time_limit=tl, auto_reset=True, state_out=st_a, timestep=ts_a)
    ep = ops.EpochStash(B, m, N, keys.device, every=3, serial=serial)
    ep.before_launch()
    st_b, ts_b = ops.connector_reset_prefetch(keys, m, N, ep.stash, ep.tick, ep.ready_delay())
    ep.after_launch()
    for t in range(steps):
        ops.random_actions(akeys, t, ts_b["action_mask"], out=act)
        ep.before_launch()
        ops.connector_step_prefetch(st_b, act, ep.stash, ep.tick, ep.ready_delay(), time_limit=tl,
                                    state_out=st_b, timestep=ts_b)
        ep.after_launch()
    ep.drain()
    torch_mod.cuda.synchronize()
    for k in ("grid", "step_count", "start", "target", "position"):
        assert torch_mod.equal(st_a[k], st_b[k]), k
    assert torch_mod.equal(st_a["key"].view(torch_mod.int32), st_b["key"].view(torch_mod.int32))
    assert torch_mod.equal(ts_a["obs_grid"], ts_b["obs_grid"])
    assert torch_mod.equal(ts_a["reward"], ts_b["reward"])
    hits = int(ep.stash["hits"].sum().item())
    assert hits > 10 * B, hits


@pytest.mark.parametrize("K,fuse,serial", [(0, False, True), (3, False, True), (3, True, True),
                                           (0, True, True), (5, True, True), (3, False, False),
                                           (4, True, False)])
def test_step_loop_equals_validated_path(rbg, torch_mod, K, fuse, serial):
    """ops.StepLoop (prebuilt launch descriptors, optional epoch stash refilled on the same or on
    a side stream, optional fused policy) is the same function as { random_actions;
    connector_step(auto_reset) } step by step."""
    from routing_board_generation_amd import ops

    m, N, B, steps, tl = 16, 10, 1024, 150, 11
    keys, pkeys = _keys(rbg, 71, B), _keys(rbg, 72, B)
    ref, rts = ops.connector_reset(keys, m, N)
    for t in range(steps):
        act = ops.random_actions(pkeys, t, rts["action_mask"])
        ref, rts = ops.connector_step(ref, act, time_limit=tl, auto_reset=True)
    st, ts = ops.alloc_state(B, m, N, "cuda"), ops.alloc_timestep(B, m, N, "cuda")
    loop = ops.StepLoop(st, ts, pkeys, time_limit=tl, refill_every=K, fuse_policy=fuse,
                        serial_refill=serial)
    loop.reset(keys)
    for t in range(steps):
        loop.step(t)
    loop.drain()
    torch_mod.cuda.synchronize()
    for k in ("grid", "step_count", "start", "target", "position"):
        assert torch_mod.equal(st[k], ref[k]), k
    assert torch_mod.equal(st["key"].view(torch_mod.int32), ref["key"].view(torch_mod.int32))
    for k in ("obs_grid", "action_mask", "reward", "discount", "step_type", "num_connections",
              "total_path_length"):
        assert torch_mod.equal(ts[k], rts[k]), k
    if K:
        assert int(loop.stash["hits"].sum().item()) > B


# ------------------------------------------------------------------------------ bench under RCCL
def test_bench_under_torchrun_with_a_one_rank_rccl_group():
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, backend nccl = RCCL),
    here with the one rank a one-GPU box allows: communicator init, barrier and the metrics
    all-reduce run for real, and the line keeps the contract's keys."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.check_output(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
         "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(root, "bench.py"),
         "--gpus", "1", "--steps", "20", "--warmup", "3", "--env-steps", "20", "--stream-batch",
         "0", "--rollout-steps", "10", "--cpu-sample", "0", "--force-process-group"],
        text=True, env=env, cwd=root, timeout=600)
    line = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
              "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 20 and line["value"] > 0
    assert line["config"]["global_batch"] == 4096


def test_bench_gpus_2_launches_itself_on_one_gpu():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts torchrun as a
    child and relays rank 0's line.  Two ranks share GPU 0 over gloo (--rehearse-single-gpu: RCCL
    refuses two ranks on one device), so the numbers mean nothing; the launch path, the sharding
    (2 x 4096 boards) and the max-over-ranks timing are what is exercised."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(
        [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10",
         "--warmup", "2", "--mode", "generator", "--cpu-sample", "0", "--rehearse-single-gpu"],
        capture_output=True, text=True, env=env, cwd=root, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 10 and line["value"] > 0
    assert line["config"]["global_batch"] == 8192 and line["scaling"] == "weak"
    # BASELINE configs[3] and the strong reading ride on the same line (bench.py: generator_block)
    c4, strong = line["c4"], line["strong"]
    assert (c4["grid_size"], c4["num_agents"], c4["global_batch"], c4["boards_per_gpu"]) == \
        (32, 20, 8192, 4096) and c4["value"] > 0 and c4["scaling"] == "weak"
    assert (strong["grid_size"], strong["num_agents"], strong["global_batch"],
            strong["boards_per_gpu"]) == (16, 10, 4096, 2048)
    assert strong["value"] > 0 and strong["scaling"] == "strong"
    assert c4["launch_ms_over_ranks"]["max"] >= c4["launch_ms_over_ranks"]["min"] > 0


@pytest.mark.parametrize("prefetch", [False, True])
def test_donated_stepping_matches_oracle(rbg, oracle, prefetch):
    """Connector(donate=True): the State is advanced in place and the TimeStep alternates between
    two buffer sets; values equal the functional path (= the oracle) step by step, and the
    timestep of step t is still intact after step t + 1."""
    from routing_board_generation_amd import ops

    m, N, B, steps, tl = 12, 6, 200, 40, 10
    env = rbg.VmapAutoResetWrapper(
        rbg.Connector(generator=rbg.ParallelRandomWalkGenerator(m, N), time_limit=tl, donate=True),
        prefetch=prefetch, prefetch_delay=3)
    keys, akeys = _keys(rbg, 91, B), _keys(rbg, 92, B)
    state, ts = env.reset(keys)
    est, ets = oracle.connector_reset_batch(to_np(keys), m, N, threads=8)
    grid_ptr = state.grid.data_ptr()
    prev_ts, prev_ets = None, None
    for t in range(steps):
        action = ops.random_actions(akeys, t, ts.observation.action_mask)
        eaction = oracle.random_actions_batch(to_np(akeys), t, ets["action_mask"])
        state, ts = env.step(state, action)
        ets = oracle.connector_step_batch(est, eaction, time_limit=tl, auto_reset=True, threads=8)
        assert state.grid.data_ptr() == grid_ptr
        assert_dicts_equal(state_to_np(state), est, f"step {t} state.")
        assert_dicts_equal(timestep_to_np(ts), ets, f"step {t} timestep.")
        if prev_ts is not None:
            assert_dicts_equal(timestep_to_np(prev_ts), prev_ets, f"step {t - 1} timestep, kept.")
        prev_ts, prev_ets = ts, {k: np.copy(v) for k, v in ets.items()}
    for sess in env.unwrapped._sessions.values():
        sess.drain()


@pytest.mark.parametrize("donate,prefetch", [(False, False), (True, True)])
def test_multi_to_single_aggregates_inside_the_launch(rbg, oracle, donate, prefetch):
    """The reference's training env, VmapAutoResetWrapper(MultiToSingleWrapper(Connector))
    (setup_train.py:158-167): reward summed over agents in float32 agent order, discount = max,
    produced by the step launch itself (rbg_timestep.reward_total / discount_max)."""
    from routing_board_generation_amd import ops

    m, N, B, steps, tl = 16, 10, 300, 60, 14
    env = rbg.VmapAutoResetWrapper(rbg.MultiToSingleWrapper(
        rbg.Connector(generator=rbg.ParallelRandomWalkGenerator(m, N), time_limit=tl,
                      donate=donate)), prefetch=prefetch, prefetch_delay=3)
    keys, akeys = _keys(rbg, 101, B), _keys(rbg, 102, B)
    state, ts = env.reset(keys)
    est, ets = oracle.connector_reset_batch(to_np(keys), m, N, threads=8)
    assert ts.reward.shape == (B,) and ts.discount.shape == (B,)
    assert (to_np(ts.reward) == 0).all() and (to_np(ts.discount) == 1).all()
    seen_nonzero = 0
    for t in range(steps):
        action = ops.random_actions(akeys, t, ts.observation.action_mask)
        eaction = oracle.random_actions_batch(to_np(akeys), t, ets["action_mask"])
        state, ts = env.step(state, action)
        ets = oracle.connector_step_batch(est, eaction, time_limit=tl, auto_reset=True, threads=8)
        total = np.zeros((B,), dtype=np.float32)
        for i in range(N):  # ((0 + r_0) + r_1) + ... in float32
            total = (total + ets["reward"][:, i]).astype(np.float32)
        assert (to_np(ts.reward).view(np.uint32) == total.view(np.uint32)).all(), t
        assert (to_np(ts.discount) == ets["discount"].max(axis=1)).all(), t
        assert np.allclose(to_np(ts.reward), ets["reward"].sum(axis=1), atol=1e-6)
        assert (to_np(ts.step_type) == ets["step_type"]).all()
        assert (to_np(ts.observation.grid) == ets["obs_grid"]).all()
        assert_dicts_equal(state_to_np(state), est, f"step {t} state.")
        seen_nonzero += int((total != 0).sum())
    assert seen_nonzero > B
    for sess in env.unwrapped._sessions.values():
        sess.drain()


# ------------------------------------------------------------------------------ many boards
SWEEP = [  # (m, N, log2 B at scale 1): all three sub-wave widths, both Threefry pipelines, the
    # register / LDS variants of the start-cell shuffle, odd and even cell counts
    (16, 10, 17), (8, 5, 15), (10, 5, 15), (12, 7, 14), (16, 15, 13), (32, 20, 12), (22, 12, 12),
    (23, 12, 12), (40, 15, 10), (9, 7, 13), (5, 6, 13), (20, 31, 11), (20, 32, 10),
    (20, 40, 10), (32, 33, 9), (40, 63, 8), (16, 64, 8),
]


@pytest.mark.parametrize("m,N,log2b", SWEEP)
def test_many_boards_match_oracle(rbg, oracle, torch_mod, m, N, log2b):
    """Rare paths (threshold retries of the start-cell shuffle, long walks, collisions) only show
    up in large samples.  RBG_SWEEP_SCALE=8 multiplies the batch sizes (2.1 M boards, ~20 s)."""
    import os

    from routing_board_generation_amd import ops

    B = (1 << log2b) * int(os.environ.get("RBG_SWEEP_SCALE", "1"))
    keys_all = oracle.split(oracle.prng_key(1000 + m * 64 + N), B)
    for lo in range(0, B, 65536):
        kn = keys_all[lo:lo + 65536]
        kt = torch_mod.from_numpy(kn.view(np.int32)).cuda().view(torch_mod.uint32)
        out = ops.prw_generate(kt, m, N, split_first=True, want_solved=True, want_train=True,
                               want_state=True, want_iters=True)
        st, solved, iters = oracle.generate_batch(kn, m, N, threads=16)
        assert (to_np(out["solved_grid"]) == solved).all()
        assert (to_np(out["train_grid"]) == st["grid"]).all()
        assert (to_np(out["start"]) == st["start"]).all()
        assert (to_np(out["target"]) == st["target"]).all()
        assert (to_np(out["iters"]) == iters).all()
        assert (to_np(out["key_out"]) == st["key"]).all()


def test_batches_beyond_two_gib_per_field(rbg, torch_mod):
    """Sized for 288 GB: 3 M boards in ONE generator launch (2.9 GiB per grid field) and 300 k
    envs in one reset / step launch (2.9 GiB of observation) -- every offset past 2^31 bytes.
    Boards are independent, so slices of the big launch must equal small launches of the same
    keys (shard invariance at a size the oracle is not needed for)."""
    from routing_board_generation_amd import ops

    B = 3_000_000
    keys = rbg.split(rbg.PRNGKey(7), B)
    kw = dict(split_first=True, want_solved=True, want_train=True, want_state=True, want_iters=True)
    out = ops.prw_generate(keys, 16, 10, **kw)
    for lo in (0, 1_500_000, 2_097_152 - 500, B - 1000):
        sub = ops.prw_generate(keys[lo:lo + 1000].contiguous(), 16, 10, **kw)
        for k in ("solved_grid", "train_grid", "start", "target", "key_out", "iters"):
            assert bool((out[k][lo:lo + 1000] == sub[k]).all()), (lo, k)
    del out
    B2, tail = 300_000, 512
    k2 = keys[:B2].contiguous()
    results = []
    for kk in (k2, k2[B2 - tail:].contiguous()):
        env = rbg.VmapAutoResetWrapper(rbg.Connector(rbg.ParallelRandomWalkGenerator(16, 10),
                                                     time_limit=2))
        policy = ops.RandomPolicy(kk, 10)
        state, ts = env.reset(kk)
        for t in range(3):  # (time_limit 2: the last step restarts every env from its stash row)
            state, ts = env.step(state, policy(t, ts.observation.action_mask))
        results.append((state, ts))
        for sess in env.unwrapped._sessions.values():
            sess.drain()
    (sb, tb), (ss, tsm) = results
    assert tb.observation.grid.numel() * 4 > 2 ** 31
    assert bool((sb.grid[B2 - tail:] == ss.grid).all()) and bool((sb.key[B2 - tail:] == ss.key).all())
    assert bool((tb.observation.grid[B2 - tail:] == tsm.observation.grid).all())
    assert bool((tb.reward[B2 - tail:] == tsm.reward).all())
    assert bool((tb.observation.action_mask[B2 - tail:] == tsm.observation.action_mask).all())


def test_random_agent_loop_example_runs():
    """examples/random_agent_loop.py: the reference's acting loop on the drop-in classes."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.check_output(
        [sys.executable, os.path.join(root, "examples", "random_agent_loop.py"), "--epochs", "2",
         "--batch-per-gpu", "512", "--n-steps", "60"], text=True, cwd=root, timeout=600)
    lines = [ln for ln in out.splitlines() if ln.startswith("epoch")]
    assert len(lines) == 2 and "episodes" in lines[1], out
    assert int(lines[1].split(" GPU(s), ")[1].split(" episodes")[0]) > 256, out
    # the same epochs collected by one recording launch each: same episodes, same connections
    out2 = subprocess.check_output(
        [sys.executable, os.path.join(root, "examples", "random_agent_loop.py"), "--epochs", "2",
         "--batch-per-gpu", "512", "--n-steps", "60", "--one-launch-per-epoch"], text=True,
        cwd=root, timeout=600)
    lines2 = [ln for ln in out2.splitlines() if ln.startswith("epoch")]
    counts = lambda ln: ln.split(" GPU(s), ")[1].split(", mean step reward")[0]  # noqa: E731
    assert [counts(ln) for ln in lines2] == [counts(ln) for ln in lines], (out, out2)


@pytest.mark.parametrize("m,N,B,steps,tl", [(16, 10, 4096, 110, 50),  # BASELINE configs[2], full size
                                            (8, 5, 4096, 100, 12), (12, 7, 1024, 90, 25),
                                            (32, 20, 256, 80, 40), (12, 33, 96, 60, 20)])
def test_long_auto_reset_rollouts_match_oracle(rbg, oracle, m, N, B, steps, tl):
    """Many envs over several episodes each (default wrapper: stash + refills): state and
    timestep against the oracle after every step."""
    from routing_board_generation_amd import ops

    env = rbg.VmapAutoResetWrapper(
        rbg.Connector(generator=rbg.ParallelRandomWalkGenerator(m, N), time_limit=tl))
    keys, akeys = _keys(rbg, 201, B), _keys(rbg, 202, B)
    state, ts = env.reset(keys)
    est, ets = oracle.connector_reset_batch(to_np(keys), m, N, threads=16)
    for t in range(steps):
        action = ops.random_actions(akeys, t, ts.observation.action_mask)
        eaction = oracle.random_actions_batch(to_np(akeys), t, ets["action_mask"])
        assert (to_np(action) == eaction).all(), t
        state, ts = env.step(state, action)
        ets = oracle.connector_step_batch(est, eaction, time_limit=tl, auto_reset=True, threads=16)
        assert_dicts_equal(state_to_np(state), est, f"step {t} state.")
        assert_dicts_equal(timestep_to_np(ts), ets, f"step {t} timestep.")
    for sess in env.unwrapped._sessions.values():
        sess.drain()


@pytest.mark.parametrize("m,N,B,n_steps,tl", [(16, 10, 1024, 50, 50), (10, 5, 777, 40, 9),
                                              (16, 10, 512, 120, 25)])
def test_rollout_with_two_slot_stash_equals_plain_rollout(rbg, torch_mod, m, N, B, n_steps, tl):
    """rbg_connector_rollout_prefetch with a depth-2 stash refilled after every launch: same
    states and counters as the plain fused rollout, and once warm every reset is a stash hit --
    including the second and third reset of an env within one launch (short time limits)."""
    from routing_board_generation_amd import ops

    keys, pkeys = _keys(rbg, 301, B), _keys(rbg, 302, B)
    launches = 6
    ref, _ = ops.connector_reset(keys, m, N)
    ref_stats = []
    for i in range(launches):
        ref, st, _ = ops.connector_rollout(ref, pkeys, i * n_steps, n_steps, time_limit=tl)
        ref_stats.append({k: v.clone() for k, v in st.items()})
    ep = ops.EpochStash(B, m, N, keys.device, every=1, serial=True, depth=2)
    ep.before_launch()
    cur, _ = ops.connector_reset_prefetch(keys, m, N, ep.stash, ep.tick, ep.ready_delay())
    ep.after_launch()
    episodes = 0
    for i in range(launches):
        before = int(ep.stash["hits"].sum().item())
        cur, st, _ = ops.connector_rollout(cur, pkeys, i * n_steps, n_steps, time_limit=tl,
                                           epochs=ep)
        for k in ops.ROLLOUT_STATS:
            assert torch_mod.equal(st[k], ref_stats[i][k]), (i, k)
        n_ep = int(st["episodes"].sum().item())
        hits = int(ep.stash["hits"].sum().item()) - before
        if i >= 1 and tl * 2 >= n_steps:  # at most two resets per env and launch: all served
            assert hits == n_ep, (i, hits, n_ep)
        episodes += n_ep
    for k in ("grid", "step_count", "start", "target", "position"):
        assert torch_mod.equal(cur[k], ref[k]), k
    assert torch_mod.equal(cur["key"].view(torch_mod.int32), ref["key"].view(torch_mod.int32))
    # two slots serve at most two resets per env and launch; the rest walk inline
    total_hits = int(ep.stash["hits"].sum().item())
    assert total_hits > (0.6 if tl * 2 >= n_steps else 0.3) * episodes, (total_hits, episodes)


def test_output_chunks_are_reused_only_when_unreferenced(rbg, torch_mod):
    """The functional step hands out fresh memory every step; a chunk of outputs is reused once
    nothing outside the session refers to it (Python references to the tensors handed out, or
    anything derived from them).  Tensors a caller keeps -- directly or through a slice -- must
    never change; a loop that keeps nothing must stop allocating."""
    from routing_board_generation_amd import ops

    torch = torch_mod
    m, N, B = 8, 3, 64
    env = rbg.VmapAutoResetWrapper(rbg.Connector(generator=rbg.ParallelRandomWalkGenerator(m, N),
                                                 time_limit=6))
    keys = _keys(rbg, 5, B)
    state, ts = env.reset(keys)
    (sess,) = env.unwrapped._sessions.values()
    sess.chunk_steps = 4  # small chunks: many turn-overs in a short test
    made = []
    new_chunk = sess._new_chunk
    sess._new_chunk = lambda: made.append(1) or new_chunk()
    policy = ops.RandomPolicy(keys, N)
    kept = []  # (tensor or slice kept alive, its value when it was returned)
    for t in range(120):
        state, ts = env.step(state, policy(t, ts.observation.action_mask))
        if t % 7 == 0:
            kept.append((ts.observation.grid, ts.observation.grid.clone()))
        if t % 11 == 0:
            kept.append((ts.reward[3:5], ts.reward[3:5].clone()))  # a derived view only
        if t % 13 == 0:
            kept.append((state.grid, state.grid.clone()))
    torch.cuda.synchronize()
    for held, snapshot in kept:
        assert torch.equal(held, snapshot), "a tensor the caller still holds was overwritten"
    n_with_refs = len(made)
    # now a loop that keeps nothing: after the pool has turned over no chunk is allocated any more
    kept.clear()
    del held, snapshot
    for t in range(120, 160):
        state, ts = env.step(state, policy(t, ts.observation.action_mask))
    made.clear()
    for t in range(160, 260):
        state, ts = env.step(state, policy(t, ts.observation.action_mask))
    assert len(made) == 0 and n_with_refs > 3, (len(made), n_with_refs)
    sess.drain()


@pytest.mark.parametrize("groups,B", [(2, 256), (3, 100)])
def test_grouped_step_loop_equals_one_group(rbg, torch_mod, groups, B):
    """ops.GroupedStepLoop: the batch stepped as contiguous groups on streams of their own gives
    exactly the buffers of one StepLoop (a group's step depends only on its own state)."""
    from routing_board_generation_amd import ops

    torch = torch_mod
    m, N, steps = 16, 10, 70
    keys, akeys = _keys(rbg, 61, B), _keys(rbg, 62, B)
    dev = keys.device
    out = []
    for g in (1, groups):
        st, ts = ops.alloc_state(B, m, N, dev), ops.alloc_timestep(B, m, N, dev)
        kw = dict(time_limit=11, refill_every=5, fuse_policy=True)
        loop = (ops.StepLoop(st, ts, akeys, **kw) if g == 1
                else ops.GroupedStepLoop(st, ts, akeys, groups=g, **kw))
        loop.reset(keys)
        for t in range(steps):
            loop.step(t)
        loop.drain()
        torch.cuda.synchronize()
        out.append(({k: v.clone() for k, v in st.items()}, {k: v.clone() for k, v in ts.items()}))
    for a, b in zip(out[0], out[1]):
        for k in a:
            assert torch.equal(a[k], b[k]), k
    assert int(out[0][0]["step_count"].max()) <= 11


def test_c4_global_batch_of_baseline_configs_3(rbg, oracle, torch_mod):
    """BASELINE configs[3]: ParallelRandomWalk 32x32 / 20 agents, batch 32768 sharded over 8 GPUs.
    On one GPU: the eight shards (sharding.shard_keys, the reference's reshape(devices,
    per_device, 2), setup_train.py:396-400) generated separately == ONE launch of the whole
    batch == the oracle, board for board."""
    from routing_board_generation_amd import ops, sharding

    torch = torch_mod
    m, N, B, G = 32, 20, 32768, 8
    keys = _keys(rbg, 0, B)  # split(PRNGKey(0), 32768)
    whole = ops.prw_generate(keys, m, N, split_first=True, want_solved=True, want_state=True,
                             want_iters=True)
    names = ("solved_grid", "train_grid", "start", "target", "position", "step_count", "key_out",
             "iters")
    for g in range(G):
        lo, hi = sharding.shard_bounds(B, g, G)
        assert hi - lo == 4096
        part = ops.prw_generate(sharding.shard_keys(keys, g, G), m, N, split_first=True,
                                want_solved=True, want_state=True, want_iters=True)
        for k in names:
            assert torch.equal(part[k], whole[k][lo:hi]), (g, k)
    est, esolved, eiters = oracle.generate_batch(to_np(keys), m, N, threads=16)
    assert (to_np(whole["solved_grid"]) == esolved).all()
    assert (to_np(whole["iters"]) == eiters).all()
    got = {"grid": to_np(whole["train_grid"]), "start": to_np(whole["start"]),
           "target": to_np(whole["target"]), "position": to_np(whole["position"]),
           "key": to_np(whole["key_out"]), "step_count": to_np(whole["step_count"])}
    assert_dicts_equal(got, est, "C4 global batch: ")
    check_solved_boards(esolved[:2048], est["start"][:2048], est["target"][:2048])
