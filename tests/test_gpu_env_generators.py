"""GPU parity of the coupled env path for EVERY generator the reference can put behind Connector
(routing_board_generation/rl_training/setup_train.py:112-161: offline_seed_extension,
offline_parallel_rw, online_uniform, online_seq_rw, online_random_seed, online_parallel_rw).

The oracle side is composed from pieces that are each pinned or compared elsewhere:
  connector_step_batch(auto_reset=False)  +  generator batch call on split(state.key)[0] for the
  envs whose timestep is LAST  +  connector_observe_batch of the fresh states
which is [3P] jumanji.wrappers.VmapAutoResetWrapper._auto_reset written out (only the observation
of the terminal timestep is replaced).  State and the full timestep are compared after every step.

Parity status: the env semantics are "parity unpinned" (DESIGN.md section 2): these tests compare
the HIP path with this project's oracle.
"""
import numpy as np
import pytest

from helpers import assert_dicts_equal, state_to_np, timestep_to_np, to_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rbg():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (there is no CPU fallback to hide behind)")
    import routing_board_generation_amd as pkg

    pkg._native.lib()
    return pkg


def _keys(rbg, seed, B):
    return rbg.split(rbg.PRNGKey(seed), B)


GENERATORS = ["online_parallel_rw", "online_uniform", "online_seq_rw", "online_random_seed",
              "offline_parallel_rw", "offline_seed_extension",
              # not a generation_type of the reference: the opt-in Philox stream of the walk, whose
              # State.key is not a split (its own rule in the stash protocol); checked against
              # this project's restatement of the stream only (oracle/prw_philox.c)
              "parallel_rw_philox"]


def make_pair(rbg, oracle, name, m, N, n_boards=64):
    """(package generator, oracle batch function keys [B, 2] -> state dict)."""
    from routing_board_generation_amd.interface.board_generator_interface import make_online_generator

    if name == "parallel_rw_philox":
        return (rbg.ParallelRandomWalkPhiloxGenerator(m, N),
                lambda keys: oracle.prw_philox_batch(keys, m, N, threads=8)[0])
    if name.startswith("offline"):
        # cfg.env.seed_extension.* of agent_training/configs/env/connector.yaml go to the dataset
        gen = make_online_generator(name, m, N, randomness=0.0, two_sided=True,
                                    extension_iterations=1, extension_steps=1e23,
                                    number_of_boards=n_boards)
        heads, targets = to_np(gen._heads), to_np(gen._targets)
        return gen, lambda keys: oracle.dataset_sample_batch(keys, heads, targets, m)[0]
    gen = make_online_generator(name, m, N)
    if name == "online_parallel_rw":
        return gen, lambda keys: oracle.generate_batch(keys, m, N, "parallel_random_walk",
                                                       threads=8, want_solved=False)[0]
    if name == "online_uniform":
        return gen, lambda keys: oracle.generate_batch(keys, m, N, "uniform", threads=8,
                                                       want_solved=False)[0]
    if name == "online_seq_rw":
        return gen, lambda keys: oracle.seq_rw_generator_batch(keys, m, N, threads=8)[0]
    if name == "online_random_seed":
        return gen, lambda keys: oracle.seed_extension_batch(keys, m, N, split_first=True,
                                                             threads=8)[0]
    raise ValueError(name)


def oracle_reset(oracle, gen_fn, keys):
    st = gen_fn(keys)
    return st, oracle.connector_observe_batch(st, threads=8)


def oracle_auto_reset_step(oracle, gen_fn, est, eaction, tl):
    """VmapAutoResetWrapper(Connector(g)).step on the oracle side; steps `est` in place."""
    ets = oracle.connector_step_batch(est, eaction, time_limit=tl, auto_reset=False, threads=8)
    idx = np.nonzero(ets["step_type"] == 2)[0]
    if idx.size:
        new_keys = np.stack([oracle.split(k)[0] for k in est["key"][idx]])
        fresh = gen_fn(new_keys)
        fts = oracle.connector_observe_batch(fresh, threads=8)
        for f in est:
            est[f][idx] = fresh[f]
        for f in ("obs_grid", "action_mask", "obs_step_count"):
            ets[f][idx] = fts[f]
    return ets


def aggregate(ets, N):
    """MultiToSingleWrapper: reward ((0 + r_0) + r_1) + ... in float32, discount max."""
    out = dict(ets)
    total = np.zeros(ets["reward"].shape[:1], dtype=np.float32)
    for i in range(N):
        total = (total + ets["reward"][:, i]).astype(np.float32)
    out["reward"] = total
    out["discount"] = ets["discount"].max(axis=1)
    return out


@pytest.mark.parametrize("prefetch,delay,depth", [(False, 0, 0), (True, 1, 0), (True, 5, 0),
                                                  (True, 7, 3), (True, 9, 4)])
@pytest.mark.parametrize("name", GENERATORS)
def test_auto_reset_rollout_matches_oracle(rbg, oracle, name, prefetch, delay, depth):
    """The reference's training env VmapAutoResetWrapper(MultiToSingleWrapper(Connector(g)))
    (setup_train.py:158-167) for every generation_type, several episodes at 16x16/10 under random
    legal actions: state and full timestep after every step."""
    from routing_board_generation_amd import ops

    m, N, B, steps, tl = 16, 10, 96, 40, 9
    gen, gen_fn = make_pair(rbg, oracle, name, m, N)
    env = rbg.VmapAutoResetWrapper(rbg.MultiToSingleWrapper(
        rbg.Connector(generator=gen, time_limit=tl)), prefetch=prefetch, prefetch_delay=delay)
    if depth:  # slots per env of the next-episode stash (default: the generator's own choice)
        env.unwrapped._gen.refill["depth"] = depth
    keys, akeys = _keys(rbg, 31, B), _keys(rbg, 32, B)
    state, ts = env.reset(keys)
    est, ets = oracle_reset(oracle, gen_fn, to_np(keys))
    assert_dicts_equal(state_to_np(state), est, "reset state.")
    for f in ("obs_grid", "action_mask", "obs_step_count", "num_connections", "total_path_length"):
        assert (timestep_to_np(ts)[f] == ets[f]).all(), f
    assert ts.reward.shape == (B,) and (to_np(ts.reward) == 0).all()
    n_last = 0
    for t in range(steps):
        action = ops.random_actions(akeys, t, ts.observation.action_mask)
        eaction = oracle.random_actions_batch(to_np(akeys), t, ets["action_mask"])
        assert (to_np(action) == eaction).all(), f"policy diverged at step {t}"
        state, ts = env.step(state, action)
        ets = oracle_auto_reset_step(oracle, gen_fn, est, eaction, tl)
        assert_dicts_equal(state_to_np(state), est, f"{name} step {t} state.")
        assert_dicts_equal(timestep_to_np(ts), aggregate(ets, N), f"{name} step {t} timestep.")
        n_last += int(to_np(ts.last()).sum())
    assert n_last >= 3 * B, "several episodes per env were expected"
    for sess in env.unwrapped._sessions.values():
        if prefetch:
            # the generator's own refill producer served the restarts (an entry is usable at most
            # 2 * delay launches after its request; episodes last up to 9 steps)
            hits = int(to_np(sess.epochs.stash["hits"]).sum())
            assert sess.epochs.stash["depth"] == (depth or sess.epochs.stash["depth"])
            # (with 3-4 slots entries are requested 2-3 episodes ahead: nearly every reset hits)
            assert hits >= (0.8 if delay == 1 or depth else 0.3) * n_last, (hits, n_last)
        sess.drain()


@pytest.mark.parametrize("name", GENERATORS)
def test_native_generators_take_the_fused_path(rbg, name):
    """Every generator of setup_train.py:112-152 steps through ONE launch per step: the Connector
    must not fall back to regenerate-every-env-every-step (Connector._generic_auto_reset)."""
    import torch

    from routing_board_generation_amd.interface.board_generator_interface import make_online_generator

    kw = dict(randomness=0.0, two_sided=True, number_of_boards=16) if name.startswith("offline") else {}
    gen = (rbg.ParallelRandomWalkPhiloxGenerator(8, 3) if name == "parallel_rw_philox"
           else make_online_generator(name, 8, 3, **kw))
    env = rbg.VmapAutoResetWrapper(rbg.Connector(generator=gen, time_limit=2))

    def refuse(*a, **k):
        raise AssertionError("generic auto-reset path taken")

    env.unwrapped._generic_auto_reset = refuse
    keys = _keys(rbg, 3, 16)
    state, ts = env.reset(keys)
    for _ in range(5):
        state, ts = env.step(state, torch.zeros((16, 3), dtype=torch.int32, device="cuda"))
    assert int(to_np(state.step_count).max()) <= 2
    # a single key keeps the unbatched shapes (jumanji's Connector.reset) and the same board
    one_state, one_ts = env.unwrapped.reset(keys[5])
    b_state, _ = env.unwrapped.reset(keys)
    assert one_state.grid.shape == (8, 8) and one_ts.observation.grid.shape == (3, 8, 8)
    assert (to_np(one_state.grid) == to_np(b_state.grid[5])).all()
    assert (to_np(one_state.key) == to_np(b_state.key[5])).all()


class _PythonGenerator:
    """A user-defined generator (no native producer): the documented fallback path."""

    def __init__(self, inner):
        self._inner = inner
        self.grid_size, self.num_agents = inner.grid_size, inner.num_agents

    def __call__(self, key):
        return self._inner(key)


def test_user_defined_generator_still_works_through_the_generic_path(rbg, oracle):
    from routing_board_generation_amd import ops

    m, N, B, tl = 8, 4, 32, 4
    gen = _PythonGenerator(rbg.UniformRandomGenerator(m, N))
    gen_fn = lambda keys: oracle.generate_batch(keys, m, N, "uniform", want_solved=False)[0]
    env = rbg.VmapAutoResetWrapper(rbg.Connector(generator=gen, time_limit=tl))
    keys, akeys = _keys(rbg, 6, B), _keys(rbg, 7, B)
    state, ts = env.reset(keys)
    est, ets = oracle_reset(oracle, gen_fn, to_np(keys))
    for t in range(11):
        action = ops.random_actions(akeys, t, ts.observation.action_mask)
        eaction = oracle.random_actions_batch(to_np(akeys), t, ets["action_mask"])
        state, ts = env.step(state, action)
        ets = oracle_auto_reset_step(oracle, gen_fn, est, eaction, tl)
        assert_dicts_equal(state_to_np(state), est, f"step {t} state.")
        assert_dicts_equal(timestep_to_np(ts), ets, f"step {t} timestep.")


@pytest.mark.parametrize("record", [False, True])
@pytest.mark.parametrize("prefetch", [False, True])
@pytest.mark.parametrize("name", GENERATORS)
def test_fused_random_rollout_for_every_generator(rbg, oracle, name, prefetch, record):
    """benchmark_on_random_agent.py:59-102 runs on whatever generator the config names: the fused
    (and the recording) rollout launch against the oracle's step-by-step loop."""
    m, N, B, tl, n_steps = 16, 10, 64, 7, 12
    gen, gen_fn = make_pair(rbg, oracle, name, m, N)
    env = rbg.VmapAutoResetWrapper(rbg.Connector(generator=gen, time_limit=tl),
                                   prefetch=prefetch, prefetch_delay=1)
    if prefetch:  # the stash path for every generator, whatever its measured preference
        env.unwrapped._gen.refill.update(rollout_stash=True, record_stash=True)
    keys, akeys = _keys(rbg, 41, B), _keys(rbg, 42, B)
    state, ts = env.reset(keys)
    est, ets = oracle_reset(oracle, gen_fn, to_np(keys))
    t0 = 0
    for launch in range(3):
        if record:
            state, stats, actions, traj = env.random_rollout_record(state, akeys, t0, n_steps)
            tnp = {k: v for k, v in timestep_to_np(traj).items()}
        else:
            state, stats, last = env.random_rollout(state, akeys, t0, n_steps, want_timestep=True)
        episodes = np.zeros((B,), np.int32)
        for s in range(n_steps):
            eaction = oracle.random_actions_batch(to_np(akeys), t0 + s, ets["action_mask"])
            ets = oracle_auto_reset_step(oracle, gen_fn, est, eaction, tl)
            episodes += (ets["step_type"] == 2).astype(np.int32)
            if record:
                assert (to_np(actions[s]) == eaction).all(), (launch, s)
                assert_dicts_equal({k: v[s] for k, v in tnp.items()}, ets,
                                   f"{name} launch {launch} step {s} trajectory.")
        assert_dicts_equal(state_to_np(state), est, f"{name} launch {launch} state.")
        assert (to_np(stats["episodes"]) == episodes).all()  # (the recording launch counts too)
        if not record:
            assert_dicts_equal(timestep_to_np(last), ets, f"{name} launch {launch} last timestep.")
        t0 += n_steps


def test_adaptive_refill_period_follows_short_episodes(rbg, oracle):
    """The generator's default stash schedule is sized for episodes of about the time limit; when
    episodes are much shorter than the schedule assumes (here 4 steps against a period of 32)
    resets miss the stash and run the generator inside the step launch.  The session watches
    the miss counter and refills more often -- without any effect on the results (compared with
    the oracle throughout)."""
    from routing_board_generation_amd import ops

    m, N, B, tl, steps = 10, 4, 48, 4, 260
    gen, gen_fn = make_pair(rbg, oracle, "online_seq_rw", m, N)
    # (the generator's rolling schedule with its period for 50-step episodes given explicitly:
    # left to itself the env would shorten the period to fit a time limit of 4)
    env = rbg.VmapAutoResetWrapper(rbg.Connector(generator=gen, time_limit=tl), prefetch_delay=32)
    keys, akeys = _keys(rbg, 71, B), _keys(rbg, 72, B)
    state, ts = env.reset(keys)
    est, ets = oracle_reset(oracle, gen_fn, to_np(keys))
    (sess,) = env.unwrapped._sessions.values()
    assert not sess.epochs.serial and sess.epochs.K == 32  # the rolling default
    periods = []
    for t in range(steps):
        action = ops.random_actions(akeys, t, ts.observation.action_mask)
        eaction = oracle.random_actions_batch(to_np(akeys), t, ets["action_mask"])
        state, ts = env.step(state, action)
        ets = oracle_auto_reset_step(oracle, gen_fn, est, eaction, tl)
        assert_dicts_equal(state_to_np(state), est, f"step {t} state.")
        assert_dicts_equal(timestep_to_np(ts), ets, f"step {t} timestep.")
        periods.append(sess.epochs.K)
    assert sess.epochs.fell_back and sess.epochs.serial and sess.epochs.adaptive
    # The designed end point is 16, not lower: eight slots per env ask 8 x 4 = 32 steps ahead, and
    # in line a request is served at most K steps later, so the misses per refill stop as soon as
    # K <= 16 and the rule (halve WHILE misses keep arriving, schedule.RefillSchedule) has no reason
    # to go further.  Whether a run also visits 8 depends on when the asynchronous miss snapshots
    # land -- one more halving happens if a storm's last snapshot arrives after K reached 16 (round
    # 3's first version of this test saw 8 on one box and was written as <= 6, which no rule
    # promises) -- so 16 is asserted, together with the rule's floor (K0 / 8 = 4 after the
    # fallback's K0 = 32).
    assert 4 <= min(periods) <= 16, periods[::20]
    misses = int(to_np(sess.epochs.stash["misses"]).sum())
    hits = int(to_np(sess.epochs.stash["hits"]).sum())
    assert misses > 0 and hits > misses, (hits, misses)
    sess.drain()


SHAPES = [(5, 2, 9), (7, 4, 33), (12, 33, 16), (32, 20, 12), (40, 15, 8)]


@pytest.mark.parametrize("m,N,B", SHAPES)
@pytest.mark.parametrize("name", GENERATORS)
def test_env_generators_at_other_shapes(rbg, oracle, name, m, N, B):
    """Odd and large grids, many agents (N = 33: full-wave boards everywhere; 40x40 SeedExtension:
    its LDS area forces fewer waves per workgroup in the env kernels): reset, auto-resetting steps
    with and without the stash, one recording rollout launch -- all against the oracle."""
    from routing_board_generation_amd import ops

    tl = 4
    gen, gen_fn = make_pair(rbg, oracle, name, m, N, n_boards=24)
    keys, akeys = _keys(rbg, 81, B), _keys(rbg, 82, B)
    for prefetch in (False, True):
        env = rbg.VmapAutoResetWrapper(rbg.Connector(generator=gen, time_limit=tl),
                                       prefetch=prefetch, prefetch_delay=2)
        if prefetch:
            env.unwrapped._gen.refill.update(rollout_stash=True, record_stash=True)
        state, ts = env.reset(keys)
        est, ets = oracle_reset(oracle, gen_fn, to_np(keys))
        assert_dicts_equal(state_to_np(state), est, f"{name} {m}x{m}/{N} reset state.")
        for t in range(11):
            action = ops.random_actions(akeys, t, ts.observation.action_mask)
            eaction = oracle.random_actions_batch(to_np(akeys), t, ets["action_mask"])
            state, ts = env.step(state, action)
            ets = oracle_auto_reset_step(oracle, gen_fn, est, eaction, tl)
            assert_dicts_equal(state_to_np(state), est, f"{name} {m}x{m}/{N} step {t} state.")
            assert_dicts_equal(timestep_to_np(ts), ets, f"{name} {m}x{m}/{N} step {t} timestep.")
        state, stats, actions, traj = env.random_rollout_record(state, akeys, 11, 6)
        tnp = timestep_to_np(traj)
        for s in range(6):
            eaction = oracle.random_actions_batch(to_np(akeys), 11 + s, ets["action_mask"])
            ets = oracle_auto_reset_step(oracle, gen_fn, est, eaction, tl)
            assert (to_np(actions[s]) == eaction).all()
            assert_dicts_equal({k: v[s] for k, v in tnp.items()}, ets,
                               f"{name} {m}x{m}/{N} rollout step {s}.")
        assert_dicts_equal(state_to_np(state), est, f"{name} {m}x{m}/{N} rollout state.")
        for sess in env.unwrapped._sessions.values():
            sess.drain()


@pytest.mark.parametrize("name", ["online_random_seed", "online_uniform"])
def test_donated_stepping_for_other_generators(rbg, oracle, name):
    """Connector(donate=True) -- the State advanced in place, two alternating TimeStep buffer
    sets -- with a generator other than the walk behind it."""
    from routing_board_generation_amd import ops

    m, N, B, tl = 16, 10, 80, 6
    gen, gen_fn = make_pair(rbg, oracle, name, m, N)
    env = rbg.VmapAutoResetWrapper(rbg.Connector(generator=gen, time_limit=tl, donate=True),
                                   prefetch=True, prefetch_delay=3)
    keys, akeys = _keys(rbg, 91, B), _keys(rbg, 92, B)
    state, ts = env.reset(keys)
    est, ets = oracle_reset(oracle, gen_fn, to_np(keys))
    grid_ptr = state.grid.data_ptr()
    for t in range(25):
        action = ops.random_actions(akeys, t, ts.observation.action_mask)
        eaction = oracle.random_actions_batch(to_np(akeys), t, ets["action_mask"])
        state, ts = env.step(state, action)
        ets = oracle_auto_reset_step(oracle, gen_fn, est, eaction, tl)
        assert state.grid.data_ptr() == grid_ptr  # in place
        assert_dicts_equal(state_to_np(state), est, f"{name} step {t} state.")
        assert_dicts_equal(timestep_to_np(ts), ets, f"{name} step {t} timestep.")
    for sess in env.unwrapped._sessions.values():
        sess.drain()


def test_step_and_refill_of_another_generator_capture_into_a_graph(rbg, oracle):
    """The generator-agnostic calls keep the C-ABI's contract -- asynchronous on the caller's
    stream, no hidden allocation or synchronisation -- so a step (with the SeedExtension generator
    behind it) and its stash refill (a memset, the row-listing launch and the producer) capture
    into a hipGraph and replay."""
    import torch

    from routing_board_generation_amd import ops

    m, N, B, tl = 12, 6, 48, 3
    gen, gen_fn = make_pair(rbg, oracle, "online_random_seed", m, N)
    spec = gen.native_spec()
    keys = _keys(rbg, 95, B)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        st, ts = ops.alloc_state(B, m, N, keys.device), ops.alloc_timestep(B, m, N, keys.device)
        stash = ops.alloc_stash(B, m, N, keys.device, depth=2)
        ops.connector_reset_prefetch(keys, m, N, stash, 0, 1, state=st, timestep=ts, gen=spec)
        ops.stash_refill(stash, m, gen=spec)
        noop = torch.zeros((B, N), dtype=torch.int32, device=keys.device)
        # every launch of the graph is "tick 1 + refill": entries requested in it are usable in
        # the next replay (ready_delay 0 would be wrong: the refill follows the step)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            ops.connector_step_prefetch(st, noop, stash, 1, 0, time_limit=tl, state_out=st,
                                        timestep=ts, gen=spec)
            ops.stash_refill(stash, m, gen=spec)
    torch.cuda.current_stream().wait_stream(side)
    est, ets = oracle_reset(oracle, gen_fn, to_np(keys))
    for t in range(10):
        graph.replay()
        torch.cuda.synchronize()
        ets = oracle_auto_reset_step(oracle, gen_fn, est, to_np(noop), tl)
        got = {k: to_np(v) for k, v in st.items()}
        assert_dicts_equal(got, est, f"replay {t} state.")
        assert (to_np(ts["obs_grid"]) == ets["obs_grid"]).all()
    assert int(to_np(stash["hits"]).sum()) >= 2 * B  # three restarts per env, all but maybe the first from the stash


def test_benchmark_on_random_agent_example_runs():
    """examples/benchmark_on_random_agent.py: the reference's generator comparison under a random
    agent (BenchmarkData's fields per epoch) from the fused rollout's counters."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.check_output(
        [sys.executable, os.path.join(root, "examples", "benchmark_on_random_agent.py"),
         "--epochs", "2", "--batch", "256", "--n-steps", "60"], text=True, cwd=root, timeout=600)
    rows = [json.loads(ln) for ln in out.splitlines() if ln.startswith("{")]
    assert [r["generator_type"] for r in rows] == GENERATORS[:6]
    for r in rows:
        assert len(r["episode_length"]) == 2 and 1.0 <= r["average_steps_till_board_terminates"] <= 50.0
        assert 0.0 <= r["average_proportion_of_wires_connected"] <= 1.0
        assert r["average_total_wire_length"] >= 0.0 and r["average_reward_per_wire"] < 0.5


@pytest.mark.parametrize("depth,delay,tl", [(2, 2, 3), (3, 5, 3), (4, 11, 2), (4, 3, 5), (2, 13, 4)])
def test_stash_request_rule_under_hits_and_misses(rbg, oracle, depth, delay, tl):
    """Short episodes against long refill periods: resets are a mix of stash hits (from any of
    the 2-4 slots) and misses that run the generator inline, rows go stale and are re-requested
    -- the results must be the oracle's whatever the mix."""
    from routing_board_generation_amd import ops

    m, N, B, steps = 8, 4, 200, 90
    gen, gen_fn = make_pair(rbg, oracle, "online_uniform", m, N)
    env = rbg.VmapAutoResetWrapper(rbg.Connector(generator=gen, time_limit=tl),
                                   prefetch=True, prefetch_delay=delay)
    env.unwrapped._gen.refill["depth"] = depth
    keys, akeys = _keys(rbg, 55, B), _keys(rbg, 56, B)
    state, ts = env.reset(keys)
    est, ets = oracle_reset(oracle, gen_fn, to_np(keys))
    for t in range(steps):
        action = ops.random_actions(akeys, t, ts.observation.action_mask)
        eaction = oracle.random_actions_batch(to_np(akeys), t, ets["action_mask"])
        state, ts = env.step(state, action)
        ets = oracle_auto_reset_step(oracle, gen_fn, est, eaction, tl)
        assert_dicts_equal(state_to_np(state), est, f"d{depth} K{delay} step {t} state.")
        assert_dicts_equal(timestep_to_np(ts), ets, f"d{depth} K{delay} step {t} timestep.")
    (sess,) = env.unwrapped._sessions.values()
    hits = int(to_np(sess.epochs.stash["hits"]).sum())
    misses = int(to_np(sess.epochs.stash["misses"]).sum())
    assert hits > 0 and hits + misses >= B * (steps // tl - 1), (hits, misses)
    if delay > depth * tl:
        assert misses > 0, "this schedule was meant to miss"
    sess.drain()


@pytest.mark.parametrize("B,steps,tl", [(192, 1500, 50), (4096, 700, 50), (4096, 400, 20)])
@pytest.mark.parametrize("name", ["online_parallel_rw", "online_seq_rw", "online_random_seed",
                                  "offline_seed_extension"])
def test_long_runs_do_not_depend_on_the_stash_schedule(rbg, name, B, steps, tl):
    """1500 auto-resetting steps (about 40 episodes per env) through the drop-in classes, once
    with every restart running the generator inline and once with the generator's default
    next-episode-stash schedule (output chunks reused, adaptive period): the same final State,
    the same reward total, the same number of episodes.  The B = 4096 rows are the benchmark's
    batch: the refills that roll beside the stepping stream (SequentialRandomWalk, SeedExtension)
    then really overlap hundreds of step launches, and no host synchronisation paces them; with a
    time limit of 20 the env shortens the refill period so that the rolling form still fits."""
    import torch

    from routing_board_generation_amd import ops
    from routing_board_generation_amd.interface.board_generator_interface import make_online_generator

    m, N = 16, 10
    keys, akeys = _keys(rbg, 61, B), _keys(rbg, 62, B)
    results = []
    for prefetch in (False, True):
        kw = dict(randomness=0.0, two_sided=True, number_of_boards=512) if name.startswith("offline") else {}
        env = rbg.VmapAutoResetWrapper(rbg.MultiToSingleWrapper(
            rbg.Connector(generator=make_online_generator(name, m, N, **kw), time_limit=tl)),
            prefetch=prefetch)
        policy = ops.RandomPolicy(akeys, N)
        state, ts = env.reset(keys)
        total = torch.zeros((B,), dtype=torch.float64, device=keys.device)
        episodes = torch.zeros((), dtype=torch.int64, device=keys.device)
        for t in range(steps):
            state, ts = env.step(state, policy(t, ts.observation.action_mask))
            total += ts.reward
            episodes += ts.last().sum()
        torch.cuda.synchronize()
        results.append((state_to_np(state), to_np(total), int(episodes)))
        for sess in env.unwrapped._sessions.values():
            sess.drain()
            ep = sess.epochs
            if prefetch and B == 4096 and name in ("online_seq_rw", "online_random_seed"):
                # the rolling form held (no fallback to in-line refills) and served the resets
                assert ep is not None and not ep.serial and not ep.fell_back
                hits, misses = int(to_np(ep.stash["hits"]).sum()), int(to_np(ep.stash["misses"]).sum())
                assert hits > 0.98 * int(episodes) and misses < 0.02 * int(episodes), (hits, misses)
    assert_dicts_equal(results[0][0], results[1][0], f"{name} final state.")
    assert (results[0][1] == results[1][1]).all() and results[0][2] == results[1][2]
    assert results[0][2] > (steps // (tl + 10)) * B


@pytest.mark.parametrize("name", GENERATORS)
def test_policy_inside_the_step_launch_for_every_generator(rbg, name):
    """The masked-uniform random policy computed INSIDE the step launch (rbg_connector_step_gen with
    policy_keys: its own kernel instantiation for the plain stash form, a run-time branch for the
    deep one) gives exactly the trajectory of { rbg_random_actions; step with those actions } --
    every generator, its default stash schedule, buffers compared after every step."""
    import torch

    from routing_board_generation_amd import ops
    from routing_board_generation_amd.interface.board_generator_interface import make_online_generator

    m, N, B, tl, steps = 16, 10, 192, 9, 45
    kw = dict(randomness=0.0, two_sided=True, number_of_boards=40) if name.startswith("offline") else {}
    gen = (rbg.ParallelRandomWalkPhiloxGenerator(m, N) if name == "parallel_rw_philox"
           else make_online_generator(name, m, N, **kw)).native_spec()
    sched = {"every": 16, "depth": 1, "serial": True, "lead": 2, "streams": 2, **gen.refill}
    every = min(sched["every"], 6) if sched["every"] else 0  # (short episodes: refill often)
    keys, pkeys = _keys(rbg, 33, B), _keys(rbg, 34, B)
    loops = []
    for fuse in (False, True):
        st, ts = ops.alloc_state(B, m, N, "cuda"), ops.alloc_timestep(B, m, N, "cuda")
        loop = ops.StepLoop(st, ts, pkeys, time_limit=tl, refill_every=every, fuse_policy=fuse,
                            gen=gen, stash_depth=sched["depth"], serial_refill=sched["serial"],
                            refill_lead=sched["lead"], refill_streams=sched["streams"])
        loop.reset(keys)
        loops.append((loop, st, ts))
    for t in range(steps):
        for loop, _, _ in loops:
            loop.step(t)
        (_, sa, ta), (_, sb, tb) = loops
        for k in ("grid", "step_count", "start", "target", "position"):
            assert torch.equal(sa[k], sb[k]), (name, t, k)
        assert torch.equal(sa["key"].view(torch.int32), sb["key"].view(torch.int32)), (name, t)
        for k in ("obs_grid", "action_mask", "reward", "discount", "step_type"):
            assert torch.equal(ta[k], tb[k]), (name, t, k)
    for loop, _, _ in loops:
        loop.drain()
    assert int((loops[0][2]["step_type"] == 2).sum()) >= 0
