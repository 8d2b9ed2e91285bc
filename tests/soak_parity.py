"""Randomised parity soak (a script, not collected by pytest): HIP path against the CPU oracle on
random grid sizes, agent counts, batch sizes, options and seeds, for every generator of the
reference's setup_train.py:112-152 (+ the opt-in Philox walk) and for the coupled env path
(per-step API with and without the next-episode stash, fused rollouts), until a time budget is
spent.  The fixed-shape sweeps of tests/test_gpu_*.py choose their shapes by hand; this one does
not, which is how a path nobody thought of gets exercised.

    python tests/soak_parity.py --minutes 10 --seed 1 --out gpurun_out/soak.jsonl

One JSON line per case (kind, parameters, boards / env-steps compared, ok) + a summary line;
exit code 1 if any case differed.  Every case is reproducible from its line (the seed of its
keys is in it).  Parity status as in DESIGN.md section 2: the generators are pinned by the
reference's goldens through the oracle, the env semantics are "parity unpinned".
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from helpers import assert_dicts_equal, state_to_np, timestep_to_np, to_np  # noqa: E402


ALL_KINDS = "prw,uniform,seq_rw,seed_ext,philox,dataset,env,rollout"


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=5.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default="")
    ap.add_argument("--kinds", default=ALL_KINDS)
    ap.add_argument("--max-boards", type=int, default=6000, help="largest generator batch")
    args = ap.parse_args()
    summary, failures = run(args)
    return 1 if failures else 0


def run(args, quiet: bool = False):
    """The soak itself: `args` carries minutes, seed, out, kinds, max_boards (the command line's,
    or tests/test_gpu_soak_slice.py's bounded slice).  Returns (summary dict, failed case lines)."""
    import torch

    import routing_board_generation_amd as rbg
    from oracle import oracle
    from routing_board_generation_amd import ops
    import test_gpu_env_generators as eg
    from test_gpu_seed_extension import _compare as compare_seed_extension

    assert torch.cuda.is_available(), "the soak needs a HIP device"
    rbg._native.lib()
    rnd = random.Random(args.seed)
    out = open(args.out, "a") if args.out else None
    deadline = time.time() + 60.0 * args.minutes
    totals = {}
    failures = []

    def keys_for(seed, B):
        return rbg.split(rbg.PRNGKey(seed), B)

    def shape(min_m=2, max_n=64, cells_per_agent=1):
        m = rnd.choice([rnd.randint(min_m, 12), rnd.randint(min_m, 24), rnd.randint(min_m, 40)])
        cap = max(1, min(max_n, (m * m) // cells_per_agent))
        N = rnd.choice([rnd.randint(1, min(cap, 8)), rnd.randint(1, min(cap, 24)),
                        rnd.randint(1, cap)])
        return m, N

    def batch(m, N, cost=1.0):
        # the oracle's time per board grows with the board; keep a case to a few seconds
        top = max(1, int(args.max_boards * 256.0 / (m * m * max(1.0, N / 8.0)) / cost))
        return rnd.choice([rnd.randint(1, 70), rnd.randint(1, max(1, min(top, 1000))),
                           rnd.randint(1, max(1, top))])

    # ------------------------------------------------------------------ generators
    def case_prw(seed):
        m, N = shape()
        B = batch(m, N)
        keys = keys_for(seed, B)
        got = ops.prw_generate(keys, m, N, split_first=True, want_solved=True, want_train=True,
                               want_state=True, want_iters=True)
        st, solved, iters = oracle.generate_batch(to_np(keys), m, N, threads=16)
        assert (to_np(got["solved_grid"]) == solved).all(), "solved grid"
        assert (to_np(got["iters"]) == iters).all(), "iteration counts"
        assert_dicts_equal({"grid": to_np(got["train_grid"]), "start": to_np(got["start"]),
                            "target": to_np(got["target"]), "key": to_np(got["key_out"])},
                           {k: st[k] for k in ("grid", "start", "target", "key")})
        return dict(m=m, N=N, B=B), B

    def case_uniform(seed):
        m, N = shape(cells_per_agent=2)
        B = batch(m, N)
        keys = keys_for(seed, B)
        state = rbg.UniformRandomGenerator(m, N)(keys)
        st = oracle.generate_batch(to_np(keys), m, N, "uniform", threads=16, want_solved=False)[0]
        assert_dicts_equal(state_to_np(state), st)
        return dict(m=m, N=N, B=B), B

    def case_seq_rw(seed):
        m, N = shape(min_m=3)
        B = batch(m, N, cost=4.0)
        keys = keys_for(seed, B)
        state = rbg.SequentialRandomWalkGenerator(m, N)(keys)
        st = oracle.seq_rw_generator_batch(to_np(keys), m, N, threads=16)[0]
        assert_dicts_equal(state_to_np(state), st)
        return dict(m=m, N=N, B=B), B

    def case_seed_ext(seed):
        m, N = shape()
        N = max(1, min(N, (m // 2) * (m // 2)))
        kw = {}
        if rnd.random() < 0.6:
            kw["randomness"] = rnd.choice([1.0, round(rnd.random(), 3)])
        if rnd.random() < 0.3:
            kw["two_sided"] = False
        if rnd.random() < 0.4:
            kw["extension_iterations"] = rnd.choice([0, 2, 2, 3, 5])
        if rnd.random() < 0.2:
            kw["extension_steps"] = rnd.choice([1, 2, 2.5, 7, 30])
        B = batch(m, N, cost=12.0 * max(1, kw.get("extension_iterations", 1)))
        keys = keys_for(seed, B)
        want_solved = rnd.random() < 0.7
        got = ops.seed_extension_generate(keys, m, N, split_first=True, want_solved=want_solved,
                                          **kw)
        est, esolved, _, esweeps = oracle.seed_extension_batch(to_np(keys), m, N, threads=16,
                                                               return_live=True, **kw)
        compare_seed_extension(got, est, esolved, esweeps, "seed extension")
        return dict(m=m, N=N, B=B, solved=want_solved, **kw), B

    def case_philox(seed):
        m, N = shape()
        B = batch(m, N)
        keys = keys_for(seed, B)
        state = rbg.ParallelRandomWalkPhiloxGenerator(m, N)(keys)
        st = oracle.prw_philox_batch(to_np(keys), m, N, threads=16)[0]
        assert_dicts_equal(state_to_np(state), st)
        return dict(m=m, N=N, B=B), B

    def case_dataset(seed):
        # the offline sampler: randint over a span that is usually not a power of two
        m, N = shape(min_m=4, max_n=16)
        name = rnd.choice(["offline_parallel_rw", "offline_seed_extension"])
        if name == "offline_seed_extension":
            N = max(1, min(N, (m // 2) * (m // 2)))
        n_boards = rnd.choice([1, 2, 3, 7, 64, 100, 255, 257, 1000])
        gen, gen_fn = eg.make_pair(rbg, oracle, name, m, N, n_boards=n_boards)
        B = rnd.randint(1, 4000)
        keys = keys_for(seed, B)
        assert_dicts_equal(state_to_np(gen(keys)), gen_fn(to_np(keys)))
        return dict(name=name, m=m, N=N, B=B, n_boards=n_boards), B

    # ------------------------------------------------------------------ the coupled env path
    def env_shape(name):
        if name == "online_uniform":
            m, N = shape(min_m=3, max_n=40, cells_per_agent=2)
        elif name in ("online_random_seed", "offline_seed_extension"):
            m, N = shape(min_m=4, max_n=40)
            N = max(1, min(N, (m // 2) * (m // 2)))
        else:
            m, N = shape(min_m=3, max_n=40)
        return m, N

    def case_env(seed):
        name = rnd.choice(eg.GENERATORS)
        m, N = env_shape(name)
        heavy = name in ("online_random_seed", "online_seq_rw")
        B = rnd.randint(1, 40 if heavy else 160)
        if m > 24:
            B = min(B, 24)
        tl = rnd.randint(1, 12)
        steps = rnd.randint(tl + 1, 3 * tl + 8)
        prefetch = rnd.random() < 0.8
        delay = rnd.choice([None, 1, rnd.randint(1, 12)])
        depth = rnd.choice([0, 0, 1, 2, 3, 5, 8])
        single = rnd.random() < 0.5
        gen, gen_fn = eg.make_pair(rbg, oracle, name, m, N)
        core = rbg.Connector(generator=gen, time_limit=tl)
        donate = rnd.random() < 0.3  # the State advances in place, TimeStep buffers alternate
        env = rbg.VmapAutoResetWrapper(rbg.MultiToSingleWrapper(core) if single else core,
                                       prefetch=prefetch, prefetch_delay=delay, donate=donate)
        if depth and env.unwrapped._gen is not None:
            env.unwrapped._gen.refill["depth"] = depth
        keys, akeys = keys_for(seed, B), keys_for(seed + 1, B)
        state, ts = env.reset(keys)
        est, ets = eg.oracle_reset(oracle, gen_fn, to_np(keys))
        assert_dicts_equal(state_to_np(state), est, "reset state.")
        for t in range(steps):
            action = ops.random_actions(akeys, t, ts.observation.action_mask)
            eaction = oracle.random_actions_batch(to_np(akeys), t, ets["action_mask"])
            assert (to_np(action) == eaction).all(), f"policy diverged at step {t}"
            state, ts = env.step(state, action)
            ets = eg.oracle_auto_reset_step(oracle, gen_fn, est, eaction, tl)
            assert_dicts_equal(state_to_np(state), est, f"step {t} state.")
            assert_dicts_equal(timestep_to_np(ts), eg.aggregate(ets, N) if single else ets,
                               f"step {t} timestep.")
        for sess in env.unwrapped._sessions.values():
            sess.drain()
        torch.cuda.synchronize()
        return dict(name=name, m=m, N=N, B=B, time_limit=tl, steps=steps, prefetch=prefetch,
                    delay=delay, depth=depth, multi_to_single=single, donate=donate), B * steps

    def case_rollout(seed):
        name = rnd.choice(eg.GENERATORS)
        m, N = env_shape(name)
        heavy = name in ("online_random_seed", "online_seq_rw")
        B = rnd.randint(1, 40 if heavy else 160)
        if m > 24:
            B = min(B, 24)
        tl = rnd.randint(1, 12)
        launches = rnd.randint(1, 3)
        record = rnd.random() < 0.5
        gen, gen_fn = eg.make_pair(rbg, oracle, name, m, N)
        env = rbg.VmapAutoResetWrapper(rbg.Connector(generator=gen, time_limit=tl),
                                       prefetch=rnd.random() < 0.8)
        keys, akeys = keys_for(seed, B), keys_for(seed + 1, B)
        state, ts = env.reset(keys)
        est, ets = eg.oracle_reset(oracle, gen_fn, to_np(keys))
        t0, total = 0, 0
        for _ in range(launches):
            n = rnd.randint(1, 2 * tl + 6)
            if record:
                state, stats, actions, traj = env.random_rollout_record(state, akeys, t0, n)
            else:
                state, stats, last = env.random_rollout(state, akeys, t0, n, want_timestep=True)
            episodes = np.zeros((B,), np.int64)
            for s in range(n):
                eaction = oracle.random_actions_batch(to_np(akeys), t0 + s, ets["action_mask"])
                ets = eg.oracle_auto_reset_step(oracle, gen_fn, est, eaction, tl)
                episodes += ets["step_type"] == 2
                if record:
                    assert (to_np(actions[s]) == eaction).all(), f"action of step {s}"
                    got = {k: to_np(v[s]) for k, v in (
                        ("step_type", traj.step_type), ("reward", traj.reward),
                        ("discount", traj.discount), ("obs_grid", traj.observation.grid),
                        ("obs_step_count", traj.observation.step_count),
                        ("num_connections", traj.extras["num_connections"]),
                        ("total_path_length", traj.extras["total_path_length"]))}
                    got["action_mask"] = to_np(traj.observation.action_mask[s]).astype(np.uint8)
                    assert_dicts_equal(got, {k: ets[k] for k in got}, f"recorded step {s} ")
            assert_dicts_equal(state_to_np(state), est, "state after the launch.")
            assert (to_np(stats["episodes"]) == episodes).all(), "episode counts"
            if not record:
                assert_dicts_equal(timestep_to_np(last), ets, "last timestep.")
            t0 += n
            total += n
        torch.cuda.synchronize()
        return dict(name=name, m=m, N=N, B=B, time_limit=tl, launches=launches, steps=total,
                    record=record), B * total

    cases = {"prw": case_prw, "uniform": case_uniform, "seq_rw": case_seq_rw,
             "seed_ext": case_seed_ext, "philox": case_philox, "dataset": case_dataset,
             "env": case_env, "rollout": case_rollout}
    kinds = [k for k in args.kinds.split(",") if k]
    n_cases, last_print = 0, time.time()
    while time.time() < deadline:
        kind = kinds[n_cases % len(kinds)]
        seed = rnd.randrange(1 << 31)
        line = {"kind": kind, "key_seed": seed}
        t_case = time.time()
        try:
            params, units = cases[kind](seed)
            line.update(params, units=units, ok=True)
            tot = totals.setdefault(kind, {"cases": 0, "units": 0})
            tot["cases"] += 1
            tot["units"] += units
        except AssertionError as e:
            line.update(ok=False, error=str(e)[:400], soak_seed=args.seed, case_index=n_cases)
            failures.append(line)
        except ValueError as e:  # a shape the library (or the oracle) refuses: not a parity matter
            line.update(ok=None, refused=str(e)[:200])
        line["seconds"] = round(time.time() - t_case, 2)
        if out:
            out.write(json.dumps(line) + "\n")
            out.flush()
        n_cases += 1
        if time.time() - last_print > 30:
            last_print = time.time()
            print(json.dumps({"progress": n_cases, "failures": len(failures), "totals": totals}),
                  flush=True)
    summary = {"summary": True, "soak_seed": args.seed, "minutes": args.minutes, "cases": n_cases,
               "failures": len(failures), "totals": totals}
    if not quiet:
        print(json.dumps(summary), flush=True)
        for f in failures[:20]:
            print(json.dumps(f), flush=True)
    if out:
        out.write(json.dumps(summary) + "\n")
        out.close()
    return summary, failures


if __name__ == "__main__":
    sys.exit(main())
