#!/usr/bin/env python3
"""Measure every BASELINE.json config on one MI355X (developer tool; the driver uses bench.py).

    python tools/bench_configs.py > gpurun_out/configs.json
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import routing_board_generation_amd as rbg  # noqa: E402
from routing_board_generation_amd import ops  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us per launch


def cpu_rate(fn, units):
    t0 = time.perf_counter()
    fn()
    return units / (time.perf_counter() - t0)


def main():
    from oracle import oracle as O

    from bench import host_cpu_share

    threads = host_cpu_share()
    rows = []

    def gen_row(name, m, N, B, kind="prw"):
        keys = rbg.split(rbg.PRNGKey(0), B)
        if kind == "prw":
            f = lambda: ops.prw_generate(keys, m, N, split_first=True, want_solved=False, want_state=True)  # noqa
            cf = lambda: O.generate_batch(keys.cpu().numpy(), m, N, threads=threads, want_solved=False)  # noqa
        else:
            f = lambda: ops.seq_rw_generate(keys, m, N, split_first=True, want_solved=False)  # noqa
            cf = lambda: O.seq_rw_generator_batch(keys.cpu().numpy(), m, N, threads=threads)  # noqa
        us = timed(f, 20 if B >= 1024 else 50)
        cf()
        rows.append({"config": name, "grid": m, "agents": N, "batch": B, "us_per_launch": us,
                     "boards_per_s": B / us * 1e6, "cpu_port_boards_per_s": cpu_rate(cf, B),
                     "cpu_threads": threads})

    gen_row("C1 ParallelRandomWalk 8x8/5 B=1", 8, 5, 1)
    gen_row("C2 ParallelRandomWalk 16x16/10 B=4096", 16, 10, 4096)
    gen_row("C4 ParallelRandomWalk 32x32/20 B=4096 (one GPU's shard of 32768)", 32, 20, 4096)
    gen_row("C5 SequentialRandomWalk 16x16/10 B=4096", 16, 10, 4096, kind="seq")
    gen_row("ref ParallelRandomWalk 10x10/5 B=100 (notebook batch: 19.7 ms on JAX-CPU)", 10, 5, 100)

    # C3: env.step rollout, per-step API and fused
    B, m, N = 4096, 16, 10
    keys = rbg.split(rbg.PRNGKey(0), B)
    st, ts = ops.connector_reset(keys, m, N)
    act = torch.empty((B, N), dtype=torch.int32, device="cuda")
    t = [0]

    def step():
        ops.random_actions(keys, t[0], ts["action_mask"], out=act)
        ops.connector_step(st, act, auto_reset=True, state_out=st, timestep=ts)
        t[0] += 1

    us = timed(step, 100)
    rows.append({"config": "C3 Connector.step rollout 16x16/10 B=4096 (per-step API, obs written)",
                 "us_per_step": us, "env_steps_per_s": B / us * 1e6})
    stats = {k: torch.empty((B,), dtype=torch.int32, device="cuda") for k in ops.ROLLOUT_STATS}
    us = timed(lambda: ops.connector_rollout(st, keys, 0, 50, state_out=st, stats=stats), 10)
    rows.append({"config": "C3 Connector.step rollout 16x16/10 B=4096 (fused rbg_connector_rollout, 50 steps/launch)",
                 "us_per_launch": us, "env_steps_per_s": B * 50 / us * 1e6})
    est, _ = O.connector_reset_batch(keys.cpu().numpy(), m, N, threads=threads)
    mask = O.connector_observe_batch(est, threads=threads)["action_mask"]

    def cpu_steps():
        m_ = mask
        for s in range(10):
            a = O.random_actions_batch(keys.cpu().numpy(), s, m_)
            m_ = O.connector_step_batch(est, a, auto_reset=True, threads=threads)["action_mask"]

    rows.append({"config": "C3 CPU port (oracle, per-step)", "cpu_threads": threads,
                 "env_steps_per_s": cpu_rate(cpu_steps, B * 10)})
    print(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
