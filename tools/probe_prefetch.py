#!/usr/bin/env python3
"""Developer probe: per-step API env-steps/s with and without the next-episode stash."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import routing_board_generation_amd as rbg  # noqa: E402
from routing_board_generation_amd import ops  # noqa: E402

B, m, N, steps = 4096, 16, 10, 300
keys = rbg.split(rbg.PRNGKey(0), B)
act = torch.empty((B, N), dtype=torch.int32, device="cuda")


def run(prefetch, delay=4):
    if prefetch:
        pf = ops.Prefetcher(B, m, N, keys.device, delay=delay)
        tick = pf.before_launch()
        st, ts = ops.connector_reset_prefetch(keys, m, N, pf.stash, tick, pf.delay)
        pf.after_launch()
    else:
        st, ts = ops.connector_reset(keys, m, N)

    def step(t):
        ops.random_actions(keys, t, ts["action_mask"], out=act)
        if prefetch:
            tick = pf.before_launch()
            ops.connector_step_prefetch(st, act, pf.stash, tick, pf.delay, state_out=st, timestep=ts)
            pf.after_launch()
        else:
            ops.connector_step(st, act, auto_reset=True, state_out=st, timestep=ts)

    for t in range(60):
        step(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(60, 60 + steps):
        step(t)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    hits = int(pf.stash["hits"].sum().item()) if prefetch else 0
    print(f"prefetch={prefetch} delay={delay}: {dt / steps * 1e6:7.1f} us/step  "
          f"{B * steps / dt / 1e6:7.1f} M env-steps/s  stash hits {hits}")
    if prefetch:
        pf.drain()


run(False)
for d in (2, 4, 8):
    run(True, d)


def run_loop(delay, fuse=False):
    st, ts = ops.alloc_state(B, m, N, "cuda"), ops.alloc_timestep(B, m, N, "cuda")
    loop = ops.StepLoop(st, ts, keys, refill_every=delay, fuse_policy=fuse)
    loop.reset(keys)
    for t in range(60):
        loop.step(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(60, 60 + steps):
        loop.step(t)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    hits = int(loop.stash["hits"].sum().item()) if delay else 0
    print(f"StepLoop refill_every={delay} fuse_policy={fuse}: {dt / steps * 1e6:7.1f} us/step  "
          f"{B * steps / dt / 1e6:7.1f} M env-steps/s  stash hits {hits}")
    loop.drain()
    return st


ref = run_loop(0)
for d in (1, 2, 4, 6):
    got = run_loop(d)
    assert torch.equal(got["grid"], ref["grid"]) and torch.equal(got["position"], ref["position"])
for d in (0, 4, 6, 8):
    got = run_loop(d, fuse=True)
    assert torch.equal(got["grid"], ref["grid"]) and torch.equal(got["position"], ref["position"])
print("StepLoop states identical with and without prefetch / fused policy")
