#!/usr/bin/env python3
"""Developer probe: env.step over one batch as G independent groups of envs on G HIP streams.
A step launch of group g at step t + 1 depends only on group g's launch at step t, so the launch
overhead / ramp / write tail of one group can overlap the streaming of the others."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import routing_board_generation_amd as rbg  # noqa: E402
from routing_board_generation_amd import ops  # noqa: E402

m, N, B, steps = 16, 10, 4096, 600
dev = torch.device("cuda", 0)
keys = rbg.split(rbg.PRNGKey(0), B)


def run(groups, fuse, refill_every=16):
    per = B // groups
    streams = [torch.cuda.Stream(dev) for _ in range(groups)] if groups > 1 else [torch.cuda.current_stream(dev)]
    loops = []
    for g, s in enumerate(streams):
        with torch.cuda.stream(s):
            st, ts = ops.alloc_state(per, m, N, dev), ops.alloc_timestep(per, m, N, dev)
            k = keys[g * per:(g + 1) * per].contiguous()
            loop = ops.StepLoop(st, ts, k, refill_every=refill_every, fuse_policy=fuse)
            loop.reset(k)
            loops.append(loop)
    for t in range(40):
        for loop in loops:
            loop.step(t)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for t in range(40, 40 + steps):
        for loop in loops:
            loop.step(t)
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    for loop in loops:
        loop.drain()
    torch.cuda.synchronize(dev)
    return el / steps * 1e6


for fuse in (True, False):
    for groups in (1, 2, 4):
        us = run(groups, fuse)
        print(f"fuse_policy={fuse} groups={groups}: {us:7.2f} us per step of {B} envs = {B / us:7.1f} M env-steps/s",
              flush=True)
