#!/usr/bin/env python3
"""Developer probe: where does the HOST time of a functional env.step go?  (cProfile over the
drop-in wrapper loop; the GPU side is ~12-16 us per step at B = 4096.)"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import routing_board_generation_amd as rbg  # noqa: E402
from routing_board_generation_amd import ops  # noqa: E402

B, m, N = 4096, 16, 10
keys = rbg.split(rbg.PRNGKey(0), B)
env = rbg.VmapAutoResetWrapper(rbg.Connector(generator=rbg.ParallelRandomWalkGenerator(m, N)))
state, ts = env.reset(keys)
act = torch.zeros((B, N), dtype=torch.int32, device="cuda")


def loop(n):
    global state, ts
    for t in range(n):
        state, ts = env.step(state, act)


loop(200)
torch.cuda.synchronize()
t0 = time.perf_counter()
loop(2000)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"host issue time {t_issue / 2000 * 1e6:.1f} us/step, wall {(time.perf_counter() - t0) / 2000 * 1e6:.1f} us/step")
pr = cProfile.Profile()
pr.enable()
loop(2000)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
