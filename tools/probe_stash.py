#!/usr/bin/env python3
"""Developer probe: per-step time of the fused-policy step loop against the refill period K and
the stash depth (1: one next-episode slot per env; 2: two, requested an episode ahead)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import routing_board_generation_amd as rbg  # noqa: E402
from routing_board_generation_amd import ops  # noqa: E402

B, m, N, n = 4096, 16, 10, 600
dev = torch.device("cuda", 0)
keys = rbg.split(rbg.PRNGKey(0), B)
for fuse in (True, False):
    for depth in (1, 2):
        for K in (8, 16, 24, 32, 48, 64):
            st, ts = ops.alloc_state(B, m, N, dev), ops.alloc_timestep(B, m, N, dev)
            loop = ops.StepLoop(st, ts, keys, refill_every=K, fuse_policy=fuse, stash_depth=depth)
            loop.reset(keys)
            for t in range(64):
                loop.step(t)
            loop.stash["hits"].zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for t in range(64, 64 + n):
                loop.step(t)
            e1.record()
            torch.cuda.synchronize()
            loop.drain()
            episodes = int((ts["step_type"] == 2).sum())  # (last step only; hits tell the rest)
            print(f"fuse={fuse} depth={depth} K={K:3d}: {e0.elapsed_time(e1) / n * 1e3:6.2f} us/step  "
                  f"stash hits {int(loop.stash['hits'].sum())}", flush=True)
