#!/usr/bin/env python3
"""Developer probe: SeedExtension launch time against batch size and shape (GPU box)."""
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import routing_board_generation_amd as rbg  # noqa: E402
from routing_board_generation_amd import ops  # noqa: E402


def timed(fn, reps):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


rows = []
for (m, N, B) in ((16, 10, 256), (16, 10, 1024), (16, 10, 4096), (16, 10, 16384), (16, 10, 65536),
                  (10, 5, 100), (10, 5, 4096), (32, 20, 4096)):
    keys = rbg.split(rbg.PRNGKey(0), B)
    out = ops.seed_extension_generate(keys, m, N, split_first=True)
    sw = out["sweeps"].float()
    us = timed(lambda: ops.seed_extension_generate(keys, m, N, split_first=True), 5)
    us_solved = timed(lambda: ops.seed_extension_generate(keys, m, N, split_first=True,
                                                          want_solved=True, want_train=False), 5)
    rows.append({"m": m, "N": N, "B": B, "us": us, "us_with_bfs": us_solved,
                 "boards_per_s": B / us * 1e6, "sweeps_mean": sw.mean().item(),
                 "sweeps_max": sw.max().item(),
                 "us_per_sweep_of_longest": us / sw.max().item()})
    print(json.dumps(rows[-1]), flush=True)
