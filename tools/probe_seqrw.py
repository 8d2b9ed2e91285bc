#!/usr/bin/env python3
"""Developer probe: is the SequentialRandomWalk launch bounded by its slowest board?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import routing_board_generation_amd as rbg  # noqa: E402
from routing_board_generation_amd import ops  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


m, N, B = 16, 10, 4096
keys = rbg.split(rbg.PRNGKey(0), B)
out = ops.seq_rw_generate(keys, m, N, split_first=True, want_solved=False)
att = out["attempts"].cpu()
order = torch.argsort(att)
print("attempts histogram", torch.bincount(att).tolist())
print(f"all 4096 boards: {timed(lambda: ops.seq_rw_generate(keys, m, N, True, want_solved=False)):8.1f} us")
for name, idx in (("worst board alone", order[-1:]), ("2nd worst alone", order[-2:-1]),
                  ("median board alone", order[2048:2049]), ("the 3234 one-attempt boards", order[:3234]),
                  ("all but the 16 worst", order[:-16]), ("all but the 64 worst", order[:-64])):
    k = keys.view(torch.int32)[idx.to(keys.device)].contiguous().view(torch.uint32)
    us = timed(lambda: ops.seq_rw_generate(k, m, N, True, want_solved=False))
    print(f"{name:32s} B={k.shape[0]:5d}: {us:8.1f} us  (max attempts {int(att[idx].max())})")
