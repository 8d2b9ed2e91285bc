#!/usr/bin/env python3
"""Developer probe: duration of connector_step_kernel launches in a replayed rollout (recorded
actions, step launches only) for the auto-reset variants, at several batch sizes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import routing_board_generation_amd as rbg  # noqa: E402
from routing_board_generation_amd import ops  # noqa: E402

m, N, n = 16, 10, 320
dev = torch.device("cuda", 0)


def replay(B, refill_every, auto_reset, skip_refill_spans=True):
    keys = rbg.split(rbg.PRNGKey(0), B)
    rec = torch.empty((n, B, N), dtype=torch.int32, device=dev)
    for rp in (False, True):
        st, ts = ops.alloc_state(B, m, N, dev), ops.alloc_timestep(B, m, N, dev)
        loop = ops.StepLoop(st, ts, keys, refill_every=refill_every, auto_reset=auto_reset)
        loop.reset(keys)
        if not rp:
            for t in range(n):
                loop.step(t)
                rec[t].copy_(loop.action)
            loop.drain()
            continue
        spans = [[torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]]
        if refill_every:
            ep = loop._epochs
            queue_refill = ep.after_launch

            def after_launch(stream=None):
                due = ep.tick == ep._refill_after
                if due:
                    spans[-1][1].record()
                queue_refill(stream)
                if due:
                    spans.append([torch.cuda.Event(enable_timing=True),
                                  torch.cuda.Event(enable_timing=True)])
                    spans[-1][0].record()

            ep.after_launch = after_launch
        torch.cuda.synchronize()
        spans[-1][0].record()
        for t in range(n):
            loop.step(t, action=rec[t])
        spans[-1][1].record()
        torch.cuda.synchronize()
        loop.drain()
        return sum(a.elapsed_time(b) for a, b in spans) / n * 1e3


for B in (4096, 8192):
    print(f"B={B}: stash (K=16) {replay(B, 16, True):6.2f}  no auto-reset {replay(B, 0, False):6.2f}  "
          f"inline reset {replay(B, 0, True):6.2f} us per step launch", flush=True)
