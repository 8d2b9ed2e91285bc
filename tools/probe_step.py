#!/usr/bin/env python3
"""Developer probe: HBM rate of a pure Connector.step launch (no resets) vs batch size."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import routing_board_generation_amd as rbg  # noqa: E402
from routing_board_generation_amd import ops  # noqa: E402

sys.path.insert(0, ROOT)
from bench import env_step_bytes  # noqa: E402

m, N = 16, 10
for B in (4096, 16384, 65536):
    keys = rbg.split(rbg.PRNGKey(0), B)
    st, ts = ops.connector_reset(keys, m, N)
    act = torch.zeros((B, N), dtype=torch.int32, device="cuda")  # NOOP: nothing ever terminates
    so = ops.alloc_state(B, m, N, "cuda")
    for _ in range(3):
        ops.connector_step(st, act, time_limit=1 << 30, state_out=so, timestep=ts)
    reps = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        ops.connector_step(st, act, time_limit=1 << 30, state_out=so, timestep=ts)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    gb = B * env_step_bytes(m, N) / 1e9
    print(f"B={B:6d}  {us:8.1f} us/launch  {gb / (us * 1e-6) / 1e3:6.2f} TB/s algorithmic")
