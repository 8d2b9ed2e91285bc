#!/usr/bin/env python3
"""Developer probe: the walk's in-line stash refill (one launch per K steps, ~33 us, latency-bound)
overlapped with the ONE step launch that follows it, instead of queued in front of it.

  serial    step_t, refill, step_t+1, ...            rows usable from tick t + 1   (the default)
  overlap   step_t, [refill on a side stream || step_t+1], step_t+2 waits for the refill;
            rows usable from tick t + 2

Same trajectory either way (recorded actions); prints us per step of 4096 envs for both, several
repetitions, and checks the final States against each other."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import routing_board_generation_amd as rbg  # noqa: E402
from routing_board_generation_amd import _native as nat  # noqa: E402
from routing_board_generation_amd import ops  # noqa: E402

m, N, B = 16, 10, 4096


def run(lib, keys, rec, n, K, overlap, deep, dev):
    main = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(device=dev)
    st, ts = ops.alloc_state(B, m, N, dev), ops.alloc_timestep(B, m, N, dev)
    s, t = ops._state_struct(st, B, m, N), ops._timestep_struct(ts, B, m, N)
    bufs = ops.alloc_stash(B, m, N, dev, 1, rolling=deep)  # deep: in-flight marks -> priority form
    stash = ops._stash_struct(bufs, B, m, N)
    p = nat.EnvParams(50, -0.03, 0.1, 1)
    ms, ss = C.c_void_p(main.cuda_stream), C.c_void_p(side.cuda_stream)
    launched, done = torch.cuda.Event(), torch.cuda.Event()

    def ck(rc):
        assert rc == 0, (rc, lib.rbg_last_error())

    ck(lib.rbg_connector_reset_prefetch(keys.data_ptr(), B, m, N, C.byref(s), C.byref(t),
                                        C.byref(stash), 0, 1, ms))
    ck(lib.rbg_stash_refill(B, m, N, C.byref(stash), ms))
    tick, refill_after, pending = 1, K, False
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    e0.record()
    for i in range(n):
        if pending and tick == wait_at:
            main.wait_event(done)
            pending = False
        delay = refill_after + (2 if overlap else 1) - tick
        ck(lib.rbg_connector_step_prefetch(B, m, N, C.byref(s), C.byref(s), rec[i].data_ptr(),
                                           C.byref(p), C.byref(t), C.byref(stash), tick, delay, ms))
        if tick == refill_after:
            if overlap:
                launched.record(main)
                side.wait_event(launched)
                ck(lib.rbg_stash_refill(B, m, N, C.byref(stash), ss))
                done.record(side)
                pending, wait_at = True, tick + 2
            else:
                ck(lib.rbg_stash_refill(B, m, N, C.byref(stash), ms))
            refill_after = tick + K
        tick += 1
    e1.record()
    torch.cuda.synchronize(dev)
    main.wait_stream(side)
    return e0.elapsed_time(e1) / n * 1e3, st, int(bufs["hits"].sum()), int(bufs["misses"].sum())


def main():
    dev = torch.device("cuda", 0)
    lib = nat.lib()
    n, K = 600, 20
    keys = rbg.split(rbg.PRNGKey(0, device=dev), B)
    st0, _ = ops.connector_reset(keys, m, N)
    _, _, traj = ops.connector_rollout_record(st0, keys, 0, n, with_obs_grid=False)
    rec = traj["action"]
    ref = None
    for rep in range(3):
        for name, overlap, deep in (("serial", False, False), ("overlap", True, False),
                                    ("overlap_prio", True, True), ("serial_prio", False, True)):
            us, st, hits, misses = run(lib, keys, rec, n, K, overlap, deep, dev)
            if ref is None:
                ref = st
            same = all(torch.equal(st[f], ref[f]) for f in ops.STATE_FIELDS)
            print(json.dumps({"variant": name, "rep": rep, "K": K, "us_per_step": round(us, 3),
                              "hits": hits, "misses": misses, "final_state_equal": same}), flush=True)
            assert same


if __name__ == "__main__":
    main()
