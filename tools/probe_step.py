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
for B in ((4096, 16384, 65536) if len(sys.argv) == 1 else ()):
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


def probe_times():
    """Per-wave timeline of one pure step launch at B = 4096 (probe build -DRBG_PROBE_STEP_TIMES:
    s_memrealtime at wave entry and after its last store, 10 ns ticks, one clock for the chip)."""
    import ctypes as C
    import subprocess

    import numpy as np

    from routing_board_generation_amd import _native as nat

    so = os.path.join(ROOT, "gpurun_out", "librbg_probe_step_times.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + nat.HIPCC_FLAGS +
                          ["-DRBG_PROBE_STEP_TIMES", "-o", so, nat.SRC_PATH])
    lib = C.CDLL(so)
    for name, (restype, argtypes) in nat.SIGNATURES.items():
        getattr(lib, name).restype, getattr(lib, name).argtypes = restype, argtypes
    for B in (4096, 16384):
        keys = rbg.split(rbg.PRNGKey(0), B)
        st, ts = ops.connector_reset(keys, m, N)
        act = torch.zeros((B, N), dtype=torch.int32, device="cuda")
        so_ = ops.alloc_state(B, m, N, "cuda")
        s_in, s_out = ops._state_struct(st, B, m, N), ops._state_struct(so_, B, m, N)
        t_ = ops._timestep_struct(ts, B, m, N)
        p = nat.EnvParams(1 << 30, -0.03, 0.1, 0)
        for rep in range(4):
            rc = lib.rbg_connector_step(B, m, N, C.byref(s_in), C.byref(s_out),
                                        C.c_void_p(act.data_ptr()), C.byref(p), C.byref(t_),
                                        C.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0, rc
            torch.cuda.synchronize()
        t0 = ts["num_connections"].cpu().numpy().astype(np.uint32)
        t1 = ts["total_path_length"].cpu().numpy().astype(np.uint32)
        hw = ts["obs_step_count"].cpu().numpy()
        base = t0.min()
        start = (t0 - base).astype(np.int64) * 10  # ns
        end = (t1 - base).astype(np.int64) * 10
        dur = end - start
        xcd, cu = hw >> 16, (hw >> 8) & 0xFF
        print(f"B={B}: launch span (first wave start -> last wave end) {end.max() / 1e3:.2f} us; "
              f"wave duration mean {dur.mean() / 1e3:.2f} p50 {np.median(dur) / 1e3:.2f} "
              f"p95 {np.percentile(dur, 95) / 1e3:.2f} max {dur.max() / 1e3:.2f} us")
        qs = [0, 10, 25, 50, 75, 90, 100]
        print("  start time percentiles (us): " +
              " ".join(f"p{q}={np.percentile(start, q) / 1e3:.2f}" for q in qs))
        print("  end time percentiles (us):   " +
              " ".join(f"p{q}={np.percentile(end, q) / 1e3:.2f}" for q in qs))
        # concurrency: waves in flight over time
        ev = np.concatenate([np.stack([start, np.ones_like(start)], 1),
                             np.stack([end, -np.ones_like(end)], 1)])
        ev = ev[np.argsort(ev[:, 0], kind="stable")]
        inflight = np.cumsum(ev[:, 1])
        grid = np.arange(0, end.max(), 500)
        idx = np.searchsorted(ev[:, 0], grid, side="right") - 1
        print("  waves in flight every 0.5 us: " + " ".join(str(int(inflight[i])) for i in idx))
        per_xcd = [int((xcd == x).sum()) for x in range(8)]
        print(f"  waves per XCD {per_xcd}; distinct (xcd, cu) pairs {len(set(zip(xcd, cu)))}")
        # by block index: when do late blocks start?
        blk = np.arange(B) // 4
        for lo in range(0, B // 4, B // 32):
            sel = (blk >= lo) & (blk < lo + B // 32)
            print(f"  blocks {lo:5d}..: start {start[sel].mean() / 1e3:6.2f}  "
                  f"dur {dur[sel].mean() / 1e3:5.2f}  end {end[sel].mean() / 1e3:6.2f} us")


if __name__ == "__main__" and "--times" in sys.argv:
    probe_times()


def probe_gaps():
    """Launch-interval floor of this stack: a trivial kernel back to back, the step kernel back to
    back through direct ctypes calls, and the same step launches replayed from a hipGraph."""
    import ctypes as C

    from routing_board_generation_amd import _native as nat

    lib = nat.lib()
    flags = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--flags=")]
    if flags:
        import subprocess
        so = os.path.join(ROOT, "gpurun_out", "librbg_probe_gaps.so")
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + nat.HIPCC_FLAGS +
                              flags[0].split() + ["-o", so, nat.SRC_PATH])
        lib = C.CDLL(so)
        for name, (restype, argtypes) in nat.SIGNATURES.items():
            getattr(lib, name).restype, getattr(lib, name).argtypes = restype, argtypes
        print(f"probe build with {flags[0]}")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = torch.empty((4, 2), dtype=torch.uint32, device="cuda")

    def span(fn, reps):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    print(f"trivial kernel (rbg_split, 1 block): {span(lambda: lib.rbg_split(1, 2, 4, C.c_void_p(out.data_ptr()), stream), 500):.2f} us per launch")
    for B in (2048, 4096, 4096, 8192):
        keys = rbg.split(rbg.PRNGKey(0), B)
        st, ts = ops.connector_reset(keys, m, N)
        act = torch.zeros((B, N), dtype=torch.int32, device="cuda")
        so_ = ops.alloc_state(B, m, N, "cuda")
        s_in, s_out = ops._state_struct(st, B, m, N), ops._state_struct(so_, B, m, N)
        t_ = ops._timestep_struct(ts, B, m, N)
        p = nat.EnvParams(1 << 30, -0.03, 0.1, 0)

        def launch():
            rc = lib.rbg_connector_step(B, m, N, C.byref(s_in), C.byref(s_out),
                                        C.c_void_p(act.data_ptr()), C.byref(p), C.byref(t_),
                                        C.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0

        direct = span(launch, 300)
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            launch()
            with torch.cuda.graph(g, stream=side):
                for _ in range(20):
                    launch()
        torch.cuda.current_stream().wait_stream(side)
        graph = span(g.replay, 20) / 20
        print(f"B={B:6d}: direct {direct:7.2f} us/launch   hipGraph of 20 {graph:7.2f} us/launch",
              flush=True)


if __name__ == "__main__" and "--gaps" in sys.argv:
    probe_gaps()


def probe_reset_waves():
    """Timeline of ONE step launch of a running auto-reset rollout (stash on): when do the waves
    that restart an episode finish, compared with the others?  (probe build, see probe_times)"""
    import ctypes as C
    import subprocess

    import numpy as np

    from routing_board_generation_amd import _native as nat

    so = os.path.join(ROOT, "gpurun_out", "librbg_probe_step_times.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + nat.HIPCC_FLAGS +
                          ["-DRBG_PROBE_STEP_TIMES", "-o", so, nat.SRC_PATH])
    lib = C.CDLL(so)
    for name, (restype, argtypes) in nat.SIGNATURES.items():  # (plain ints would be truncated)
        getattr(lib, name).restype, getattr(lib, name).argtypes = restype, argtypes
    B = 4096
    keys = rbg.split(rbg.PRNGKey(0), B)
    st, ts = ops.alloc_state(B, m, N, "cuda"), ops.alloc_timestep(B, m, N, "cuda")
    loop = ops.StepLoop(st, ts, keys, refill_every=16)
    loop.reset(keys)
    for t in range(150):
        loop.step(t)
    torch.cuda.synchronize()
    loop._lib = lib  # the next launches come from the probe build
    for t in range(150, 154):
        loop.step(t)
        torch.cuda.synchronize()
        t0 = ts["num_connections"].cpu().numpy().astype(np.uint32)
        t1 = ts["total_path_length"].cpu().numpy().astype(np.uint32)
        last = ts["step_type"].cpu().numpy() == 2
        base = t0.min()
        start = (t0 - base).astype(np.int64) * 10
        end = (t1 - base).astype(np.int64) * 10
        print(f"step {t}: {int(last.sum())} restarting waves; launch span {end.max() / 1e3:.2f} us; "
              f"others end p50 {np.percentile(end[~last], 50) / 1e3:.2f} p99 "
              f"{np.percentile(end[~last], 99) / 1e3:.2f} max {end[~last].max() / 1e3:.2f}; "
              f"restarting end p50 {np.percentile(end[last], 50) / 1e3:.2f} max "
              f"{end[last].max() / 1e3:.2f}; their duration p50 "
              f"{np.percentile((end - start)[last], 50) / 1e3:.2f} vs "
              f"{np.percentile((end - start)[~last], 50) / 1e3:.2f} us")


if __name__ == "__main__" and "--resets" in sys.argv:
    probe_reset_waves()
