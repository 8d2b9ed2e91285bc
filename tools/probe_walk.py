#!/usr/bin/env python3
"""Developer probe (not part of the product or the tests): where does prw_generate_kernel spend
its time?  Times the kernel over batch sizes, and optionally over probe builds that cap the walk
loop (-DRBG_PROBE_MAX_ITERS=k) to separate init + write-out from the per-iteration cost.

    python tools/probe_walk.py [--caps 0,8,16,32] [--m 16 --n 10]
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import routing_board_generation_amd as rbg  # noqa: E402
from routing_board_generation_amd import _native as nat  # noqa: E402
from routing_board_generation_amd import ops  # noqa: E402


def time_generate(B, m, N, reps=100, lib=None):
    keys = rbg.split(rbg.PRNGKey(0), B)
    out = ops.prw_generate(keys, m, N, split_first=True, want_solved=False, want_state=True,
                           want_iters=True)
    torch.cuda.synchronize()

    def launch():
        if lib is None:
            ops.prw_generate(keys, m, N, split_first=True, want_solved=False, want_state=True,
                             want_iters=True, out=out)
        else:
            rc = lib.rbg_prw_generate(
                C.c_void_p(keys.data_ptr()), B, m, N, 1, None, C.c_void_p(out["train_grid"].data_ptr()),
                C.c_void_p(out["start"].data_ptr()), C.c_void_p(out["target"].data_ptr()),
                C.c_void_p(out["position"].data_ptr()), C.c_void_p(out["step_count"].data_ptr()),
                C.c_void_p(out["key_out"].data_ptr()), C.c_void_p(out["iters"].data_ptr()),
                C.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0, rc

    for _ in range(5):
        launch()
    # back-to-back launches between ONE event pair: a pair per launch has a ~12 us floor here
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ts = [e0.elapsed_time(e1) / reps]
    it = out["iters"]
    return ts[len(ts) // 2] * 1e3, float(it.float().mean()), int(it.max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=16)
    ap.add_argument("--n", type=int, default=10)
    ap.add_argument("--caps", type=str, default="")
    ap.add_argument("--stages", type=str, default="",
                    help="probe builds that stop early: 0 empty kernel, 1 after the shuffle, "
                         "2 after board_clear (all with cap 0)")
    ap.add_argument("--batches", type=str, default="64,256,1024,2048,4096,8192,16384,65536")
    args = ap.parse_args()
    print(f"# {args.m}x{args.m}/{args.n}: kernel time (us, p50) vs batch")
    for B in [int(x) for x in args.batches.split(",")]:
        us, mean_it, max_it = time_generate(B, args.m, args.n)
        print(f"B={B:6d}  {us:9.1f} us  {B / us:8.2f} boards/us  iters mean {mean_it:.1f} max {max_it}")
    for cap in [int(x) for x in args.caps.split(",") if x]:
        so = os.path.join(ROOT, "gpurun_out", f"librbg_probe_{cap}.so")
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + nat.HIPCC_FLAGS +
                              [f"-DRBG_PROBE_MAX_ITERS={cap}", "-o", so, nat.SRC_PATH])
        lib = C.CDLL(so)
        for B in (1024, 4096):
            us, mean_it, max_it = time_generate(B, args.m, args.n, lib=lib)
            print(f"cap={cap:4d} B={B:5d}  {us:9.1f} us  (iters mean {mean_it:.1f} max {max_it})")
    probe_stages(args)


def probe_stages(args):
    for stage in [int(x) for x in args.stages.split(",") if x]:
        so = os.path.join(ROOT, "gpurun_out", f"librbg_probe_stage{stage}.so")
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + nat.HIPCC_FLAGS +
                              [f"-DRBG_PROBE_STAGE={stage}", "-DRBG_PROBE_MAX_ITERS=0", "-o", so,
                               nat.SRC_PATH])
        lib = C.CDLL(so)
        for B in (1024, 4096):
            us, _, _ = time_generate(B, args.m, args.n, lib=lib)
            print(f"stage={stage} B={B:5d}  {us:9.1f} us")


if __name__ == "__main__":
    main()
