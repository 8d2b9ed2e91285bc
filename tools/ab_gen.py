#!/usr/bin/env python3
"""Developer probe: prw_generate_kernel of several builds of the library on ONE box.

    python tools/ab_gen.py [--reps 3] label=path/to/lib.so ...

16x16/10, B = 4096 (and B = 100), keys split(PRNGKey(0), B); launches back to back between one
HIP-event pair, libraries alternating; outputs compared bit for bit with the first library's."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import routing_board_generation_amd as rbg  # noqa: E402
from routing_board_generation_amd import ops  # noqa: E402

m, N = 16, 10


def run(lib, keys, out, reps):
    B = keys.shape[0]
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    args = (C.c_void_p(keys.data_ptr()), B, m, N, 1, None, C.c_void_p(out["train_grid"].data_ptr()),
            C.c_void_p(out["start"].data_ptr()), C.c_void_p(out["target"].data_ptr()),
            C.c_void_p(out["position"].data_ptr()), C.c_void_p(out["step_count"].data_ptr()),
            C.c_void_p(out["key_out"].data_ptr()), C.c_void_p(out["iters"].data_ptr()), stream)
    for _ in range(20):
        assert lib.rbg_prw_generate(*args) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        lib.rbg_prw_generate(*args)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    args = sys.argv[1:]
    reps = 3
    if args and args[0] == "--reps":
        reps = int(args[1]); args = args[2:]
    libs = []
    for a in args:
        label, path = a.split("=", 1)
        lib = C.CDLL(path if os.path.isabs(path) else os.path.join(ROOT, path))
        lib.rbg_prw_generate.restype = C.c_int
        libs.append((label, lib))
    for B in (4096, 100):
        keys = rbg.split(rbg.PRNGKey(0), B)
        ref = None
        rows = {label: [] for label, _ in libs}
        for r in range(reps):
            for label, lib in libs:
                out = ops.prw_generate(keys, m, N, split_first=True, want_solved=False,
                                       want_state=True, want_iters=True)
                us = run(lib, keys, out, 400)
                torch.cuda.synchronize()
                if ref is None:
                    ref = {k: v.clone() for k, v in out.items()}
                same = all(torch.equal(out[k].view(torch.int32) if out[k].dtype == torch.uint32 else out[k],
                                       ref[k].view(torch.int32) if ref[k].dtype == torch.uint32 else ref[k])
                           for k in ref)
                assert same, f"{label}: output differs"
                rows[label].append(us)
        for label, _ in libs:
            v = sorted(rows[label])
            print(f"B={B:5d} {label:>16s}  " + "  ".join(f"{x:7.2f}" for x in rows[label]) +
                  f"   median {v[len(v) // 2]:7.2f} us", flush=True)


if __name__ == "__main__":
    main()
