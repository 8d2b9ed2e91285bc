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


import ctypes as C  # noqa: E402
import subprocess  # noqa: E402

from routing_board_generation_amd import _native as nat  # noqa: E402

# --flags='-DX;-DY': also time probe builds (C5 shape only), interleaved with the library as built
variants = [("as built", nat.lib())]
for i, fl in enumerate([x for a in sys.argv if a.startswith("--flags=") for x in a[8:].split(";") if x]):
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out",
                      f"librbg_probe_se_{i}.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + nat.HIPCC_FLAGS +
                          fl.split() + ["-o", so, nat.SRC_PATH])
    lib = C.CDLL(so)
    for name, (restype, argtypes) in nat.SIGNATURES.items():
        getattr(lib, name).restype, getattr(lib, name).argtypes = restype, argtypes
    variants.append((fl, lib))
if len(variants) > 1:
    m, N, B = 16, 10, 4096
    keys = rbg.split(rbg.PRNGKey(0), B)
    out = ops.seed_extension_generate(keys, m, N, split_first=True)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for rep in range(3):
        for label, lib in variants:
            def launch():
                rc = lib.rbg_seed_extension_generate(
                    keys.data_ptr(), B, m, N, 1, 0.0, 1, 1, 2 ** 31 - 1, None,
                    out["train_grid"].data_ptr(), out["start"].data_ptr(),
                    out["target"].data_ptr(), out["position"].data_ptr(),
                    out["step_count"].data_ptr(), out["key_out"].data_ptr(),
                    out["sweeps"].data_ptr(), stream)
                assert rc == 0, rc
            print(f"{label:28s} {timed(launch, 5):9.1f} us per launch (C5: 16x16/10, B=4096)", flush=True)
    sys.exit(0)

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
