#!/usr/bin/env python3
"""Developer probe: fused rollout launch time for full-wave envs (LPB 64: N >= 16), with the
library as built and with probe builds given as --flags='-D...;-D...'."""
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


def run(lib, label):
    for (m, N, B, n_steps) in ((32, 20, 4096, 50), (16, 40, 4096, 50), (16, 10, 4096, 50)):
        keys, pkeys = rbg.split(rbg.PRNGKey(0), B), rbg.split(rbg.PRNGKey(1), B)
        st, _ = ops.connector_reset(keys, m, N)
        s = ops._state_struct(st, B, m, N)
        p = nat.EnvParams(50, -0.03, 0.1, 1)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

        def launch(i):
            rc = lib.rbg_connector_rollout(B, m, N, C.byref(s), C.byref(s),
                                           C.c_void_p(pkeys.data_ptr()), i * n_steps, n_steps,
                                           C.byref(p), None, None, stream)
            assert rc == 0, rc

        for i in range(3):
            launch(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(10):
            launch(3 + i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        print(f"{label}: {m}x{m}/{N} B={B} {n_steps} steps/launch: {us:8.1f} us  "
              f"{B * n_steps / us:7.1f} env-steps/us", flush=True)


for a in nat.SIGNATURES:
    pass
run(nat.lib(), "as built")
for i, fl in enumerate([x for a in sys.argv if a.startswith("--flags=") for x in a[8:].split(";") if x]):
    so = os.path.join(ROOT, "gpurun_out", f"librbg_probe_rollout_{i}.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + nat.HIPCC_FLAGS +
                          fl.split() + ["-o", so, nat.SRC_PATH])
    lib = C.CDLL(so)
    for name, (restype, argtypes) in nat.SIGNATURES.items():
        getattr(lib, name).restype, getattr(lib, name).argtypes = restype, argtypes
    run(lib, fl)
