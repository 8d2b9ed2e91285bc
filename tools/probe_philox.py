#!/usr/bin/env python3
"""Developer probe: the generator on the opt-in Philox stream against the default (jax Threefry)
path, per batch size; --flags=... also times a probe build."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import routing_board_generation_amd as rbg
from routing_board_generation_amd import ops
def timed(fn, reps):
    for _ in range(3): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
import ctypes as C, os, subprocess
from routing_board_generation_amd import _native as nat
libs=[("lpb16", nat.lib())]
for fl in [a[8:] for a in sys.argv if a.startswith("--flags=")]:
    so=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "librbg_probe_philox.so")
    subprocess.check_call([os.environ.get("HIPCC","/opt/rocm/bin/hipcc")]+nat.HIPCC_FLAGS+fl.split()+["-o",so,nat.SRC_PATH])
    lib=C.CDLL(so)
    for name,(rt,at) in nat.SIGNATURES.items(): getattr(lib,name).restype,getattr(lib,name).argtypes=rt,at
    libs.append((fl,lib))
if len(libs)>1:
    for B in (4096,4096,16384,65536):
        keys=rbg.split(rbg.PRNGKey(0),B); o=ops.prw_generate_philox(keys,16,10)
        st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for label,lib in libs:
            f=lambda: lib.rbg_prw_generate_philox(keys.data_ptr(),B,16,10,None,o["train_grid"].data_ptr(),o["start"].data_ptr(),o["target"].data_ptr(),o["position"].data_ptr(),o["step_count"].data_ptr(),o["key_out"].data_ptr(),o["iters"].data_ptr(),st)
            print(f"B={B} {label}: {timed(f,200):8.2f} us",flush=True)
    sys.exit(0)
for (m,N,B) in ((16,10,4096),(16,10,16384),(16,10,65536),(32,20,4096),(8,5,4096)):
    keys=rbg.split(rbg.PRNGKey(0),B)
    o1=ops.prw_generate(keys,m,N,split_first=True,want_solved=False,want_state=True,want_iters=True)
    o2=ops.prw_generate_philox(keys,m,N)
    t1=timed(lambda: ops.prw_generate(keys,m,N,split_first=True,want_solved=False,want_state=True,want_iters=True,out=o1),100)
    t2=timed(lambda: ops.prw_generate_philox(keys,m,N,out=o2),100)
    print(f"{m}x{m}/{N} B={B}: threefry(jax) {t1:8.1f} us   philox {t2:8.1f} us   T max {int(o1['iters'].max())} / {int(o2['iters'].max())}", flush=True)
