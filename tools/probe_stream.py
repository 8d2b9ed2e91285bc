"""Developer probe: connector_step_kernel as a pure stream at three batch sizes, for the shipped
build or a variant (e.g. `python tools/probe_stream.py -DRBG_PROBE_PLAIN_OBS_STORES`: plain instead
of non-temporal observation stores)."""
import os, sys
sys.path.insert(0, os.getcwd())
from routing_board_generation_amd import _native as nat
flags = sys.argv[1] if len(sys.argv) > 1 else ""
if flags:
    so = os.path.join(os.getcwd(), "gpurun_out", "librbg_probe_stream.so")
    nat.hipcc_compile(flags.split() + ["-o", so, nat.SRC_PATH]); nat.LIB_PATH = so
import torch, json
import routing_board_generation_amd as rbg
from routing_board_generation_amd import ops
import bench
dev = torch.device("cuda", 0)
for Bs in (4096, 65536, 262144):
    r = bench.pure_stream(rbg, ops, torch, dev, 16, 10, bench.env_step_bytes(16, 10), Bs, copy_gbs=5800.0)
    print(flags or "default", Bs, round(r["kernel_avg_ms"]*1e3, 1), "us", round(r["achieved"], 0), "GB/s", flush=True)
    torch.cuda.empty_cache()
