"""Developer probe: the fused random rollout with its stash refilled BESIDE the next launch
(EpochStash(every=1, serial=False, depth=2): requested in launch L, generated during launch
L + 1, consumed in launch L + 2) against the refill behind every launch.
usage: probe_rollout_rolling.py [generation_types] [steps-per-launch] [launches]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import routing_board_generation_amd as rbg  # noqa: E402
from routing_board_generation_amd import ops  # noqa: E402
from routing_board_generation_amd.interface.board_generator_interface import make_online_generator  # noqa: E402

B, m, N = 4096, 16, 10
T = int(sys.argv[2]) if len(sys.argv) > 2 else 50
L = int(sys.argv[3]) if len(sys.argv) > 3 else 40
if os.environ.get("RBG_PROBE_FLAGS"):
    from routing_board_generation_amd import _native as nat

    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out",
                      "librbg_probe_rollout.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    nat.hipcc_compile(os.environ["RBG_PROBE_FLAGS"].split() + ["-o", so, nat.SRC_PATH])
    nat.LIB_PATH = so  # before anything loads the library
DEPTH = int(os.environ.get("DEPTH", "2"))
dev = torch.device("cuda", 0)
keys = rbg.split(rbg.PRNGKey(0, device=dev), B)
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["online_parallel_rw"]
for name in names:
    gen = make_online_generator(name, m, N).native_spec()
    final = {}
    for cfg in os.environ.get("CFGS", "serial,rolling:1,rolling:2,serial,rolling:1").split(","):
        for record in (False, True):
            if cfg == "serial":
                ep = ops.EpochStash(B, m, N, dev, every=1, serial=True, depth=DEPTH, gen=gen)
            else:
                ep = ops.EpochStash(B, m, N, dev, every=1, serial=False, depth=DEPTH, gen=gen, lead=2,
                                    streams=int(cfg.split(":")[1]))
                if os.environ.get("NO_WATCH"):
                    ep._watch_misses = lambda s: None
            ep.before_launch()
            st, _ = ops.connector_reset_prefetch(keys, m, N, ep.stash, ep.tick, ep.ready_delay(), gen=gen)
            ep.after_launch()
            traj = ops.alloc_trajectory(T, B, m, N, dev) if record else None
            stats = {k: torch.empty((B,), dtype=torch.int32, device=dev) for k in ops.ROLLOUT_STATS}

            def launch(i):
                if record:
                    ops.connector_rollout_record(st, keys, i * T, T, state_out=st, trajectory=traj,
                                                 epochs=ep, gen=gen)
                else:
                    ops.connector_rollout(st, keys, i * T, T, state_out=st, stats=stats, epochs=ep, gen=gen)

            for i in range(4):
                launch(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            h0, m0 = int(ep.stash["hits"].sum()), int(ep.stash["misses"].sum())
            e0.record()
            for i in range(4, 4 + L):
                launch(i)
            e1.record()
            torch.cuda.synchronize()
            ep.drain()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / L * 1e3
            digest = int(st["grid"].to(torch.int64).sum()) * 31 + int(st["key"].to(torch.int64).sum())
            same = final.setdefault(record, digest) == digest
            print(json.dumps({"generator": name, "refill": cfg, "record": record, "steps": T,
                              "us_per_launch": round(us, 1),
                              "env_steps_per_s_M": round(B * T / us, 1),
                              "hits": int(ep.stash["hits"].sum()) - h0,
                              "misses": int(ep.stash["misses"].sum()) - m0,
                              "fell_back": ep.fell_back, "same_final_state": same}), flush=True)
