"""Developer probe: env-steps/s of the per-step path for every generation_type under different
next-episode-stash schedules (refill period K, refill on the stepping stream or beside it, one or
two slots per env).  Decides the defaults of GeneratorSpec.refill.

  python tools/probe_envgen.py [--steps 600] [--names online_seq_rw,online_random_seed]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--names", default="online_parallel_rw,online_uniform,offline_parallel_rw,"
                                       "online_seq_rw,online_random_seed")
    ap.add_argument("--grid", type=int, default=16)
    ap.add_argument("--agents", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--configs", default="0:1:1,16:1:1,16:1:2,48:1:2,16:0:1,16:0:2,32:0:2,48:0:2",
                    help="K:serial:depth[:lead[:streams]]; K = 0 is the inline (fused) reset; serial = 0: "
                         "rolling refills beside the stepping stream")
    ap.add_argument("--time-limit", type=int, default=50)
    ap.add_argument("--warm", type=int, default=60, help="untimed steps before the timed window")
    ap.add_argument("--chunks", type=int, default=1, help="time the window in this many pieces")
    ap.add_argument("--flags", default="", help="extra hipcc flags: time a variant build of the "
                                                "library (built here, into gpurun_out/)")
    ap.add_argument("--lib", default="", help="time a prebuilt library of the same ABI instead "
                                              "of the tree's (tools/build_ab_libs.sh)")
    ap.add_argument("--no-miss-snapshot", action="store_true",
                    help="rolling refills without the per-epoch miss counter snapshot")
    args = ap.parse_args()
    if args.no_miss_snapshot:
        from routing_board_generation_amd import session

        session.EpochStash.SNAPSHOT_MISSES = False
    if args.lib:
        from routing_board_generation_amd import _native as nat

        nat.LIB_PATH = os.path.abspath(args.lib)  # before anything loads the library
    if args.flags:
        from routing_board_generation_amd import _native as nat

        so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out",
                          "librbg_probe_envgen.so")
        os.makedirs(os.path.dirname(so), exist_ok=True)
        nat.hipcc_compile(args.flags.split() + ["-o", so, nat.SRC_PATH])
        nat.LIB_PATH = so  # before anything loads the library
    import torch

    import routing_board_generation_amd as rbg
    from routing_board_generation_amd import ops
    from routing_board_generation_amd.interface.board_generator_interface import make_online_generator

    dev = torch.device("cuda", 0)
    m, N, B = args.grid, args.agents, args.batch
    keys = rbg.split(rbg.PRNGKey(0, device=dev), B)
    rows = []
    for name in args.names.split(","):
        kw = dict(randomness=0.0, two_sided=True, number_of_boards=10000) if name.startswith("offline") else {}
        gen = make_online_generator(name, m, N, **kw).native_spec()
        # episodes of the timed window (the trajectory does not depend on the stash schedule)
        st, ts = ops.alloc_state(B, m, N, dev), ops.alloc_timestep(B, m, N, dev)
        loop = ops.StepLoop(st, ts, keys, refill_every=0, fuse_policy=True, gen=gen,
                            time_limit=args.time_limit)
        loop.reset(keys)
        n_last = torch.zeros((), dtype=torch.int64, device=dev)
        for t in range(args.warm + args.steps):
            loop.step(t)
            if t >= args.warm:
                n_last += (ts["step_type"] == 2).sum()
        episodes = int(n_last.item())
        for cfg in args.configs.split(","):
            K, serial, depth, lead, streams = ([int(x) for x in cfg.split(":")] + [2, 2])[:5]
            st, ts = ops.alloc_state(B, m, N, dev), ops.alloc_timestep(B, m, N, dev)
            loop = ops.StepLoop(st, ts, keys, refill_every=K, fuse_policy=True, gen=gen,
                                serial_refill=bool(serial), stash_depth=depth, refill_lead=lead,
                                refill_streams=streams, time_limit=args.time_limit)
            loop.reset(keys)
            for t in range(args.warm):
                loop.step(t)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.chunks + 1)]
            torch.cuda.synchronize(dev)
            if K:
                loop.stash["hits"].zero_()
                loop.stash["misses"].zero_()
            t0 = time.perf_counter()
            e0.record()
            marks[0].record()
            per = max(1, args.steps // args.chunks)
            for t in range(args.warm, args.warm + args.steps):
                loop.step(t)
                if (t - args.warm + 1) % per == 0 and (t - args.warm + 1) // per <= args.chunks:
                    marks[(t - args.warm + 1) // per].record()
            e1.record()
            torch.cuda.synchronize(dev)
            wall = time.perf_counter() - t0
            loop.drain()
            us = e0.elapsed_time(e1) / args.steps * 1e3
            hits = int(loop.stash["hits"].sum().item()) if K else 0
            misses = int(loop.stash["misses"].sum().item()) if K else 0
            rows.append({"generator": name, "batch": B, "K": K, "serial": serial, "depth": depth,
                         **({} if serial else {"lead": lead, "streams": streams,
                                               "fell_back": loop._epochs.fell_back}),
                         "time_limit": args.time_limit, "misses": misses,
                         "us_per_step": round(us, 2), "wall_us_per_step": round(wall / args.steps * 1e6, 2),
                         "env_steps_per_s": round(B / us * 1e6 / 1e6, 1),
                         "episodes": episodes, "stash_hits": hits,
                         **({"chunks_us_per_step": [round(a.elapsed_time(b) / per * 1e3, 1)
                                                    for a, b in zip(marks, marks[1:])]}
                            if args.chunks > 1 else {})})
            print(json.dumps(rows[-1]), flush=True)
            torch.cuda.synchronize(dev)


if __name__ == "__main__":
    main()
