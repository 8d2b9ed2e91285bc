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
    ap.add_argument("--placement", action="store_true",
                    help="probe build that reports where each wave ran (HW_ID / XCC_ID): waves "
                         "per SIMD for one launch of 4096 boards")
    ap.add_argument("--times", action="store_true",
                    help="probe build that records every wave's start and end time (s_memtime): "
                         "start delay and duration by block index for one launch of 4096 boards")
    ap.add_argument("--subsets", action="store_true",
                    help="time subsets / orderings of the C2 keys (longest pair alone, sorted by "
                         "walk length, ...): what bounds the launch")
    ap.add_argument("--prio", type=str, default="",
                    help="probe builds that raise the wave priority once the walk passes an "
                         "iteration count: comma list of AT:PRIO pairs, e.g. 24:1,32:1,32:3")
    ap.add_argument("--flags", type=str, default="",
                    help="probe builds with extra hipcc flags: ';'-separated flag sets, e.g. "
                         "'-falign-loops=64;-falign-loops=256'")
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
    for spec in [x for x in args.prio.split(",") if x]:
        at, prio = (int(v) for v in spec.split(":"))
        so = os.path.join(ROOT, "gpurun_out", f"librbg_probe_prio_{at}_{prio}.so")
        os.makedirs(os.path.dirname(so), exist_ok=True)
        if at == -1:    # whole loop at `prio`, never lowered
            defs = [f"-DRBG_WALK_PRIO_LOOP={prio}"]
        elif at == -2:  # whole loop at `prio`, back to 0 for the write-out
            defs = [f"-DRBG_WALK_PRIO_LOOP={prio}", "-DRBG_WALK_PRIO_RESET"]
        else:
            defs = [f"-DRBG_WALK_PRIO_AT={at}", f"-DRBG_WALK_PRIO={prio}"]
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + nat.HIPCC_FLAGS +
                              defs + ["-o", so, nat.SRC_PATH])
        lib = C.CDLL(so)
        for B in (4096, 16384):
            runs = [time_generate(B, args.m, args.n, reps=300, lib=lib)[0] for _ in range(3)]
            print(f"prio {prio} from iteration {at:3d}: B={B:5d}  "
                  + "  ".join(f"{u:7.2f}" for u in runs) + " us", flush=True)
    for i, fl in enumerate([x for x in args.flags.split(";") if x]):
        so = os.path.join(ROOT, "gpurun_out", f"librbg_probe_flags_{i}.so")
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + nat.HIPCC_FLAGS +
                              fl.split() + ["-o", so, nat.SRC_PATH])
        lib = C.CDLL(so)
        for B in (100, 4096, 16384):
            runs = [time_generate(B, args.m, args.n, reps=300, lib=lib)[0] for _ in range(3)]
            print(f"flags {fl!r}: B={B:5d}  " + "  ".join(f"{u:7.2f}" for u in runs) + " us",
                  flush=True)
    if args.placement:
        probe_placement(args)
    if args.times:
        probe_times(args)
    if args.subsets:
        probe_subsets(args)


def probe_subsets(args):
    m, N = args.m, args.n
    keys = rbg.split(rbg.PRNGKey(0), 4096)
    full = ops.prw_generate(keys, m, N, split_first=True, want_solved=False, want_state=True,
                            want_iters=True)
    it = full["iters"].cpu()
    order = torch.argsort(it)

    def timeit(idx, reps=300):
        k = keys.view(torch.int32)[idx.to(keys.device)].contiguous().view(torch.uint32)
        out = ops.prw_generate(k, m, N, split_first=True, want_solved=False, want_state=True,
                               want_iters=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.prw_generate(k, m, N, split_first=True, want_solved=False, want_state=True,
                             want_iters=True, out=out)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3, int(out["iters"].max()), float(out["iters"].float().mean())

    pm = torch.maximum(it[0::2], it[1::2])
    po = torch.argsort(pm, descending=True)
    cases = (("longest pair alone", order[-2:]), ("median pair alone", order[2047:2049]),
             ("shortest pair alone", order[:2]), ("2048 longest boards", order[-2048:]),
             ("2048 shortest boards", order[:2048]), ("all 4096, natural order", torch.arange(4096)),
             ("all 4096, sorted by T descending", order.flip(0)),
             ("all 4096, natural pairs ordered by max T descending",
              torch.stack([2 * po, 2 * po + 1], 1).reshape(-1)))
    for name, idx in cases:
        us, mx, mean = timeit(idx)
        print(f"{name:52s} B={idx.numel():5d} {us:7.2f} us  max T {mx:3d}  mean T {mean:5.1f}")


def probe_times(args):
    """Per-wave timeline of one C2-shaped launch: 8 s_memtime stamps per wave (entry, keys,
    before / after the start-cell shuffle, board ready, loop start, loop end, outputs stored).
    s_memtime is per XCD, so times are taken relative to the first wave of the same XCD."""
    import numpy as np

    so = os.path.join(ROOT, "gpurun_out", "librbg_probe_times.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + nat.HIPCC_FLAGS +
                          ["-DRBG_PROBE_TIMES", "-o", so, nat.SRC_PATH])
    lib = C.CDLL(so)
    B = 4096
    keys = rbg.split(rbg.PRNGKey(0), B)
    out = ops.prw_generate(keys, args.m, args.n, split_first=True, want_solved=True,
                           want_state=True, want_iters=True)
    for rep in range(3):
        rc = lib.rbg_prw_generate(
            C.c_void_p(keys.data_ptr()), B, args.m, args.n, 1,
            C.c_void_p(out["solved_grid"].data_ptr()),
            C.c_void_p(out["train_grid"].data_ptr()), C.c_void_p(out["start"].data_ptr()),
            C.c_void_p(out["target"].data_ptr()), C.c_void_p(out["position"].data_ptr()),
            C.c_void_p(out["step_count"].data_ptr()), C.c_void_p(out["key_out"].data_ptr()),
            C.c_void_p(out["iters"].data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
        torch.cuda.synchronize()
    n_waves = B // 2
    T = out["iters"].cpu().numpy().reshape(-1, 2).max(axis=1)
    st = out["solved_grid"].cpu().numpy().reshape(-1)[:n_waves * 8].astype(np.uint32).reshape(-1, 8)
    hw = out["step_count"].cpu().numpy().reshape(-1, 2)[:, 0]
    xcd = hw >> 16
    rel = np.zeros((n_waves, 8), np.int64)
    for x in np.unique(xcd):
        sel = xcd == x
        rel[sel] = (st[sel] - st[sel, 0].min()).astype(np.uint32)
    block = np.arange(n_waves) // 4
    second = block >= block.max() // 2 + 1
    names = ["entry", "keys", "pre-shuffle", "shuffle", "board", "pipeline", "loop", "stored"]
    print(f"kernel span per XCD (shader cycles): {[int(rel[xcd == x, 7].max()) for x in np.unique(xcd)]}")
    for label, sel in (("first half of the blocks", ~second), ("second half of the blocks", second)):
        seg = np.diff(rel[sel], axis=1).mean(axis=0)
        print(f"{label}: entry at {rel[sel, 0].mean():7.0f} (max {rel[sel, 0].max()}); segments "
              + ", ".join(f"{n} {v:6.0f}" for n, v in zip(names[1:], seg))
              + f"; loop/T {((rel[sel, 6] - rel[sel, 5]) / np.maximum(T[sel], 1)).mean():6.0f}")
    for w in np.argsort(rel[:, 7])[-6:][::-1]:
        print(f"late finisher: wave {w} (block {block[w]}, xcd {xcd[w]}) T {T[w]}: "
              + " ".join(f"{n} {v}" for n, v in zip(names, rel[w])))
    w = int(np.argmax(T))
    print(f"longest board: wave {w} (block {block[w]}) T {T[w]}: "
          + " ".join(f"{n} {v}" for n, v in zip(names, rel[w])))


def probe_placement(args):
    import collections

    so = os.path.join(ROOT, "gpurun_out", "librbg_probe_hwid.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + nat.HIPCC_FLAGS +
                          ["-DRBG_PROBE_HWID", "-o", so, nat.SRC_PATH])
    lib = C.CDLL(so)
    B = 4096
    keys = rbg.split(rbg.PRNGKey(0), B)
    out = ops.prw_generate(keys, args.m, args.n, split_first=True, want_solved=False,
                           want_state=True, want_iters=True)
    for rep in range(3):
        rc = lib.rbg_prw_generate(
            C.c_void_p(keys.data_ptr()), B, args.m, args.n, 1, None,
            C.c_void_p(out["train_grid"].data_ptr()), C.c_void_p(out["start"].data_ptr()),
            C.c_void_p(out["target"].data_ptr()), C.c_void_p(out["position"].data_ptr()),
            C.c_void_p(out["step_count"].data_ptr()), C.c_void_p(out["key_out"].data_ptr()),
            C.c_void_p(out["iters"].data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
        torch.cuda.synchronize()
        ids = out["iters"].cpu().tolist()
        # one entry per board; boards of one wave report the same place: count waves
        per_simd = collections.Counter()
        per_cu = collections.Counter()
        seen = set()
        lpb_boards = 2 if 2 * args.n + 2 <= 32 else 1
        for b in range(0, B, lpb_boards):
            v = ids[b]
            xcc, se, sh, cu, simd = v >> 16, (v >> 13) & 7, (v >> 12) & 1, (v >> 8) & 15, (v >> 4) & 3
            per_simd[(xcc, se, sh, cu, simd)] += 1
            per_cu[(xcc, se, sh, cu)] += 1
        print(f"launch {rep}: {len(per_cu)} CUs, {len(per_simd)} SIMDs used; waves per SIMD histogram "
              f"{sorted(collections.Counter(per_simd.values()).items())}; waves per CU histogram "
              f"{sorted(collections.Counter(per_cu.values()).items())}")


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
