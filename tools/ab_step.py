#!/usr/bin/env python3
"""Developer probe: same-box A/B of connector_step_kernel across several builds of the library.

    python tools/ab_step.py [--reps 3] [--steps 300] label=path/to/lib.so[:abi7] ...
    python tools/ab_step.py --build "lean=-DRBG_PROBE_X ..." ...   (hipcc on the box, from csrc/)

Every library is driven through the entry points that exist since ABI 7 (rbg_connector_reset_prefetch,
rbg_connector_step_prefetch, rbg_stash_refill, rbg_random_actions), so a build of an earlier round
can sit beside the current one.  Two populations per library, alternating over the libraries,
`reps` times:
  replay  the auto-resetting rollout of bench.py --mode step-replay (B = 4096, 16x16/10, one-slot
          stash refilled every 20 launches on the launches' own stream, actions of a recorded pass):
          HIP-event pairs around every run of step launches between two refills
  pure    NOOP actions, auto-reset off, no stash: the common path of the kernel alone
The final State of every replay is compared across the libraries (they must agree bit for bit).
Prints one JSON line per (library, repetition) and a summary table."""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import routing_board_generation_amd as rbg  # noqa: E402
from routing_board_generation_amd import _native as nat  # noqa: E402
from routing_board_generation_amd import ops  # noqa: E402

m, N, B, K = 16, 10, 4096, 20


class Stash7(C.Structure):  # include/rbg_prw.h::rbg_stash as of ABI 7 (round 2)
    _fields_ = [("next", nat.State), ("tag", C.c_void_p), ("ready_at", C.c_void_p),
                ("refill_key", C.c_void_p), ("hits", C.c_void_p), ("depth", C.c_int32)]


def load(path):
    lib = C.CDLL(path)
    for fn in ("rbg_connector_reset_prefetch", "rbg_connector_step_prefetch", "rbg_stash_refill",
               "rbg_random_actions", "rbg_connector_step", "rbg_abi_version"):
        getattr(lib, fn).restype = C.c_int
    return lib


def vp(t):
    return C.c_void_p(t.data_ptr())


def state_struct(s):
    return nat.State(*(s[f].data_ptr() for f in ops.STATE_FIELDS))


def timestep_struct(t):
    return nat.Timestep(*(t[f].data_ptr() if t.get(f) is not None else None
                          for f in ops.TIMESTEP_FIELDS))


def make_stash(abi, dev):
    nxt = ops.alloc_state(B, m, N, dev)
    bufs = {"next": nxt,
            "tag": torch.zeros((B, 2), dtype=torch.uint32, device=dev),
            "ready_at": torch.zeros((B,), dtype=torch.int32, device=dev),
            "refill_key": torch.zeros((B, 2), dtype=torch.uint32, device=dev),
            "hits": torch.zeros((B,), dtype=torch.int32, device=dev),
            "misses": torch.zeros((B,), dtype=torch.int32, device=dev)}
    if abi <= 7:
        st = Stash7(state_struct(nxt), bufs["tag"].data_ptr(), bufs["ready_at"].data_ptr(),
                    bufs["refill_key"].data_ptr(), bufs["hits"].data_ptr(), 1)
    else:
        st = nat.Stash(state_struct(nxt), bufs["tag"].data_ptr(), bufs["ready_at"].data_ptr(),
                       bufs["refill_key"].data_ptr(), bufs["hits"].data_ptr(),
                       bufs["misses"].data_ptr(), None, None, None, 1)
    return bufs, st


def rollout(lib, keys, n, rec, record, dev):
    """One auto-resetting rollout of n steps through `lib`; record=True draws the actions with
    rbg_random_actions and stores them in rec, else replays rec.  Returns (ms per step launch
    inside the runs between refills, ms per step overall, final state, hits)."""
    abi = lib.rbg_abi_version()
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    st, ts = ops.alloc_state(B, m, N, dev), ops.alloc_timestep(B, m, N, dev)
    s, t = state_struct(st), timestep_struct(ts)
    bufs, stash = make_stash(abi, dev)
    p = nat.EnvParams(50, -0.03, 0.1, 1)
    action = torch.empty((B, N), dtype=torch.int32, device=dev)
    tick, refill_after = 0, 0

    def ck(rc):
        assert rc == 0, rc

    def refill():
        ck(lib.rbg_stash_refill(B, m, N, C.byref(stash), stream))

    ck(lib.rbg_connector_reset_prefetch(vp(keys), B, m, N, C.byref(s), C.byref(t), C.byref(stash),
                                        0, 1, stream))
    refill()
    refill_after, tick = K, 1
    spans = [[torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    e0.record()
    spans[-1][0].record()
    for i in range(n):
        if record:
            ck(lib.rbg_random_actions(vp(keys), B, N, i, vp(ts["action_mask"]), vp(action), stream))
            rec[i].copy_(action)
            a = action
        else:
            a = rec[i]
        ck(lib.rbg_connector_step_prefetch(B, m, N, C.byref(s), C.byref(s), vp(a), C.byref(p),
                                           C.byref(t), C.byref(stash), tick,
                                           refill_after + 1 - tick, stream))
        if tick == refill_after:
            spans[-1][1].record()
            refill()
            refill_after = tick + K
            spans.append([torch.cuda.Event(enable_timing=True),
                          torch.cuda.Event(enable_timing=True)])
            spans[-1][0].record()
        tick += 1
    e1.record()
    spans[-1][1].record()
    torch.cuda.synchronize(dev)
    in_runs = sum(a_.elapsed_time(b_) for a_, b_ in spans)
    return in_runs / n, e0.elapsed_time(e1) / n, st, int(bufs["hits"].sum().item())


def pure(lib, keys, reps, dev):
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    st, ts = ops.connector_reset(keys, m, N)
    so = ops.alloc_state(B, m, N, dev)
    s_in, s_out, t = state_struct(st), state_struct(so), timestep_struct(ts)
    p = nat.EnvParams(1 << 30, -0.03, 0.1, 0)
    act = torch.zeros((B, N), dtype=torch.int32, device=dev)
    for _ in range(20):
        lib.rbg_connector_step(B, m, N, C.byref(s_in), C.byref(s_out), vp(act), C.byref(p),
                               C.byref(t), stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    e0.record()
    for _ in range(reps):
        rc = lib.rbg_connector_step(B, m, N, C.byref(s_in), C.byref(s_out), vp(act), C.byref(p),
                                    C.byref(t), stream)
        assert rc == 0
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / reps


def main():
    args = sys.argv[1:]
    reps, n = 3, 300
    libs = []
    i = 0
    while i < len(args):
        a = args[i]
        if a == "--reps":
            reps = int(args[i + 1]); i += 2; continue
        if a == "--steps":
            n = int(args[i + 1]); i += 2; continue
        if a == "--build":  # label=flags: hipcc on this box from the tree's csrc/
            label, flags = args[i + 1].split("=", 1)
            so = os.path.join(ROOT, "gpurun_out", f"librbg_ab_{label}.so")
            os.makedirs(os.path.dirname(so), exist_ok=True)
            subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] +
                                  nat.HIPCC_FLAGS + flags.split() + ["-o", so, nat.SRC_PATH])
            libs.append((label, so)); i += 2; continue
        label, path = a.split("=", 1)
        libs.append((label, path if os.path.isabs(path) else os.path.join(ROOT, path)))
        i += 1
    if not libs:
        libs = [("tree", nat.LIB_PATH)]
    dev = torch.device("cuda", 0)
    keys = rbg.split(rbg.PRNGKey(0, device=dev), B)
    loaded = [(label, load(path)) for label, path in libs]
    rec = torch.empty((n, B, N), dtype=torch.int32, device=dev)
    _, _, ref_state, _ = rollout(loaded[0][1], keys, n, rec, True, dev)
    for _, lib in loaded:  # warm every library up (code upload, clocks)
        rollout(lib, keys, n, rec, False, dev)
    rows = {label: {"replay": [], "span": [], "pure": []} for label, _ in loaded}
    for r in range(reps):
        for label, lib in loaded:
            k_step, k_span, st, hits = rollout(lib, keys, n, rec, False, dev)
            same = all(torch.equal(st[f], ref_state[f]) for f in ops.STATE_FIELDS)
            pu = pure(lib, keys, 300, dev)
            rows[label]["replay"].append(k_step * 1e3)
            rows[label]["span"].append(k_span * 1e3)
            rows[label]["pure"].append(pu * 1e3)
            print(json.dumps({"lib": label, "rep": r, "abi": lib.rbg_abi_version(),
                              "replay_us": round(k_step * 1e3, 3),
                              "with_refills_us": round(k_span * 1e3, 3),
                              "pure_us": round(pu * 1e3, 3), "stash_hits": hits,
                              "final_state_equal": bool(same)}), flush=True)
            assert same, f"{label}: final State differs from {loaded[0][0]}'s"
    print(f"{'lib':>14s} {'replay us (min / median / max)':>34s} {'pure us (min / median / max)':>34s}")
    for label, _ in loaded:
        def f(v):
            v = sorted(v)
            return f"{v[0]:7.2f} / {v[len(v) // 2]:7.2f} / {v[-1]:7.2f}"
        print(f"{label:>14s} {f(rows[label]['replay']):>34s} {f(rows[label]['pure']):>34s}")


if __name__ == "__main__":
    main()
