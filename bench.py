#!/usr/bin/env python3
"""bench.py -- boards/s of the ParallelRandomWalk generator hot path on N MI355X GPUs.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: ParallelRandomWalkGenerator(16, 10) vmapped
over 4096 keys per GPU (BASELINE.json configs[1]), producing the full Connector State in HBM.
Keys are `split(PRNGKey(0), 4096 * n_gpus)` sharded contiguously over the ranks (weak scaling:
per-GPU work is fixed), already resident in HBM when the timed region starts.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the torchrun
command above itself, as a CHILD process (before anything here has touched the GPU), and relays
rank 0's line and the child's exit code.

`--mode step-replay` and `--mode stream` run ONE launch population each (the recorded-action
replay of connector_step_kernel at B = 4096, the B = 65536 NOOP stream) and print only the
env_step part: what the rocprofv3 passes under profiles/ are taken from.

Rank 0 prints ONE JSON line.  Besides the driver contract it carries
  roofline      achieved algorithmic GB/s of the dominant kernel (prw_generate_kernel) against the
                HBM peak, from HIP-event durations measured live around every launch
  cpu_baseline  the CPU oracle (a port, oracle/prw_oracle.c) timed on this box's host cores on a
                bounded sample of the same workload
  env_step      the Connector.step rollout of BASELINE configs[2] (16x16/10, batch 4096, random
                legal actions, fused auto-reset) with its own HBM roofline -- reported beside the
                headline, never folded into `value`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GRID, AGENTS, BATCH_PER_GPU = 16, 10, 4096
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def generator_bytes_per_board(m: int, N: int) -> int:
    """Algorithmic HBM bytes of one generated State (minus Agent.id which is an
    arange that is never materialised): key in + training grid + start/target/position +
    step_count + key out."""
    return 8 + 4 * m * m + 3 * 8 * N + 4 + 8


def env_step_bytes(m: int, N: int) -> int:
    """Algorithmic HBM bytes of one env step: read grid + agents + step_count +
    key + action; write the same state + observation (N x m x m) + mask + reward + discount +
    step_type + extras."""
    read = 4 * m * m + 24 * N + 4 + 8 + 4 * N
    write = 4 * m * m + 24 * N + 4 + 8 + 4 * N * m * m + 5 * N + 8 * N + 1 + 4 + 12
    return read + write


def committed_profile(kind: str, population: str = ""):
    """The newest round's profiles/rNN<population>_<kind>.json (tools/summarize_profile.py).
    population: "" = the default run, "_step" = --mode step-replay, "_stream" = --mode stream."""
    import re

    best = None
    for f in os.listdir(os.path.join(ROOT, "profiles")):
        mt = re.fullmatch(r"r(\d+)([a-z]*)" + re.escape(population) + "_" + kind + r"\.json", f)
        if mt and (best is None or (int(mt.group(1)), mt.group(2)) > best[0]):
            best = ((int(mt.group(1)), mt.group(2)), f)
    return json.load(open(os.path.join(ROOT, "profiles", best[1]))) if best else None


def measured_traffic(kernel: str, boards: int, population: str = "", shape_boards: int = None):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (counters cannot be read from
    inside this process).  Only valid for the launch shape that run measured: B = 4096 for the
    default and "_step" runs, the stream batch for "_stream"."""
    prof = committed_profile("hbm_traffic", population)
    if prof is None or boards != (BATCH_PER_GPU if shape_boards is None else shape_boards):
        return None
    for name, rec in prof["kernels"].items():  # template instances are listed as name<...>
        if name == kernel or name.startswith(kernel + "<"):
            return rec["hbm_bytes_per_launch"]
    return None


def host_cpu_share() -> int:
    """Threads this process may really run in parallel: the cgroup CPU quota when there is one
    (a GPU box hands each job a share of its cores), else the affinity mask; capped at 64."""
    n = len(os.sched_getaffinity(0)) or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, 64)


def host_cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def measured_issue(kernel: str, boards: int):
    """VALU wave-instructions per launch from the committed SQ counter pass
    (profiles/*_issue.json, tools/summarize_profile.py); B = 4096, 16x16/10 only."""
    prof = committed_profile("issue")
    if prof is None or boards != BATCH_PER_GPU:
        return None
    for name, rec in prof["kernels"].items():
        if name == kernel or name.startswith(kernel + "<"):
            return rec
    return None


SIMDS, CLOCK_HZ, VALU_CYCLES = 1024, 2.4e9, 2  # 256 CUs x 4 SIMD-32; a wave64 VALU op = 2 passes


def issue_roofline(kernel: str, boards: int, kernel_avg_ms: float):
    """The generator is issue-, not HBM-bound: its VALU wave-instructions against the rate at
    which 1024 SIMD-32 units can issue them (one wave64 instruction per 2 cycles each)."""
    rec = measured_issue(kernel, boards)
    if rec is None:
        return None
    peak = SIMDS * CLOCK_HZ / VALU_CYCLES  # wave-instructions per second, whole chip
    ach = rec["valu_insts_per_launch"] / (kernel_avg_ms * 1e-3)
    return {"bound": "valu_issue", "achieved": ach / 1e12, "peak": peak / 1e12,
            "unit": "T wave-instr/s", "frac": ach / peak,
            "valu_wave_insts_per_launch": rec["valu_insts_per_launch"],
            "floor_ms": rec["valu_insts_per_launch"] / peak * 1e3,
            "note": "VALU wave-instructions per launch (rocprofv3 SQ_INSTS_VALU, profiles/) / "
                    "launch duration, against 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 "
                    "instruction; the launch also waits for its longest board (DESIGN.md 4.1)"}


def build_microbench() -> str:
    """tools/microbench/hbm_peak.hip -> libhbm_peak.so (built by __graft_entry__.build(); rebuilt
    here if missing -- hipcc is on the GPU box too)."""
    import subprocess

    src = os.path.join(ROOT, "tools", "microbench", "hbm_peak.hip")
    so = os.path.join(ROOT, "tools", "microbench", "libhbm_peak.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"),
                               "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, src])
    return so


def hbm_peaks(dev, nbytes: int = 1 << 32) -> dict:
    """Measured HBM rates of this GPU (quoted beside the 8 TB/s spec figure): a
    16-B-per-lane copy (read + written bytes) and a write-only fill, plain and non-temporal,
    4 GiB buffers = 16x the 256 MiB Infinity Cache (at 1 GiB, round 2's size, a quarter of a
    buffer can stay on chip between launches and the figures come out 10-25 % higher: reported
    as `at_1GiB`); best of the grid sizes tried."""
    import ctypes as C

    import torch

    lib = C.CDLL(build_microbench())
    lib.hbm_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
    lib.hbm_fill.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
    src = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def rate(fn, moved):
        best = 0.0
        for blocks in (1024, 2048, 4096, 16384):
            for _ in range(2):
                fn(blocks)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            e0.record()
            for _ in range(10):
                fn(blocks)
            e1.record()
            torch.cuda.synchronize(dev)
            best = max(best, moved / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e9)
        return best

    out = {}
    for nt in (0, 1):
        tag = "_nontemporal" if nt else ""
        out["copy" + tag] = rate(lambda b: lib.hbm_copy(dst.data_ptr(), src.data_ptr(), nbytes,
                                                        nt, b, stream), 2 * nbytes)
        out["fill" + tag] = rate(lambda b: lib.hbm_fill(dst.data_ptr(), nbytes, nt, b, stream),
                                 nbytes)
    del src, dst
    torch.cuda.empty_cache()
    out["unit"] = "GB/s"
    out["buffer_bytes"] = nbytes
    out["note"] = ("tools/microbench/hbm_peak.hip: float4 copy (bytes read + written) and "
                   f"write-only fill over {nbytes >> 30} GiB buffers, each wave on contiguous "
                   "16 KiB chunks")
    if nbytes > (1 << 30):
        small = hbm_peaks(dev, 1 << 30)
        out["at_1GiB"] = {k: small[k] for k in ("copy", "fill", "copy_nontemporal",
                                                "fill_nontemporal")}
    return out


def copy_peak_gbs(dev) -> float:
    return hbm_peaks(dev)["copy"]


def free_port() -> int:
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n: int, argv) -> int:
    """`bench.py --gpus N` (N > 1) outside a launcher: start the one-process-per-GPU job as a
    CHILD (this parent has not imported torch.cuda or made any HIP call, and never replaces
    itself: an exec from a GPU-initialised process is not allowed on this pool) and relay its
    output and exit code.  Rank 0 of the child prints the JSON line straight to our stdout."""
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def rehearse_launch(args) -> None:
    """CPU-only rehearsal of the multi-rank plumbing (tests/test_bench_launch.py): gloo ranks
    shard the keys' index range, meet at the barriers, all-reduce a metrics vector and rank 0
    prints a line of the driver's shape with no measurement in it."""
    import torch.distributed as dist

    from routing_board_generation_amd import sharding

    rank, _, world = sharding.init_distributed(backend="gloo", single_device=True)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}")
    lo, hi = sharding.shard_bounds(BATCH_PER_GPU * world, rank, world)
    if dist.is_initialized():
        dist.barrier()
    launch_ms = 0.03 + 0.01 * rank  # stands in for this rank's event-timed launch duration
    red = sharding.all_reduce_metrics({"boards": float((hi - lo) * args.steps),
                                       "elapsed_s": 1.0 + rank, "launch_ms_max": launch_ms,
                                       "neg_launch_ms_min": -launch_ms})
    coll = sharding.collective_info()
    # the two other readings of N > 1 (main(): generator_block): BASELINE configs[3] -- 4096 keys
    # per rank of split(PRNGKey(0), 4096 N) at 32x32/20 -- and the strong reading -- the global
    # batch of 4096 keys cut into N shards
    blocks = {}
    for name, total in (("c4", BATCH_PER_GPU * world), ("strong", BATCH_PER_GPU)):
        blo, bhi = sharding.shard_bounds(total, rank, world)
        bred = sharding.all_reduce_metrics({"boards": float((bhi - blo) * args.steps),
                                            "elapsed_s": 1.0 + rank})
        blocks[name] = {"global_batch": total, "boards": bred["boards"],
                        "boards_per_gpu": bhi - blo,
                        "elapsed_s_max_over_ranks": bred["elapsed_s"],
                        "scaling": "weak" if name == "c4" else "strong"}
    if rank == 0:
        print(json.dumps({"metric": "boards_per_sec", "value": None, "unit": "boards/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "rehearsal": True, "boards": red["boards"], **blocks,
                          "elapsed_s_max_over_ranks": red["elapsed_s"], "scaling": "weak",
                          "launch_ms_over_ranks": {"min": -red["neg_launch_ms_min"],
                                                   "max": red["launch_ms_max"]},
                          "collective": coll}),
              flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(sample_batches: int):
    """Oracle (CPU port of the reference algorithm) on the same workload: all host threads and a
    single thread, plus the reference's own published case (10x10/5, batch 100) on this host."""
    import numpy as np  # noqa: F401

    from oracle import oracle as O

    threads = host_cpu_share()
    if sample_batches < 0:
        sample_batches = max(8, threads * 2)  # ~20 core-seconds at ~6k boards/s/core
    keys = O.split(O.prng_key(0), BATCH_PER_GPU * sample_batches)
    O.generate_batch(keys[:1024], GRID, AGENTS, threads=threads, want_solved=False)  # warm up
    t0 = time.perf_counter()
    O.generate_batch(keys, GRID, AGENTS, threads=threads, want_solved=False)  # ONE parallel region
    dt = time.perf_counter() - t0
    boards = sample_batches * BATCH_PER_GPU
    t0 = time.perf_counter()
    O.generate_batch(keys[:2 * BATCH_PER_GPU], GRID, AGENTS, threads=1, want_solved=False)
    dt1 = time.perf_counter() - t0
    # the reference's published case: 19.7 ms per vmapped batch of 100 boards at 10x10/5
    k100 = O.split(O.prng_key(0), 100)
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        O.generate_batch(k100, 10, 5, threads=threads, want_solved=False)
    ms100 = (time.perf_counter() - t0) / reps * 1e3
    t0 = time.perf_counter()
    for _ in range(reps):
        O.generate_batch(k100, 10, 5, threads=1, want_solved=False)
    ms100_1 = (time.perf_counter() - t0) / reps * 1e3
    return {"value": boards / dt, "unit": "boards/s", "cores": threads, "kind": "port",
            "single_thread": {"value": 2 * BATCH_PER_GPU / dt1, "unit": "boards/s", "cores": 1,
                              "sample": f"{2 * BATCH_PER_GPU} boards 16x16/10, {dt1:.2f} s"},
            "host_cpu": host_cpu_model(),
            "sample": f"{boards} boards 16x16/10 (keys split(PRNGKey(0), {boards})) in one "
                      f"OpenMP region, {dt:.2f} s wall on {threads} threads "
                      f"(~{dt * threads:.0f} core-seconds)",
            "reference_case_10x10_5_batch100": {
                "cpu_port_ms_per_batch": ms100, "cpu_port_ms_per_batch_single_thread": ms100_1,
                "published_ms_per_batch": 19.7,
                "published_source": "package_evaluation/profiling_generators.ipynb, vmapped "
                                    "jitted ParallelRandomWalkGenerator on JAX-CPU, hardware "
                                    "unspecified"},
            "note": "the port restates the reference's own formulation (N private grid copies "
                    "per agent step, max-join, correction masks, a full stable sort per board) in "
                    "plain C, because the JAX-CPU reference cannot run on this box; it is a "
                    "reported baseline, and the GPU/CPU ratio is NOT a speed-up claim about the "
                    "kernels -- the roofline fractions are"}


def timed(fn, reps):
    import torch

    # warm up for ~30 ms of GPU work (at least 3 calls): a row that follows a host-only stretch or
    # a tiny configuration otherwise times the clock ramp and first-use costs of its kernel (the
    # C2 row after the one-board C1 row read 83 instead of 32 us per launch)
    t_end = time.perf_counter() + 0.03
    calls = 0
    while calls < 3 or (time.perf_counter() < t_end and calls < 2000):
        fn()
        calls += 1
        if calls % 16 == 0:
            torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us per launch


def cpu_rate(fn, units):
    t0 = time.perf_counter()
    fn()
    return units / (time.perf_counter() - t0)


WALK_REFILL_EVERY = 20  # ParallelRandomWalkGenerator.native_spec().refill["every"]
GENERATION_TYPES = ("online_parallel_rw", "online_uniform", "online_seq_rw", "online_random_seed",
                    "offline_parallel_rw", "offline_seed_extension")


def env_steps_by_generator(keys, m, N, steps, close=None):
    """Connector.step for every generation_type of the reference's setup_train.py:112-152 (the
    policy inside the step launch, auto-reset on, each generator's default next-episode-stash
    schedule): env-steps/s per GPU.  Returns {generation_type: row}."""
    import torch

    from routing_board_generation_amd import ops
    from routing_board_generation_amd.interface.board_generator_interface import make_online_generator

    dev, B = keys.device, keys.shape[0]
    rows = {}
    for name in GENERATION_TYPES:
        kw = {}
        if name.startswith("offline"):  # cfg.env.seed_extension.* (connector.yaml), 10 000 boards
            kw = dict(randomness=0.0, two_sided=True, extension_iterations=1, extension_steps=1e23,
                      number_of_boards=10000)
        gen = make_online_generator(name, m, N, **kw).native_spec()
        sched = {"every": 16, "depth": 1, "serial": True, "lead": 2, "streams": 2, **gen.refill}
        st, ts = ops.alloc_state(B, m, N, dev), ops.alloc_timestep(B, m, N, dev)
        loop = ops.StepLoop(st, ts, keys, refill_every=sched["every"], fuse_policy=True, gen=gen,
                            stash_depth=sched["depth"], serial_refill=sched["serial"],
                            refill_lead=sched["lead"], refill_streams=sched["streams"])
        loop.reset(keys)
        # (rolling refills: the reset's own refill generates depth x B boards, and the pipeline of
        # refills in flight is full after lead + 1 epochs)
        warm = max(60, (2 if sched["serial"] else sched["lead"] + 3) * sched["every"])
        for t in range(warm):
            loop.step(t)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        if sched["every"]:
            loop.stash["hits"].zero_()
            loop.stash["misses"].zero_()
        t0 = time.perf_counter()
        e0.record()
        for t in range(warm, warm + steps):
            loop.step(t)
        e1.record()
        # host wall to the last STEP's completion (the stepping stream): the rolling generators'
        # refills still in flight on their own streams belong to later steps -- a device-wide
        # synchronize charged ~2 ms of them to 300 steps and read as "host-bound" in round 3
        e1.synchronize()
        wall = time.perf_counter() - t0
        t_drain = time.perf_counter()
        el = close(t0) if close else None
        torch.cuda.synchronize(dev)
        drain_ms = (time.perf_counter() - t_drain) * 1e3
        loop.drain()
        us = e0.elapsed_time(e1) / steps * 1e3
        rows[name] = {"us_per_step": us, "env_steps_per_s": B / us * 1e6,
                      "stash": ({"refill_every": sched["every"], "slots_per_env": sched["depth"],
                                 **({} if sched["serial"] else
                                    {"rolling": True, "lead_epochs": sched["lead"],
                                     "generator_streams": sched["streams"]}),
                                 "hits": int(loop.stash["hits"].sum().item()),
                                 "misses": int(loop.stash["misses"].sum().item())}
                                if sched["every"] else "none (the generator runs inside the step "
                                                       "launch of the envs that restart)")}
        rows[name]["host_wall_us_per_step"] = wall / steps * 1e6
        rows[name]["refills_in_flight_at_the_end_ms"] = drain_ms
    base = rows["online_parallel_rw"]["us_per_step"]
    for r in rows.values():
        r["vs_online_parallel_rw"] = r["us_per_step"] / base
    return rows



def all_configs():
    """Every BASELINE.json config on one GPU, each with the CPU port timed beside it (the
    cpu_baseline leg applied to the other configs; `--all-configs`, not the driver's line)."""
    import torch

    import routing_board_generation_amd as rbg
    from oracle import oracle as O
    from routing_board_generation_amd import ops

    threads = host_cpu_share()
    rows = [{"note": "HIP-event time per call of the allocating Python op (fresh output tensors "
                     "every call, 20-50 calls after warm-up): a few microseconds above the "
                     "pre-bound launches of the default line, and host-bound for the tiny configs"}]

    def gen_row(name, m, N, B, kind="prw"):
        keys = rbg.split(rbg.PRNGKey(0), B)
        if kind == "prw":
            f = lambda: ops.prw_generate(keys, m, N, split_first=True, want_solved=False, want_state=True)  # noqa
            cf = lambda: O.generate_batch(keys.cpu().numpy(), m, N, threads=threads, want_solved=False)  # noqa
        elif kind == "seq":
            f = lambda: ops.seq_rw_generate(keys, m, N, split_first=True, want_solved=False)  # noqa
            cf = lambda: O.seq_rw_generator_batch(keys.cpu().numpy(), m, N, threads=threads)  # noqa
        elif kind == "uniform":
            f = lambda: ops.uniform_generate(keys, m, N)  # noqa
            cf = lambda: O.generate_batch(keys.cpu().numpy(), m, N, "uniform", threads=threads, want_solved=False)  # noqa
        elif kind == "philox":  # the opt-in Philox stream of the walk (not the reference's process)
            f = lambda: ops.prw_generate_philox(keys, m, N)  # noqa
            cf = lambda: O.prw_philox_batch(keys.cpu().numpy(), m, N, threads=threads)  # noqa
        else:  # SeedExtensionGenerator.__call__ (reference defaults)
            f = lambda: ops.seed_extension_generate(keys, m, N, split_first=True)  # noqa
            cf = lambda: O.seed_extension_batch(keys.cpu().numpy(), m, N, threads=threads)  # noqa
        us = timed(f, 20 if B >= 1024 else 50)
        cf()
        rows.append({"config": name, "grid": m, "agents": N, "batch": B, "us_per_launch": us,
                     "boards_per_s": B / us * 1e6, "cpu_port_boards_per_s": cpu_rate(cf, B),
                     "cpu_threads": threads})

    gen_row("C1 ParallelRandomWalk 8x8/5 B=1", 8, 5, 1)
    gen_row("C2 ParallelRandomWalk 16x16/10 B=4096", 16, 10, 4096)
    gen_row("C4 ParallelRandomWalk 32x32/20 B=4096 (one GPU's shard of 32768)", 32, 20, 4096)
    gen_row("C4 ParallelRandomWalk 32x32/20 B=32768 (the global batch in one launch on one GPU)",
            32, 20, 32768)
    gen_row("C5 SequentialRandomWalk 16x16/10 B=4096", 16, 10, 4096, kind="seq")
    gen_row("C2 shape on the opt-in Philox stream 16x16/10 B=4096", 16, 10, 4096, kind="philox")
    gen_row("C2 shape on the opt-in Philox stream 16x16/10 B=65536", 16, 10, 65536, kind="philox")
    gen_row("C2 ParallelRandomWalk 16x16/10 B=65536", 16, 10, 65536)
    gen_row("C5 SeedExtension 16x16/10 B=4096", 16, 10, 4096, kind="seed_ext")
    gen_row("ref SeedExtension 10x10/5 B=100 (notebook batch: 427 ms on JAX-CPU)", 10, 5, 100,
            kind="seed_ext")
    gen_row("ref ParallelRandomWalk 10x10/5 B=100 (notebook batch: 19.7 ms on JAX-CPU)", 10, 5, 100)
    # the rest of the reference's published timings (profiling_generators.ipynb):
    # hardware unspecified, quoted for context only -- a single board is host-bound here
    gen_row("ref ParallelRandomWalk 10x10/5 B=1 (notebook: 255 us on JAX-CPU)", 10, 5, 1)
    gen_row("ref UniformRandomGenerator 10x10/5 B=100 (notebook batch: 4.18 ms)", 10, 5, 100,
            kind="uniform")
    gen_row("ref UniformRandomGenerator 10x10/5 B=1 (notebook: 213 us)", 10, 5, 1, kind="uniform")
    gen_row("ref SequentialRandomWalk 10x10/5 B=100 (notebook cell failed; plot script: 515 ms)",
            10, 5, 100, kind="seq")
    gen_row("ref SequentialRandomWalk 10x10/5 B=1 (notebook: 685 us)", 10, 5, 1, kind="seq")
    gen_row("ref SeedExtension 10x10/5 B=1 (notebook: 998 us)", 10, 5, 1, kind="seed_ext")

    # C3: env.step rollout, per-step API and fused
    B, m, N = 4096, 16, 10
    keys = rbg.split(rbg.PRNGKey(0), B)
    st, ts = ops.connector_reset(keys, m, N)
    act = torch.empty((B, N), dtype=torch.int32, device="cuda")
    t = [0]

    def step():
        ops.random_actions(keys, t[0], ts["action_mask"], out=act)
        ops.connector_step(st, act, auto_reset=True, state_out=st, timestep=ts)
        t[0] += 1

    us = timed(step, 100)
    rows.append({"config": "C3 Connector.step rollout 16x16/10 B=4096 (per-step API, obs written)",
                 "us_per_step": us, "env_steps_per_s": B / us * 1e6})
    stats = {k: torch.empty((B,), dtype=torch.int32, device="cuda") for k in ops.ROLLOUT_STATS}
    us = timed(lambda: ops.connector_rollout(st, keys, 0, 50, state_out=st, stats=stats), 10)
    rows.append({"config": "C3 Connector.step rollout 16x16/10 B=4096 (fused rbg_connector_rollout, 50 steps/launch)",
                 "us_per_launch": us, "env_steps_per_s": B * 50 / us * 1e6})
    est, _ = O.connector_reset_batch(keys.cpu().numpy(), m, N, threads=threads)
    mask = O.connector_observe_batch(est, threads=threads)["action_mask"]

    def cpu_steps():
        m_ = mask
        for s in range(10):
            a = O.random_actions_batch(keys.cpu().numpy(), s, m_)
            m_ = O.connector_step_batch(est, a, auto_reset=True, threads=threads)["action_mask"]

    rows.append({"config": "C3 CPU port (oracle, per-step)", "cpu_threads": threads,
                 "env_steps_per_s": cpu_rate(cpu_steps, B * 10)})
    # C3's shape for every generator the reference couples to Connector (setup_train.py:112-152)
    for name, r in env_steps_by_generator(keys, m, N, 480).items():
        rows.append({"config": f"C3 Connector.step 16x16/10 B=4096, generation_type={name} "
                               "(policy inside the step launch, obs written, auto-reset)", **r})
    print(json.dumps(rows, indent=1))



def pure_stream(rbg, ops, torch, dev, m, N, sb, Bs, copy_gbs=None):
    """connector_step_kernel as a pure stream: NOOP actions (no env terminates), a batch large
    enough to leave the Infinity Cache -- how close the kernel gets to the HBM roofline when
    neither resets nor the launch shape limit it."""
    skeys = rbg.split(rbg.PRNGKey(1, device=dev), Bs)
    sst, sts = ops.connector_reset(skeys, m, N)
    sso = ops.alloc_state(Bs, m, N, dev)
    noop = torch.zeros((Bs, N), dtype=torch.int32, device=dev)
    for _ in range(3):
        ops.connector_step(sst, noop, time_limit=1 << 30, state_out=sso, timestep=sts)
    pe0, pe1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    pe0.record()
    for _ in range(20):
        ops.connector_step(sst, noop, time_limit=1 << 30, state_out=sso, timestep=sts)
    pe1.record()
    torch.cuda.synchronize(dev)
    p_ms = pe0.elapsed_time(pe1) / 20
    p_ach = Bs * sb / (p_ms * 1e-3) / 1e9
    peaks = hbm_peaks(dev) if copy_gbs is None else None
    if peaks:
        copy_gbs = max(peaks["copy"], peaks["copy_nontemporal"])
    return {"batch": Bs, "launches": 23, "kernel_avg_ms": p_ms, "achieved": p_ach, "unit": "GB/s",
            "peak": HBM_PEAK_GBS, "frac": p_ach / HBM_PEAK_GBS,
            "peak_measured_copy": copy_gbs, "frac_of_measured_copy": p_ach / copy_gbs,
            **({"peak_measured": peaks} if peaks else {}),
            "algorithmic_bytes_per_launch": Bs * sb,
            "traffic": measured_traffic("connector_step_kernel", Bs,
                                        "_stream262k" if Bs == 262144 else "_stream", Bs),
            "working_set_bytes": Bs * sb,
            "working_set_over_infinity_cache": Bs * sb / (256 * 2 ** 20),
            "note": "connector_step_kernel, NOOP actions (no terminations), working set "
                    f"{Bs * sb / 1e6:.0f} MB = {Bs * sb / (256 * 2 ** 20):.1f}x the 256 MiB "
                    "Infinity Cache; 92 % of its bytes are write-once non-temporal stores, which "
                    "is why it can exceed the measured COPY rate (a copy reads as many bytes as "
                    "it writes); `bench.py --mode stream [--stream-batch B]` runs only this and "
                    "its WRITE_SIZE pass is in profiles/"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--settle", type=float, default=0.3,
                    help="seconds of untimed launches before the warm-up steps (device clocks "
                         "and caches at steady state); 0 = none")
    ap.add_argument("--env-steps", type=int, default=300,
                    help="env steps of the secondary Connector rollout measurement (0 = skip)")
    ap.add_argument("--rehearse-single-gpu", action="store_true",
                    help="developer option: run all ranks on GPU 0 over gloo to rehearse the "
                         "multi-rank path on a one-GPU box (numbers are meaningless)")
    ap.add_argument("--force-process-group", action="store_true",
                    help="developer option: join a process group even with one rank, so that a "
                         "one-GPU box exercises RCCL init / barrier / all-reduce")
    ap.add_argument("--stream-batch", type=int, default=65536,
                    help="batch of the pure-step HBM streaming check (0 = skip)")
    ap.add_argument("--stream-batch-large", type=int, default=262144,
                    help="second, larger batch of that check in the default run: a working set "
                         "13x the Infinity Cache (3.4 GB), so that nothing of it can be "
                         "absorbed on chip between launches (0 = skip)")
    ap.add_argument("--rollout-steps", type=int, default=50,
                    help="env steps per launch of the fused random-policy rollout (0 = skip)")
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="batches of 4096 boards timed for cpu_baseline (0 = skip; default: "
                         "threads/4, i.e. ~20 core-seconds of CPU work)")
    ap.add_argument("--pipelined", action="store_true",
                    help="also time the K generator launches with three batches in flight on "
                         "three streams (off by default so that the rocprofv3 kernel statistics "
                         "of the default command describe lone launches only)")
    ap.add_argument("--mode", choices=("all", "generator", "step-replay", "stream", "record"),
                    default="all",
                    help="all = the driver's line; generator = the headline only; step-replay / "
                         "stream = ONE launch population of connector_step_kernel each, record = "
                         "the recording rollout only (the rocprofv3 passes under profiles/ are "
                         "taken from these)")
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="developer option: CPU-only rehearsal of the multi-rank launch (gloo)")
    ap.add_argument("--all-configs", action="store_true",
                    help="measure every BASELINE.json config (GPU + CPU port) and print a JSON "
                         "list instead of the driver's line (profiles/*_configs.json)")
    args = ap.parse_args()
    if args.all_configs:
        all_configs()
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: become the parent of the one-process-per-GPU job
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    if args.rehearse_launch:
        rehearse_launch(args)
        return
    if args.mode == "generator":
        args.env_steps = args.rollout_steps = 0
    elif args.mode in ("step-replay", "stream"):
        args.steps, args.warmup, args.rollout_steps, args.cpu_sample = 1, 0, 0, 0
    elif args.mode == "record":
        args.steps, args.warmup, args.env_steps, args.cpu_sample = 1, 0, 0, 0

    import torch
    import torch.distributed as dist

    import routing_board_generation_amd as rbg
    from routing_board_generation_amd import ops, sharding

    rank, local_rank, world = sharding.init_distributed(
        backend="gloo" if args.rehearse_single_gpu else None,
        single_device=args.rehearse_single_gpu, force=args.force_process_group)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}: launch with "
                         "torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU path to time")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    rbg._native.lib()

    def barrier():
        torch.cuda.synchronize(dev)
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize(dev)

    def close(t_start):
        """End of a timed region.  The clock is read when THIS rank's work is complete (a local
        synchronize); the barrier that keeps the ranks in step comes after the reading, so no
        collective sits inside anybody's interval.  The per-rank times are MAX-reduced later
        (sharding.all_reduce_metrics): the job is as slow as its slowest rank."""
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t_start
        if dist.is_initialized():
            dist.barrier()
        return el

    collective = sharding.collective_info(dev)

    if dist.is_initialized():  # the first collective pays communicator setup: keep it out of the timed region
        sharding.all_reduce_metrics({"boards": 0.0}, device=dev)

    # ---------------------------------------------------------------- inputs resident in HBM
    all_keys = rbg.split(rbg.PRNGKey(0, device=dev), BATCH_PER_GPU * world)
    keys = sharding.shard_keys(all_keys, rank, world)
    B, m, N = keys.shape[0], GRID, AGENTS
    out = ops.prw_generate(keys, m, N, split_first=True, want_solved=False, want_train=True,
                           want_state=True, want_iters=True)

    # one step = one launch of the generator into the same output buffers; the launch is bound
    # once (ops.prw_generate_launcher) because the per-call tensor checks of ops.prw_generate cost
    # the host as much as the kernel takes, and a short run would time Python instead of the GPU
    step = ops.prw_generate_launcher(keys, m, N, True, out)

    # Settle the device before the W warm-up steps: a fresh process's first launches run at the
    # idle clock and pay first-use costs (code object upload, caches), which a short run (the
    # driver uses 20 steps after 5 warm-ups) would otherwise time instead of the steady state:
    # 38.4 instead of 35.7 us per launch.  Untimed, like the warm-up steps themselves.
    settle_until = time.perf_counter() + args.settle
    while time.perf_counter() < settle_until:
        for _ in range(50):
            step()
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    # ---------------------------------------------------------------- timed region
    # ONE HIP-event pair brackets the K back-to-back launches (on torch's current stream, the
    # stream the kernel is launched on): a pair per launch costs ~12 us of event latency on this
    # stack, comparable to the kernel itself.  Average launch duration = event span / K; it
    # includes the ~1-2 us dependent-launch gap, so it can only overstate the kernel time.
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()  # torch creates a HIP event at its first record(): ~85 us each, which must not
    ev1.record()  # land in a timed region of 20 steps (0.7 ms)
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    elapsed = close(t0)
    kernel_avg_ms = ev0.elapsed_time(ev1) / args.steps

    iters = out["iters"]
    metrics = sharding.all_reduce_metrics({
        "boards": float(B * args.steps), "iters": float(iters.sum().item()),
        "max_iters": float(iters.max().item()), "elapsed_s": elapsed,
        "launch_ms_max": kernel_avg_ms, "neg_launch_ms_min": -kernel_avg_ms}, device=dev)
    total_elapsed = metrics["elapsed_s"]  # max over ranks
    value = metrics["boards"] / total_elapsed

    # ---------------------------------------------------------------- the other readings of N > 1
    # `value` is the WEAK reading of BASELINE's "batch 4096, 1/2/4/8 GPU" (4096 boards per GPU).
    # Beside it, timed exactly like it (barrier, local clock, K launches, local synchronize,
    # MAX over ranks): BASELINE configs[3] and the strong reading of the same batch.
    def generator_block(global_keys, gm, gn, steps, label):
        shard = sharding.shard_keys(rbg.split(rbg.PRNGKey(0, device=dev), global_keys), rank, world)
        o = ops.prw_generate(shard, gm, gn, split_first=True, want_solved=False, want_train=True,
                             want_state=True, want_iters=True)
        launch = ops.prw_generate_launcher(shard, gm, gn, True, o)
        for _ in range(max(3, min(args.warmup, 20))):
            launch()
        b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b0.record()
        b1.record()
        barrier()
        tb = time.perf_counter()
        b0.record()
        for _ in range(steps):
            launch()
        b1.record()
        el = close(tb)
        ms = b0.elapsed_time(b1) / steps
        red = sharding.all_reduce_metrics({"boards": float(shard.shape[0] * steps), "elapsed_s": el,
                                           "launch_ms_max": ms, "neg_launch_ms_min": -ms},
                                          device=dev)
        return {"workload": label, "grid_size": gm, "num_agents": gn,
                "global_batch": int(global_keys), "boards_per_gpu": int(shard.shape[0]),
                "keys": f"split(PRNGKey(0), {global_keys}), rank g takes shard g "
                        "(sharding.shard_keys = reset_keys.reshape(devices, per_device, 2), "
                        "setup_train.py:396-400)",
                "value": red["boards"] / red["elapsed_s"], "unit": "boards/s", "steps": steps,
                "ms_per_step": red["elapsed_s"] / steps * 1e3,
                "launch_ms_over_ranks": {"min": -red["neg_launch_ms_min"],
                                         "max": red["launch_ms_max"]}}

    scaling_blocks = None
    if args.mode in ("all", "generator"):
        scaling_blocks = {
            "c4": {**generator_block(4096 * world, 32, 20, max(1, min(args.steps, 100)),
                                     "ParallelRandomWalkGenerator 32x32 grid, 20 agents, 4096 "
                                     "keys per GPU (BASELINE configs[3]: at 8 GPUs the 32 768-key "
                                     "batch, batch-sharded)"),
                   "scaling": "weak"},
            "strong": {**generator_block(BATCH_PER_GPU, GRID, AGENTS, args.steps,
                                         "ParallelRandomWalkGenerator 16x16 grid, 10 agents, "
                                         "GLOBAL batch 4096 cut into one shard per GPU "
                                         "(512 boards per GPU at 8 GPUs)"),
                       "scaling": "strong"},
            "note": "`value` of the line is the weak reading of BASELINE's \"batch 4096, 1/2/4/8 "
                    "GPU\" (4096 boards per GPU, global batch 4096 x N); `strong` is the other "
                    "reading (global batch 4096 whatever N; a launch of 512 boards lasts as long "
                    "as its slowest board, so this reading cannot scale and says so); `c4` is the "
                    "one BASELINE config that is multi-GPU by definition.  Same timing rule as "
                    "`value`; per GPU both generate disjoint shards with no data-path collective."}

    # What actually bounds the launch: the serial latency of its longest board.  Timed live: the
    # board with the most walk iterations launched ALONE (one wave), same kernel, same outputs.
    longest = int(iters.argmax().item())
    lkeys = keys[longest:longest + 1].clone()
    lout = ops.prw_generate(lkeys, m, N, split_first=True, want_solved=False, want_train=True,
                            want_state=True, want_iters=True)
    lstep = ops.prw_generate_launcher(lkeys, m, N, True, lout)
    for _ in range(20):
        lstep()
    le0, le1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    le0.record()
    for _ in range(200):
        lstep()
    le1.record()
    torch.cuda.synchronize(dev)
    longest_ms = le0.elapsed_time(le1) / 200
    serial = {"bound": "longest_board_latency", "floor_ms": longest_ms,
              "frac": longest_ms / kernel_avg_ms, "walk_iters_of_that_board": int(iters[longest]),
              "note": "the board with the most walk iterations launched alone (one wavefront; 200 "
                      "launches, HIP events): a board is a chain of dependent iterations, so no "
                      "launch of this batch can be shorter; frac = that floor / this launch"}

    bpb = generator_bytes_per_board(m, N)
    achieved = B * bpb / (kernel_avg_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm", "kernel": "prw_generate_kernel<32> (2 boards per wavefront)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic("prw_generate_kernel", B),
                "algorithmic_bytes_per_launch": B * bpb,
                "bytes_per_board": bpb, "boards_per_launch": B,
                "kernel_avg_ms": kernel_avg_ms,
                "issue": issue_roofline("prw_generate_kernel", B, kernel_avg_ms),
                "serial": serial,
                "note": "latency-bound walk (serial dependent iterations per board): HBM fraction "
                        "reported for the record, the issue-based figure beside it; see DESIGN.md"}

    # the opt-in Philox stream of the same walk (north_star / SURVEY 8a-PRNG): a different random
    # process (not the reference's boards), so it is reported beside `value`, never as it
    philox = None
    if args.mode in ("all", "generator"):
        p_out = ops.prw_generate_philox(keys, m, N)
        p_steps = min(args.steps, 300)
        for _ in range(5):
            ops.prw_generate_philox(keys, m, N, out=p_out)
        pe0, pe1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        pe0.record()
        for _ in range(p_steps):
            ops.prw_generate_philox(keys, m, N, out=p_out)
        pe1.record()
        close(time.perf_counter())
        p_ms = pe0.elapsed_time(pe1) / p_steps
        philox = {"value_per_gpu": B / (p_ms * 1e-3), "unit": "boards/s", "ms_per_step": p_ms,
                  "steps": p_steps,
                  "note": "rbg_prw_generate_philox: the same walk on a Philox-4x32-10 stream "
                          "(project-defined, include/rbg_prw.h; its only checker is this "
                          "project's CPU restatement). One Philox evaluation per agent every "
                          "four iterations, no key chain: ~100 instructions per walk iteration "
                          "instead of 156"}
        del p_out

    pipelined = None
    if args.pipelined:
        # The same K launches with three batches in flight (three streams, three output sets): a lone
        # launch of 4096 boards ends on the serial walk of its longest board while most SIMDs idle,
        # so independent batches overlap almost for free.  Reported next to `value`, never as it.
        n_streams = 3  # + the default stream: within the 4 hardware queues HIP multiplexes streams onto
        p_streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
        p_outs = [ops.prw_generate(keys, m, N, split_first=True, want_solved=False, want_train=True,
                                   want_state=True, want_iters=True) for _ in range(n_streams)]

        import ctypes as C
        p_lib = rbg._native.lib()
        p_args = [(keys.data_ptr(), B, m, N, 1, None, o["train_grid"].data_ptr(), o["start"].data_ptr(),
                   o["target"].data_ptr(), o["position"].data_ptr(), o["step_count"].data_ptr(),
                   o["key_out"].data_ptr(), o["iters"].data_ptr(), C.c_void_p(st.cuda_stream))
                  for o, st in zip(p_outs, p_streams)]  # the launch of ops.prw_generate, stream given

        def run_pipelined(k):
            for i in range(k):
                rc = p_lib.rbg_prw_generate(*p_args[i % n_streams])
                if rc:
                    rbg._native.check(rc, "rbg_prw_generate")

        run_pipelined(args.warmup)
        barrier()
        t0 = time.perf_counter()
        run_pipelined(args.steps)
        pm = sharding.all_reduce_metrics({"boards": float(B * args.steps),
                                          "elapsed_s": close(t0)}, device=dev)
        pipelined = {"streams": n_streams, "value": pm["boards"] / pm["elapsed_s"], "unit": "boards/s",
                     "ms_per_step": pm["elapsed_s"] / args.steps * 1e3,
                     "note": "the same steps with three batches of 4096 in flight on three HIP streams "
                             "(distinct output buffers); about the rate of one large batch"}
        del p_outs

    # ---------------------------------------------------------------- secondary: env.step rollout
    # BASELINE configs[2] through the per-step API: every step = masked-uniform policy kernel +
    # connector_kernel (state in/out + full observation through HBM), auto-reset on.  The loop
    # reuses its buffers and prebuilt launch descriptors (ops.StepLoop).  Four variants:
    #   prefetch     next-episode States generated ahead of time, one refill per 20 steps (rbg_stash),
    #                so a step launch never waits for a random walk           -> `value`
    #   inline       the walk of every finished env runs inside the step launch
    #   fused_policy prefetch + policy and step in one launch (rbg_connector_rollout, 1 step)
    #   no_auto_reset  auto-reset off: finished envs stay terminal, no walk ever runs
    env_step = None
    peaks = hbm_peaks(dev) if (rank == 0 and args.mode != "generator") else None
    copy_gbs = max(peaks["copy"], peaks["copy_nontemporal"]) if peaks else None
    if args.env_steps > 0:
        sb = env_step_bytes(m, N)

        def run_loop(refill_every, fuse, auto_reset=True, groups=1):
            st = ops.alloc_state(B, m, N, dev)
            ts = ops.alloc_timestep(B, m, N, dev)
            if groups > 1:  # half-batches on streams of their own (ops.GroupedStepLoop)
                loop = ops.GroupedStepLoop(st, ts, keys, groups=groups, refill_every=refill_every,
                                           fuse_policy=fuse, auto_reset=auto_reset)
            else:
                loop = ops.StepLoop(st, ts, keys, refill_every=refill_every, fuse_policy=fuse,
                                    auto_reset=auto_reset)
            loop.reset(keys)
            for t in range(30):
                loop.step(t)
            se0, se1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            barrier()
            t1 = time.perf_counter()
            se0.record()
            for t in range(30, 30 + args.env_steps):
                loop.step(t)
            if groups > 1:
                loop.join()
            se1.record()
            el = close(t1)
            loop.drain()
            em = sharding.all_reduce_metrics({"env_steps": float(B * args.env_steps),
                                              "elapsed_s": el}, device=dev)
            return em["env_steps"] / em["elapsed_s"], se0.elapsed_time(se1) / args.env_steps

        def run_api(prefetch, donate, single_agent=False):
            """The same loop through the drop-in classes (State/TimeStep pytrees per step)."""
            env = rbg.Connector(generator=rbg.ParallelRandomWalkGenerator(m, N), donate=donate)
            if single_agent:  # the reference's training env (setup_train.py:158-167)
                env = rbg.MultiToSingleWrapper(env)
            env = rbg.VmapAutoResetWrapper(env, prefetch=prefetch)
            state, ts = env.reset(keys)
            policy = ops.RandomPolicy(keys, N)  # the acting side of the loop: one launch per step
            for t in range(30):
                state, ts = env.step(state, policy(t, ts.observation.action_mask))
            barrier()
            t1 = time.perf_counter()
            for t in range(30, 30 + args.env_steps):
                state, ts = env.step(state, policy(t, ts.observation.action_mask))
            el = close(t1)
            for sess in env.unwrapped._sessions.values():
                sess.drain()
            em = sharding.all_reduce_metrics({"env_steps": float(B * args.env_steps),
                                              "elapsed_s": el}, device=dev)
            return em["env_steps"] / em["elapsed_s"]

        def step_kernel_ms(refill_every):
            """Average duration of a connector_step_kernel launch alone: the same trajectory
            replayed with the actions of a recorded pass, so the timed stream holds step launches
            only.  HIP-event pairs bracket every run of consecutive step launches BETWEEN two
            refill launches (the refill of an epoch is queued after its last step), so the refills
            are outside the timed spans; the dependent-launch gaps inside a run are inside.
            Returns (ms per step launch, ms per step with the refills amortised in)."""
            n = args.env_steps
            # the actions of the pass to replay come from ONE recording rollout launch (the same
            # policy, bit-identical trajectory: tests), so that every connector_step_kernel launch
            # of a `--mode step-replay` run belongs to the timed population -- round 3 recorded them
            # with n more step launches, which a rocprofv3 kernel summary of that run averaged in
            # (interleaved with policy launches they take ~1.5 us longer)
            warm = 20  # replayed but untimed: first-use costs of the launch, clocks
            st0, _ = ops.connector_reset(keys, m, N)
            _, _, traj = ops.connector_rollout_record(st0, keys, 0, n + warm, with_obs_grid=False)
            rec = traj["action"]
            del st0
            st = ops.alloc_state(B, m, N, dev)
            ts = ops.alloc_timestep(B, m, N, dev)
            loop = ops.StepLoop(st, ts, keys, refill_every=refill_every)
            loop.reset(keys)
            for t in range(warm):
                loop.step(t, action=rec[t])
            ep, spans = loop._epochs, []
            queue_refill = ep.after_launch

            def after_launch(stream=None):
                due = ep.tick == ep._refill_after
                if due:
                    spans[-1][1].record()  # the run ends before the refill is queued
                queue_refill(stream)
                if due:
                    spans.append([torch.cuda.Event(enable_timing=True),
                                  torch.cuda.Event(enable_timing=True)])
                    spans[-1][0].record()

            ep.after_launch = after_launch
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            spans.append([torch.cuda.Event(enable_timing=True),
                          torch.cuda.Event(enable_timing=True)])
            torch.cuda.synchronize(dev)
            e0.record()
            spans[-1][0].record()
            for t in range(warm, warm + n):
                loop.step(t, action=rec[t])
            e1.record()
            spans[-1][1].record()  # closes the last (possibly empty) run
            torch.cuda.synchronize(dev)
            loop.drain()
            in_runs = sum(a.elapsed_time(b) for a, b in spans)
            return in_runs / n, e0.elapsed_time(e1) / n

        if args.mode == "step-replay":
            k_step, k_span = step_kernel_ms(WALK_REFILL_EVERY)
            ach = B * sb / (k_step * 1e-3) / 1e9
            if rank == 0:
                print(json.dumps({"mode": "step-replay", "kernel": "connector_step_kernel",
                                  "launches": args.env_steps + 20, "batch": B,
                                  "kernel_avg_ms": k_step, "span_with_refills_ms": k_span,
                                  "achieved": ach, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                  "bytes_per_env_step": sb}), flush=True)
            return
        if args.mode == "stream":
            print(json.dumps({"mode": "stream", "kernel": "connector_step_kernel",
                              **pure_stream(rbg, ops, torch, dev, m, N, sb, args.stream_batch,
                                            copy_gbs), "peak_measured": peaks}), flush=True)
            return
        v_pref, k_pref = run_loop(WALK_REFILL_EVERY, False)
        k_step, k_span = step_kernel_ms(WALK_REFILL_EVERY)
        v_inl, k_inl = run_loop(0, False)
        v_fus, k_fus = run_loop(WALK_REFILL_EVERY, True)
        v_grp, k_grp = run_loop(WALK_REFILL_EVERY, True, groups=2)
        v_off, k_off = run_loop(0, False, auto_reset=False)
        api = {"functional": run_api(False, False), "functional_prefetch": run_api(True, False),
               "donate_prefetch": run_api(True, True),
               "donate_prefetch_multi_to_single": run_api(True, True, single_agent=True),
               "note": "VmapAutoResetWrapper(Connector(ParallelRandomWalkGenerator)).step with "
                       "State/TimeStep pytrees; functional = fresh output tensors every step, "
                       "donate = State advanced in place + two alternating TimeStep buffers"}
        ach = B * sb / (k_step * 1e-3) / 1e9
        by_gen = env_steps_by_generator(keys, m, N, max(96, min(args.env_steps, 480)), close)
        best = "prefetch_fused_policy" if v_fus > v_pref else "prefetch"
        env_step = {"metric": "env_steps_per_sec", "value": max(v_pref, v_fus),
                    "value_variant": best, "unit": "env-steps/s",
                    "parity": "unpinned (Connector.step semantics modelled on jumanji 0.2.2; no "
                              "reference fixture exists for them)",
                    "steps": args.env_steps, "auto_reset": True,
                    "policy": "masked uniform (rbg_random_actions): its own launch per step "
                              "(prefetch, inline_reset, no_auto_reset) or computed inside the "
                              "step launch from the state's own mask (prefetch_fused_policy: "
                              "rbg_connector_step_gen with policy_keys, bit-identical)",
                    "variants": {"prefetch": v_pref, "inline_reset": v_inl,
                                 "prefetch_fused_policy": v_fus,
                                 "prefetch_fused_policy_2groups": v_grp, "no_auto_reset": v_off},
                    "ms_per_step": {"prefetch": k_pref, "inline_reset": k_inl,
                                    "prefetch_fused_policy": k_fus,
                                    "prefetch_fused_policy_2groups": k_grp,
                                    "no_auto_reset": k_off},
                    "two_groups_note": "prefetch_fused_policy_2groups = the batch as two groups "
                                       "of 2048 envs on two HIP streams (ops.GroupedStepLoop): "
                                       "bit-identical results; reported beside `value`, which "
                                       "stays the single-stream figure",
                    "wrapper_api": api,
                    "by_generation_type": {
                        **by_gen,
                        "note": "the same step for every generator the reference couples to "
                                "Connector (setup_train.py:112-152), policy inside the step "
                                "launch: a finished episode takes its successor from the "
                                "generator's own next-episode stash (or, for Uniform / the "
                                "offline dataset, draws it inline -- cheaper than a row copy); "
                                "no env is ever regenerated that did not finish.  Per GPU "
                                "(rank 0's events)."},
                    "roofline": {"bound": "hbm", "kernel": "connector_step_kernel",
                                 "achieved": ach,
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                 "peak_measured_copy": copy_gbs,
                                 "frac_of_measured_copy": ach / copy_gbs if copy_gbs else None,
                                 "traffic": measured_traffic("connector_step_kernel", B, "_step"),
                                 "algorithmic_bytes_per_launch": B * sb,
                                 "bytes_per_env_step": sb, "kernel_avg_ms": k_step,
                                 "span_with_refills_ms": k_span,
                                 "parity": "unpinned (Connector.step semantics modelled on "
                                           "jumanji 0.2.2; no reference fixture, DESIGN.md 2)",
                                 "note": "average duration of a connector_step_kernel launch of "
                                         "the prefetch variant: the rollout replayed with "
                                         "recorded actions, HIP-event pairs around every run of "
                                         "step launches between two refill launches (dependent-"
                                         "launch gaps included, refills excluded; `bench.py "
                                         "--mode step-replay` runs only this, and its rocprofv3 "
                                         "kernel stats are in profiles/); span_with_refills_ms "
                                         "amortises the one refill per 20 steps in; the whole "
                                         "step incl. the policy launch is ms_per_step"}}
        # the same kernel as a pure stream (no env terminates, batch large enough to leave the
        # Infinity Cache): how close connector_kernel gets to the HBM roofline when neither resets
        # nor the host's launch rate limit it
        if rank == 0 and args.stream_batch > 0:
            small = pure_stream(rbg, ops, torch, dev, m, N, sb, args.stream_batch, copy_gbs)
            env_step["pure_step_stream"] = small
            if args.stream_batch_large > args.stream_batch:
                # the HBM figure proper: FETCH/WRITE_SIZE count Infinity-Cache hits too, and 256 MiB
                # of an 851 MB working set can stay on chip from launch to launch; at 13x the
                # cache the rate below is what HBM itself sustains, and `frac` is quoted on it
                torch.cuda.empty_cache()
                large = pure_stream(rbg, ops, torch, dev, m, N, sb, args.stream_batch_large,
                                    copy_gbs)
                torch.cuda.empty_cache()
                env_step["pure_step_stream"] = {
                    **large, "which_is_hbm": "this row (working set 13x the Infinity Cache); "
                                             "smaller_working_set may include MALL absorption "
                                             "(3.3x the cache: up to 30 % of it can stay on chip)",
                    "smaller_working_set": small}

    # ---------------------------------------------------------------- secondary: fused rollout
    def recorded_rollout(launches):
        """The random rollout RECORDING the trajectory (rbg_connector_rollout_record): every
        step's observation, mask, reward, discount, step_type, extras and action land in
        [T, B, ...] buffers -- the per-step API's output without its per-step launches."""
        T_rec = args.rollout_steps
        rec_bytes = 4 * N * m * m + 5 * N + 8 * N + 1 + 4 + 12 + 4 * N  # written per env-step
        st, _ = ops.connector_reset(keys, m, N)
        traj = ops.alloc_trajectory(T_rec, B, m, N, dev)
        for i in range(2):
            ops.connector_rollout_record(st, keys, i * T_rec, T_rec, state_out=st, trajectory=traj)
        ce0, ce1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        t3 = time.perf_counter()
        ce0.record()
        for i in range(launches):
            ops.connector_rollout_record(st, keys, (i + 2) * T_rec, T_rec, state_out=st,
                                         trajectory=traj)
        ce1.record()
        c_elapsed = close(t3)
        cm = sharding.all_reduce_metrics({"env_steps": float(B * T_rec * launches),
                                          "elapsed_s": c_elapsed}, device=dev)
        rec_ms = ce0.elapsed_time(ce1) / launches
        rec_gbs = B * T_rec * rec_bytes / (rec_ms * 1e-3) / 1e9
        return {
            "value": cm["env_steps"] / cm["elapsed_s"], "unit": "env-steps/s",
            "launch_avg_ms": rec_ms, "steps_per_launch": T_rec, "launches": launches,
            "roofline": {"bound": "hbm", "kernel": "connector_rollout_kernel<record>",
                         "achieved": rec_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": rec_gbs / HBM_PEAK_GBS,
                         "traffic": (measured_traffic("connector_rollout_kernel", B, "_record")
                                     if T_rec == 50 else None),
                         "algorithmic_bytes_per_launch": B * T_rec * rec_bytes,
                         "bytes_per_env_step": rec_bytes},
            "note": "rbg_connector_rollout_record: the rollout above with the timestep of EVERY "
                    "step (observation, mask, reward, discount, step_type, extras) and the "
                    "actions written to [steps, B, ...] buffers -- what the per-step API returns, "
                    "one launch per steps_per_launch steps; resets walk inside the launch; "
                    "`bench.py --mode record` runs only this (its rocprofv3 passes: "
                    "profiles/*_record_*)"}

    if args.mode == "record":
        if rank == 0:
            print(json.dumps({"mode": "record", **recorded_rollout(10)}), flush=True)
        return

    rollout = None
    if args.rollout_steps > 0:
        launches = 10

        def run_rollout(use_stash):
            stats = {k: torch.empty((B,), dtype=torch.int32, device=dev) for k in ops.ROLLOUT_STATS}
            ep = None
            if use_stash:  # two slots per env, refilled behind every launch (EpochStash)
                ep = ops.EpochStash(B, m, N, dev, every=1, serial=True, depth=2)
                ep.before_launch()
                st, _ = ops.connector_reset_prefetch(keys, m, N, ep.stash, ep.tick,
                                                     ep.ready_delay())
                ep.after_launch()
            else:
                st, _ = ops.connector_reset(keys, m, N)
            for i in range(2):  # warm-up (the stash serves resets from the second launch on)
                ops.connector_rollout(st, keys, i * args.rollout_steps, args.rollout_steps,
                                      state_out=st, stats=stats, epochs=ep)
            re0, re1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            barrier()
            t2 = time.perf_counter()
            re0.record()
            for i in range(launches):
                ops.connector_rollout(st, keys, (i + 2) * args.rollout_steps, args.rollout_steps,
                                      state_out=st, stats=stats, epochs=ep)
            re1.record()
            r_elapsed = close(t2)
            rm = sharding.all_reduce_metrics(
                {"env_steps": float(B * args.rollout_steps * launches),
                 "episodes": float(stats["episodes"].sum().item()), "elapsed_s": r_elapsed},
                device=dev)
            return (rm["env_steps"] / rm["elapsed_s"], re0.elapsed_time(re1) / launches,
                    rm["episodes"])

        v_stash, ms_stash, eps = run_rollout(True)
        v_plain, ms_plain, _ = run_rollout(False)
        rollout = {"metric": "env_steps_per_sec", "value": v_stash, "unit": "env-steps/s",
                   "parity": "unpinned (Connector.step semantics modelled on jumanji 0.2.2; the "
                             "random policy is this project's, not jumanji's RandomAgent)",
                   "steps_per_launch": args.rollout_steps, "launches": launches,
                   "auto_reset": True, "kernel": "connector_rollout_kernel + stash_refill_kernel",
                   "launch_avg_ms": ms_stash, "episodes_per_launch": eps,
                   "fused_resets": {"value": v_plain, "launch_avg_ms": ms_plain},
                   "note": "rbg_connector_rollout_prefetch: random policy + step, env resident in "
                           "LDS across steps, resets served from a two-slot next-episode stash "
                           "that one refill launch per rollout launch keeps full "
                           "(fused_resets: the walks inside the rollout launch instead); no "
                           "per-step observation traffic, so not HBM-bound (state in/out once "
                           "per launch)"}

        rollout["recorded"] = recorded_rollout(launches)

    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample != 0:
        cpu = cpu_baseline(args.cpu_sample)

    if rank == 0:
        line = {
            "metric": "boards_per_sec", "value": value, "unit": "boards/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": total_elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "timing": {"elapsed_s_max_over_ranks": total_elapsed,
                       "launch_ms_over_ranks": {"min": -metrics["neg_launch_ms_min"],
                                                "max": metrics["launch_ms_max"]},
                       "note": "each rank times its own K steps (barrier, clock, K launches, "
                               "local synchronize, clock) -- no collective inside the interval; "
                               "value = boards of all ranks / the MAX of those times; "
                               "launch_ms_over_ranks = HIP-event span / K per rank"},
            "collective": collective,
            "config": {"workload": "ParallelRandomWalkGenerator 16x16 grid, 10 agents, "
                                   f"batch {BATCH_PER_GPU} per GPU (BASELINE configs[1])",
                       "grid_size": m, "num_agents": N, "batch_per_gpu": B,
                       "global_batch": int(B * world), "keys": "split(PRNGKey(0), global_batch)",
                       "sharding": f"batch-sharded x{world}, no data-path collective",
                       "mean_walk_iters": metrics["iters"] / (B * world),
                       "max_walk_iters": metrics["max_iters"]},
            "roofline": roofline, "cpu_baseline": cpu, "env_step": env_step, "rollout": rollout,
        }
        if peaks is not None:
            line["roofline"]["peak_measured"] = peaks
        if scaling_blocks is not None:
            line["c4"] = scaling_blocks["c4"]
            line["strong"] = scaling_blocks["strong"]
            line["scaling_readings"] = scaling_blocks["note"]
        if philox is not None:
            line["philox_stream"] = philox
        if pipelined is not None:
            line["pipelined"] = pipelined
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
