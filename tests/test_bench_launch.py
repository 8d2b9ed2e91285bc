"""bench.py's multi-GPU launch, rehearsed on CPU: `python bench.py --gpus 2` outside a launcher
must start the one-process-per-GPU job itself (as a child process), and rank 0's single JSON
line and the child's exit code must come back.  `--rehearse-launch` swaps the GPU work for the
distributed plumbing only (gloo): sharding bounds, barriers, the metrics all-reduce."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra,
                          capture_output=True, text=True, timeout=300, env=env)


def test_bench_gpus_2_launches_itself():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--rehearse-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["rehearsal"] is True
    assert line["boards"] == 2 * 4096 * 3          # summed over the two ranks' shards
    assert line["elapsed_s_max_over_ranks"] == 2.0  # max over ranks of (1 + rank)
    # per-rank launch times are reduced to (min, max); the live group shows in the record
    assert abs(line["launch_ms_over_ranks"]["min"] - 0.03) < 1e-12
    assert abs(line["launch_ms_over_ranks"]["max"] - 0.04) < 1e-12
    coll = line["collective"]
    assert coll["backend"] == "gloo" and coll["world_size"] == 2 and coll["all_reduce_us"] > 0
    # the two other readings of N > 1: BASELINE configs[3] (4096 keys per rank of the 4096 N-key
    # batch) and the strong reading (the global batch of 4096 cut into N shards)
    c4, strong = line["c4"], line["strong"]
    assert c4["global_batch"] == 8192 and c4["boards_per_gpu"] == 4096 and c4["scaling"] == "weak"
    assert c4["boards"] == 2 * 4096 * 3
    assert strong["global_batch"] == 4096 and strong["boards_per_gpu"] == 2048
    assert strong["boards"] == 4096 * 3 and strong["scaling"] == "strong"


def test_bench_relays_the_childs_exit_code():
    # a world size that does not match --gpus makes every rank exit non-zero
    r = _run(["--gpus", "3", "--rehearse-launch"], {"WORLD_SIZE": "1"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stdout + r.stderr)


def test_bench_single_rank_rehearsal_needs_no_launcher():
    r = _run(["--gpus", "1", "--steps", "2", "--rehearse-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["boards"] == 4096 * 2
    assert line["collective"] == {"backend": None, "world_size": 1, "all_reduce_us": None}


def test_bench_reads_each_traffic_figure_from_its_own_profile():
    """roofline.traffic comes from the committed rocprofv3 PMC pass of the SAME launch population
    (the step line once carried the bytes of the 65 536-env stream run)."""
    sys.path.insert(0, ROOT)
    import bench

    gen = bench.measured_traffic("prw_generate_kernel", 4096)
    step = bench.measured_traffic("connector_step_kernel", 4096, "_step")
    stream = bench.measured_traffic("connector_step_kernel", 65536, "_stream", 65536)
    rec = bench.measured_traffic("connector_rollout_kernel", 4096, "_record")
    assert 5.2e6 < gen < 5.6e6          # 4096 x 1284 B algorithmic
    assert 5.3e7 < step < 5.6e7         # 4096 x 12 979 B
    assert 8.5e8 < stream < 8.8e8       # 65 536 x 12 979 B
    assert 2.13e9 < rec < 2.17e9        # 50 x 4096 x 10 427 B + the State
    assert bench.measured_traffic("connector_step_kernel", 4096, "_stream", 65536) is None
    assert bench.measured_issue("prw_generate_kernel", 4096)["valu_insts_per_launch"] > 1e7
