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


def test_bench_relays_the_childs_exit_code():
    # a world size that does not match --gpus makes every rank exit non-zero
    r = _run(["--gpus", "3", "--rehearse-launch"], {"WORLD_SIZE": "1"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stdout + r.stderr)


def test_bench_single_rank_rehearsal_needs_no_launcher():
    r = _run(["--gpus", "1", "--steps", "2", "--rehearse-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["boards"] == 4096 * 2
