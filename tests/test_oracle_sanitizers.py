"""The checker checked: the CPU oracle (the only judge of the GPU parity tests) built with
AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle asan`) and driven through its
golden tests and through the shapes the GPU tests use (env rollouts with auto-reset, all four
generators, the 40x40/64 corner whose per-call work arrays are 400 KiB of stack).  CPU only."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r"""
from oracle import oracle as O
O.lib()
assert any("libprw_oracle_asan.so" in ln for ln in open("/proc/self/maps")), "sanitizer build not loaded"
k = O.split(O.prng_key(3), 64)
st, _, _ = O.generate_batch(k, 16, 10, threads=4)
ts = O.connector_observe_batch(st, threads=4)
for t in range(30):
    a = O.random_actions_batch(k, t, ts["action_mask"])
    ts = O.connector_step_batch(st, a, time_limit=7, auto_reset=True, threads=4)
O.connector_rollout_batch(st, k, 0, 12, time_limit=5, threads=4)
O.seq_rw_generator_batch(k, 16, 10, threads=4)
O.seq_rw_generator_batch(k[:16], 3, 2, threads=2)
O.seed_extension_batch(k, 16, 10, threads=4)
O.seed_extension_batch(k[:16], 12, 6, randomness=0.5, two_sided=False, extension_iterations=2, threads=4)
O.generate_batch(k[:8], 40, 64, threads=4)
O.generate_batch(k[:8], 5, 6, threads=2)
O.generate_batch(k[:8], 8, 4, "uniform", threads=2)
O.prw_philox_batch(k, 16, 10, threads=4)
h, t_, s, _ = O.generate_board_batch(k, 10, 5, threads=2)
O.dataset_sample_batch(k, h, t_, 10)
print("sanitizers clean")
"""


def _san_env():
    cc = os.environ.get("CC", "gcc")
    libs = " ".join(subprocess.check_output([cc, f"-print-file-name={n}"], text=True).strip()
                    for n in ("libasan.so", "libubsan.so"))
    env = dict(os.environ)
    env.update({"LD_PRELOAD": libs, "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1",
                "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1",
                "RBG_ORACLE_LIB": os.path.join(ROOT, "oracle", "libprw_oracle_asan.so"),
                "PYTHONPATH": ROOT})
    return env


def test_oracle_is_clean_under_asan_and_ubsan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = _san_env()
    r = subprocess.run([sys.executable, "-c", DRIVER], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "sanitizers clean" in r.stdout, (r.stdout + r.stderr)[-3000:]
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_oracle_goldens.py", "-x", "-q",
                        "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
