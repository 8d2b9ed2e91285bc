"""Shared pytest configuration.

Markers
  gpu  -- needs a real MI355X (run by the driver with `-m gpu` on the GPU box).
Everything else runs on CPU (`-m "not gpu"`).

/root/reference does not exist on the GPU box; no test may read it.  The reference's goldens are
committed as data in tests/golden/reference_goldens.json (see tests/golden/make_fixtures.py).
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X GPU")


@pytest.fixture(scope="session")
def goldens():
    with open(os.path.join(ROOT, "tests", "golden", "reference_goldens.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.build()
    return O


def as_np(x, dtype=np.int32):
    return np.asarray(x, dtype=dtype)


@pytest.fixture(scope="session")
def slow_path_lib(tmp_path_factory):
    """A build of the library (compiled where the test runs) whose rarely taken exact paths are
    the only ones: no prepared tries past the last cell in SeedExtension (RBG_SE_MARGIN=0), no
    step history for SequentialRandomWalk retries (RBG_SEQ_HISTORY_BYTES=0)."""
    import ctypes as C
    import subprocess

    from routing_board_generation_amd import _native as nat

    so = str(tmp_path_factory.mktemp("variant") / "librbg_slow_paths.so")
    nat.hipcc_compile(["-DRBG_SE_MARGIN=0", "-DRBG_SEQ_HISTORY_BYTES=0", "-o", so, nat.SRC_PATH])
    lib = C.CDLL(so)
    for name, (restype, argtypes) in nat.SIGNATURES.items():
        getattr(lib, name).restype, getattr(lib, name).argtypes = restype, argtypes
    return lib
