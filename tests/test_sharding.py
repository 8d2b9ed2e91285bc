"""Multi-process (gloo, world_size 2) tests of the batch sharding used for N > 1 GPUs.

The data path has no collective: each rank owns a contiguous key range.  These tests run on CPU;
the per-shard boards come from the oracle, so what is checked here is the partitioning and the
metrics all-reduce, not the kernels (tests/test_gpu_parity.py::test_full_size_properties checks
shard invariance of the HIP path itself).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_bounds_partition():
    from routing_board_generation_amd.sharding import shard_bounds

    for total in (0, 1, 7, 8, 4096, 32768, 100):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b and c <= d
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    # with world | total this is the reference's reshape(num_devices, per_device, 2)
    assert shard_bounds(32768, 3, 8) == (3 * 4096, 4 * 4096)
    with pytest.raises(ValueError):
        shard_bounds(8, 2, 2)


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from oracle import oracle as O
    from routing_board_generation_amd import sharding

    r, _, w = sharding.init_distributed(backend="gloo")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    B, m, N = 64, 8, 5
    keys = torch.from_numpy(O.split(O.prng_key(0), B))
    mine = sharding.shard_keys(keys, rank, world)
    lo, hi = sharding.shard_bounds(B, rank, world)
    assert mine.shape[0] == hi - lo
    st, solved, iters = O.generate_batch(mine.numpy(), m, N)
    np.save(os.path.join(tmpdir, f"solved_{rank}.npy"), solved)
    red = sharding.all_reduce_metrics({"boards": float(mine.shape[0]), "iters": float(iters.sum()),
                                       "max_iters": float(iters.max()),
                                       "elapsed_s": 1.0 + rank}, device=torch.device("cpu"))
    np.save(os.path.join(tmpdir, f"metrics_{rank}.npy"),
            np.array([red["boards"], red["iters"], red["max_iters"], red["elapsed_s"]]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_metrics(tmp_path, oracle):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    B, m, N = 64, 8, 5
    keys = oracle.split(oracle.prng_key(0), B)
    _, solved, iters = oracle.generate_batch(keys, m, N)
    parts = [np.load(tmp_path / f"solved_{r}.npy") for r in range(world)]
    assert (np.concatenate(parts) == solved).all(), "sharded boards differ from the single batch"
    for r in range(world):
        boards, it_sum, it_max, elapsed = np.load(tmp_path / f"metrics_{r}.npy")
        assert boards == B and it_sum == iters.sum() and it_max == iters.max()
        assert elapsed == 2.0  # max over ranks of (1 + rank)


def test_all_reduce_metrics_single_process():
    from routing_board_generation_amd.sharding import all_reduce_metrics

    out = all_reduce_metrics({"boards": 5.0, "max_iters": 7.0})
    assert out["boards"] == 5.0 and out["max_iters"] == 7.0 and out["env_steps"] == 0.0
