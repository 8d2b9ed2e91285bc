"""Batch sharding over the GPUs of one node: one process per GPU, `torch.distributed` (RCCL).

Replaces the reference's `jax.pmap(env.reset)(reset_keys.reshape(num_devices, per_device, 2))`
(routing_board_generation/rl_training/setup_train.py:396-400) and the `tree_map(jnp.mean)` /
`first_from_device` metric reduction of agent_training/training_script.py:78,116.

Boards and envs are independent, so the data path has NO collective: rank g simply owns the
contiguous key range [g*B/G, (g+1)*B/G) and produces exactly the boards a single GPU would have
produced for those keys (tests/test_sharding.py).  The only exchange is one small all-reduce of a
metrics vector per reporting interval -- latency-bound, so nothing here is tuned for xGMI
bandwidth.
"""
import os
from typing import Dict, Tuple

import torch
import torch.distributed as dist

from .random import split
from .types import State, TimeStep

SUM_METRICS = ("boards", "env_steps", "episodes", "num_connections", "total_path_length", "iters")
MAX_METRICS = ("max_iters", "elapsed_s", "launch_ms_max", "neg_launch_ms_min")


def init_distributed(backend: str = None, single_device: bool = False,
                     force: bool = False) -> Tuple[int, int, int]:
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (set by torch.distributed.run) and joins the
    process group.  Returns (rank, local_rank, world_size); world_size 1 needs no launcher and no
    process group unless force=True (a one-rank RCCL communicator: exercises the collective path
    on a one-GPU box).  single_device=True puts every rank on GPU 0 (a multi-rank rehearsal on a
    one-GPU box; needs backend="gloo", RCCL refuses two ranks on one device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if single_device else int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    return rank, local_rank, world


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced partition of range(total): the first (total % world) ranks get one
    extra unit.  With world | total this is the reference's reshape(num_devices, per_device)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_keys(keys: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    lo, hi = shard_bounds(keys.shape[0], rank, world)
    return keys[lo:hi].contiguous()


def reset_sharded(env, key: torch.Tensor, total_batch_size: int, rank: int,
                  world: int) -> Tuple[State, TimeStep, torch.Tensor]:
    """setup_training_state's key fan-out for this rank: params_key, reset_key, acting_key =
    split(key, 3); reset_keys = split(reset_key, total_batch_size); this rank resets its slice."""
    k3 = split(key, 3)
    reset_keys = split(k3[1], total_batch_size)
    acting_keys = split(k3[2], world)
    state, timestep = env.reset(shard_keys(reset_keys, rank, world))
    return state, timestep, acting_keys[rank]


def collective_info(device=None, reps: int = 20) -> Dict[str, object]:
    """What the live process group is and what one metrics all-reduce costs on it: evidence in a
    benchmark record that RCCL really saw N ranks (backend, world size, microseconds per
    all-reduce of the metrics vector, host wall-clock with the device synchronised)."""
    import time

    if not dist.is_initialized():
        return {"backend": None, "world_size": 1, "all_reduce_us": None}
    backend = dist.get_backend()
    if backend != "nccl":
        device = torch.device("cpu")
    elif device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    v = torch.zeros((len(SUM_METRICS),), dtype=torch.float64, device=device)
    for _ in range(3):
        dist.all_reduce(v)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_reduce(v)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    us = (time.perf_counter() - t0) / reps * 1e6
    return {"backend": backend, "world_size": dist.get_world_size(), "all_reduce_us": us,
            "vector_bytes": v.numel() * 8, "reps": reps}


def all_reduce_metrics(metrics: Dict[str, float], device=None) -> Dict[str, float]:
    """ONE all-reduce(sum) + ONE all-reduce(max) of a tiny vector (RCCL over xGMI on GPUs, gloo
    on CPU).  Keys in SUM_METRICS are summed, keys in MAX_METRICS take the max over ranks."""
    sums = [float(metrics.get(k, 0.0)) for k in SUM_METRICS]
    maxs = [float(metrics.get(k, 0.0)) for k in MAX_METRICS]
    if dist.is_initialized():
        if dist.get_backend() != "nccl":
            device = torch.device("cpu")  # gloo (CPU tests, single-GPU rehearsals)
        elif device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        s = torch.tensor(sums, dtype=torch.float64, device=device)
        m = torch.tensor(maxs, dtype=torch.float64, device=device)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        sums, maxs = s.tolist(), m.tolist()
    out = dict(zip(SUM_METRICS, sums))
    out.update(zip(MAX_METRICS, maxs))
    return out
