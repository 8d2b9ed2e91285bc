"""Tensor-level launchers: torch device tensors in, C-ABI calls on torch's current HIP stream.

PyTorch is used for device memory and streams only; every value below is produced by the HIP
kernels of csrc/prw_kernels.hip.  All functions are batched ([B, ...]) and asynchronous.
"""
import contextlib
import ctypes as C
import sys
from typing import Dict, Optional

import numpy as np
import torch

from . import _native as nat
from .schedule import MissWatch, RefillSchedule

StateBufs = Dict[str, torch.Tensor]
TimestepBufs = Dict[str, torch.Tensor]

STATE_FIELDS = ("grid", "step_count", "start", "target", "position", "key")
TIMESTEP_FIELDS = ("step_type", "reward", "discount", "obs_grid", "action_mask",
                   "obs_step_count", "num_connections", "ratio_connections", "total_path_length",
                   "reward_total", "discount_max")


def _stream(device: torch.device) -> C.c_void_p:
    """hipStream_t of torch's current stream on `device` (the raw getter costs ~0.3 us,
    torch.cuda.current_stream(device).cuda_stream ~3.7 us)."""
    idx = device.index
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(
        idx if idx is not None else torch.cuda.current_device()))


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _check(t: torch.Tensor, dtype: torch.dtype, shape, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise ValueError(f"{name} must live on a HIP device (got {t.device}); this package has "
                         "no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} must have shape {tuple(shape)}, got {tuple(t.shape)}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


INT32_MAX = 2 ** 31 - 1


class GeneratorSpec:
    """Host side of include/rbg_prw.h::rbg_generator: which generator (re)starts episodes inside
    the env launches (setup_train.py:112-152), with the tensors it points to kept alive.
    `refill` = how its next-episode stash is best used (measured): "every",
    "depth", "serial", "adaptive" for the per-step path (EpochStash arguments; every = 0: no
    stash), "rollout_stash" / "record_stash": whether the fused / the recording rollout launch
    takes its resets from a stash (defaults True / False)."""

    def __init__(self, kind: int, randomness: float = 0.0, two_sided: bool = True,
                 extension_iterations: int = 1, max_extension_steps: int = INT32_MAX,
                 heads: Optional[torch.Tensor] = None, targets: Optional[torch.Tensor] = None,
                 refill: Optional[dict] = None) -> None:
        self.kind = int(kind)
        n_boards = 0
        if self.kind == nat.GEN_DATASET:
            n_boards, N, _ = heads.shape
            _check(heads, torch.int32, (n_boards, N, 2), "heads")
            _check(targets, torch.int32, (n_boards, N, 2), "targets")
        self._keepalive = (heads, targets)
        self.struct = nat.Generator(self.kind, float(randomness), int(bool(two_sided)),
                                    int(extension_iterations), int(max_extension_steps),
                                    int(n_boards), _ptr(heads), _ptr(targets))
        self.refill = dict(refill or {})

    @property
    def device(self):
        return self._keepalive[0].device if self._keepalive[0] is not None else None


def _gen_ref(gen: Optional["GeneratorSpec"]):
    """Pointer argument for the *_gen entry points (None = ParallelRandomWalk)."""
    if gen is None or gen.kind == nat.GEN_PARALLEL_RW:
        return None
    return C.byref(gen.struct)


def check_keys(keys: torch.Tensor) -> torch.Tensor:
    if keys.dim() != 2 or keys.shape[1] != 2:
        raise ValueError(f"keys must have shape [B, 2], got {tuple(keys.shape)}")
    return _check(keys, torch.uint32, keys.shape, "keys")


def alloc_state(B: int, m: int, N: int, device) -> StateBufs:
    i32 = dict(dtype=torch.int32, device=device)
    return {
        "grid": torch.empty((B, m, m), **i32),
        "step_count": torch.empty((B,), **i32),
        "start": torch.empty((B, N, 2), **i32),
        "target": torch.empty((B, N, 2), **i32),
        "position": torch.empty((B, N, 2), **i32),
        "key": torch.empty((B, 2), dtype=torch.uint32, device=device),
    }


def alloc_timestep(B: int, m: int, N: int, device, with_obs_grid: bool = True,
                   aggregate: bool = False) -> TimestepBufs:
    """aggregate=True: reward_total / discount_max [B] (MultiToSingleWrapper inside the launch)
    instead of the per-agent reward / discount [B, N]."""
    ts = {
        "step_type": torch.empty((B,), dtype=torch.int8, device=device),
        "reward": torch.empty((B, N), dtype=torch.float32, device=device),
        "discount": torch.empty((B, N), dtype=torch.float32, device=device),
        "action_mask": torch.empty((B, N, 5), dtype=torch.bool, device=device),
        "obs_step_count": torch.empty((B,), dtype=torch.int32, device=device),
        "num_connections": torch.empty((B,), dtype=torch.int32, device=device),
        "ratio_connections": torch.empty((B,), dtype=torch.float32, device=device),
        "total_path_length": torch.empty((B,), dtype=torch.int32, device=device),
    }
    if with_obs_grid:
        ts["obs_grid"] = torch.empty((B, N, m, m), dtype=torch.int32, device=device)
    if aggregate:
        del ts["reward"], ts["discount"]
        ts["reward_total"] = torch.empty((B,), dtype=torch.float32, device=device)
        ts["discount_max"] = torch.empty((B,), dtype=torch.float32, device=device)
    return ts


def _state_struct(st: StateBufs, B: int, m: int, N: int) -> nat.State:
    _check(st["grid"], torch.int32, (B, m, m), "state.grid")
    _check(st["step_count"], torch.int32, (B,), "state.step_count")
    for f in ("start", "target", "position"):
        _check(st[f], torch.int32, (B, N, 2), f"state.agents.{f}")
    _check(st["key"], torch.uint32, (B, 2), "state.key")
    return nat.State(*[st[f].data_ptr() for f in STATE_FIELDS])


def _timestep_struct(ts: TimestepBufs, B: int, m: int, N: int) -> nat.Timestep:
    shapes = {
        "step_type": (torch.int8, (B,)), "reward": (torch.float32, (B, N)),
        "discount": (torch.float32, (B, N)), "obs_grid": (torch.int32, (B, N, m, m)),
        "action_mask": (torch.bool, (B, N, 5)), "obs_step_count": (torch.int32, (B,)),
        "num_connections": (torch.int32, (B,)), "ratio_connections": (torch.float32, (B,)),
        "total_path_length": (torch.int32, (B,)),
        "reward_total": (torch.float32, (B,)), "discount_max": (torch.float32, (B,)),
    }
    ptrs = []
    for f in TIMESTEP_FIELDS:
        t = ts.get(f)
        if t is not None:
            _check(t, shapes[f][0], shapes[f][1], f"timestep.{f}")
        ptrs.append(_ptr(t))
    return nat.Timestep(*ptrs)


# ------------------------------------------------------------------------------------- PRNG
def split(key0: int, key1: int, num: int, device) -> torch.Tensor:
    out = torch.empty((num, 2), dtype=torch.uint32, device=device)
    nat.check(nat.lib().rbg_split(key0, key1, num, out.data_ptr(), _stream(out.device)),
              "rbg_split")
    return out


def split_batch(keys: torch.Tensor, num: int) -> torch.Tensor:
    keys = check_keys(keys)
    B = keys.shape[0]
    out = torch.empty((B, num, 2), dtype=torch.uint32, device=keys.device)
    nat.check(nat.lib().rbg_split_batch(keys.data_ptr(), B, num, out.data_ptr(),
                                        _stream(keys.device)), "rbg_split_batch")
    return out


# ------------------------------------------------------------------------------------- generators
def prw_generate(keys: torch.Tensor, m: int, N: int, split_first: bool, want_solved: bool = True,
                 want_train: bool = True, want_iters: bool = False, want_state: bool = False,
                 out: Optional[dict] = None):
    """Returns dict(solved_grid?, train_grid?, start, target, position?, step_count?, key_out,
    iters?); `out` may carry preallocated buffers under the same names."""
    keys = check_keys(keys)
    B, dev = keys.shape[0], keys.device
    i32 = dict(dtype=torch.int32, device=dev)
    o = out if out is not None else {}
    if want_solved and "solved_grid" not in o:
        o["solved_grid"] = torch.empty((B, m, m), **i32)
    if want_train and "train_grid" not in o:
        o["train_grid"] = torch.empty((B, m, m), **i32)
    if "start" not in o:
        o["start"] = torch.empty((B, N, 2), **i32)
    if "target" not in o:
        o["target"] = torch.empty((B, N, 2), **i32)
    if "key_out" not in o:
        o["key_out"] = torch.empty((B, 2), dtype=torch.uint32, device=dev)
    if want_iters and "iters" not in o:
        o["iters"] = torch.empty((B,), **i32)
    if want_state and "position" not in o:
        o["position"] = torch.empty((B, N, 2), **i32)
    if want_state and "step_count" not in o:
        o["step_count"] = torch.empty((B,), **i32)
    if "position" in o:
        _check(o["position"], torch.int32, (B, N, 2), "position")
    if "step_count" in o:
        _check(o["step_count"], torch.int32, (B,), "step_count")
    for f in ("solved_grid", "train_grid"):
        if f in o:
            _check(o[f], torch.int32, (B, m, m), f)
    _check(o["start"], torch.int32, (B, N, 2), "start")
    _check(o["target"], torch.int32, (B, N, 2), "target")
    _check(o["key_out"], torch.uint32, (B, 2), "key_out")
    nat.check(nat.lib().rbg_prw_generate(
        keys.data_ptr(), B, m, N, int(split_first), _ptr(o.get("solved_grid")),
        _ptr(o.get("train_grid")), o["start"].data_ptr(), o["target"].data_ptr(),
        _ptr(o.get("position")), _ptr(o.get("step_count")), o["key_out"].data_ptr(),
        _ptr(o.get("iters")), _stream(dev)), "rbg_prw_generate")
    return o


def prw_generate_philox(keys: torch.Tensor, m: int, N: int, want_solved: bool = False,
                        want_train: bool = True, out: Optional[dict] = None):
    """The walk on the opt-in Philox stream (project-defined; include/rbg_prw.h): dict(solved_grid?,
    train_grid?, start, target, position, step_count, key_out, iters)."""
    keys = check_keys(keys)
    B, dev = keys.shape[0], keys.device
    i32 = dict(dtype=torch.int32, device=dev)
    o = out if out is not None else {}
    for name, shape, dt in (("start", (B, N, 2), torch.int32), ("target", (B, N, 2), torch.int32),
                            ("position", (B, N, 2), torch.int32), ("step_count", (B,), torch.int32),
                            ("key_out", (B, 2), torch.uint32), ("iters", (B,), torch.int32)):
        if name not in o:
            o[name] = torch.empty(shape, dtype=dt, device=dev)
    if want_solved and "solved_grid" not in o:
        o["solved_grid"] = torch.empty((B, m, m), **i32)
    if want_train and "train_grid" not in o:
        o["train_grid"] = torch.empty((B, m, m), **i32)
    nat.check(nat.lib().rbg_prw_generate_philox(
        keys.data_ptr(), B, m, N, _ptr(o.get("solved_grid")), _ptr(o.get("train_grid")),
        o["start"].data_ptr(), o["target"].data_ptr(), o["position"].data_ptr(),
        o["step_count"].data_ptr(), o["key_out"].data_ptr(), o["iters"].data_ptr(),
        _stream(dev)), "rbg_prw_generate_philox")
    return o


def seq_rw_generate(keys: torch.Tensor, m: int, N: int, split_first: bool,
                    want_solved: bool = True, want_train: bool = True):
    """SequentialRandomWalk: dict(solved_grid?, train_grid?, start, target, position, step_count,
    key_out, attempts)."""
    keys = check_keys(keys)
    B, dev = keys.shape[0], keys.device
    i32 = dict(dtype=torch.int32, device=dev)
    o = {"start": torch.empty((B, N, 2), **i32), "target": torch.empty((B, N, 2), **i32),
         "position": torch.empty((B, N, 2), **i32), "step_count": torch.empty((B,), **i32),
         "key_out": torch.empty((B, 2), dtype=torch.uint32, device=dev),
         "attempts": torch.empty((B,), **i32)}
    if want_solved:
        o["solved_grid"] = torch.empty((B, m, m), **i32)
    if want_train:
        o["train_grid"] = torch.empty((B, m, m), **i32)
    nat.check(nat.lib().rbg_seq_rw_generate(
        keys.data_ptr(), B, m, N, int(split_first), _ptr(o.get("solved_grid")),
        _ptr(o.get("train_grid")), o["start"].data_ptr(), o["target"].data_ptr(),
        o["position"].data_ptr(), o["step_count"].data_ptr(), o["key_out"].data_ptr(),
        o["attempts"].data_ptr(), _stream(dev)), "rbg_seq_rw_generate")
    return o


def prw_generate_launcher(keys: torch.Tensor, m: int, N: int, split_first: bool, out: dict):
    """A bound launch of rbg_prw_generate: validates `keys` and the buffers of `out` (as returned
    by prw_generate) ONCE and returns a callable that only makes the C call on the stream that is
    current at that time -- for loops that relaunch into the same buffers (bench.py), where the
    per-call checks of prw_generate (~35 us of Python) cost as much as the kernel takes."""
    prw_generate(keys, m, N, split_first, want_solved="solved_grid" in out,
                 want_train="train_grid" in out, want_iters="iters" in out,
                 want_state="position" in out, out=out)  # validates (and runs once)
    keys = check_keys(keys)
    B, dev = keys.shape[0], keys.device
    lib = nat.lib()
    args = (keys.data_ptr(), B, m, N, int(split_first), _ptr(out.get("solved_grid")),
            _ptr(out.get("train_grid")), out["start"].data_ptr(), out["target"].data_ptr(),
            _ptr(out.get("position")), _ptr(out.get("step_count")), out["key_out"].data_ptr(),
            _ptr(out.get("iters")))
    def launch() -> None:
        rc = lib.rbg_prw_generate(*args, _stream(dev))
        if rc:
            nat.check(rc, "rbg_prw_generate")

    launch.keepalive = (keys, out)
    return launch


def extension_steps_to_int(extension_steps) -> int:
    """The reference compares an int32 sweep counter with `extension_steps` in float32
    (post_processor_utils_jax.py:51-55; default 1e23 = no limit): the number of sweeps allowed is
    the smallest integer s with not (float32(s) < float32(extension_steps))."""
    x = float(extension_steps)
    if x != x or x <= 0.0:
        return 0
    if x >= float(1 << 24):  # beyond float32's exact integers: nothing reaches it
        return INT32_MAX
    import math
    return min(INT32_MAX, int(math.ceil(x)))


def seed_extension_generate(keys: torch.Tensor, m: int, N: int, split_first: bool,
                            randomness: float = 0.0, two_sided: bool = True,
                            extension_iterations: int = 1, extension_steps=1e23,
                            want_solved: bool = False, want_train: bool = True):
    """SeedExtension: dict(solved_grid?, train_grid?, start, target, position, step_count,
    key_out, sweeps)."""
    keys = check_keys(keys)
    B, dev = keys.shape[0], keys.device
    i32 = dict(dtype=torch.int32, device=dev)
    o = {"start": torch.empty((B, N, 2), **i32), "target": torch.empty((B, N, 2), **i32),
         "position": torch.empty((B, N, 2), **i32), "step_count": torch.empty((B,), **i32),
         "key_out": torch.empty((B, 2), dtype=torch.uint32, device=dev),
         "sweeps": torch.empty((B,), **i32)}
    if want_solved:
        o["solved_grid"] = torch.empty((B, m, m), **i32)
    if want_train:
        o["train_grid"] = torch.empty((B, m, m), **i32)
    nat.check(nat.lib().rbg_seed_extension_generate(
        keys.data_ptr(), B, m, N, int(split_first), float(randomness), int(bool(two_sided)),
        int(extension_iterations), extension_steps_to_int(extension_steps),  # -1: seed only
        _ptr(o.get("solved_grid")), _ptr(o.get("train_grid")), o["start"].data_ptr(),
        o["target"].data_ptr(), o["position"].data_ptr(), o["step_count"].data_ptr(),
        o["key_out"].data_ptr(), o["sweeps"].data_ptr(), _stream(dev)),
        "rbg_seed_extension_generate")
    return o


def uniform_generate(keys: torch.Tensor, m: int, N: int) -> StateBufs:
    keys = check_keys(keys)
    B, dev = keys.shape[0], keys.device
    st = alloc_state(B, m, N, dev)
    nat.check(nat.lib().rbg_uniform_generate(
        keys.data_ptr(), B, m, N, st["grid"].data_ptr(), st["start"].data_ptr(),
        st["target"].data_ptr(), st["position"].data_ptr(), st["step_count"].data_ptr(),
        st["key"].data_ptr(), _stream(dev)), "rbg_uniform_generate")
    return st


def dataset_sample(keys: torch.Tensor, heads: torch.Tensor, targets: torch.Tensor, m: int,
                   want_which: bool = False):
    """Sample boards of a pre-generated dataset (heads/targets [n_boards, N, 2]) -> state bufs."""
    keys = check_keys(keys)
    B, dev = keys.shape[0], keys.device
    n_boards, N, _ = heads.shape
    _check(heads, torch.int32, (n_boards, N, 2), "heads")
    _check(targets, torch.int32, (n_boards, N, 2), "targets")
    st = alloc_state(B, m, N, dev)
    which = torch.empty((B,), dtype=torch.int32, device=dev) if want_which else None
    s = _state_struct(st, B, m, N)
    nat.check(nat.lib().rbg_dataset_sample(keys.data_ptr(), B, m, N, n_boards, heads.data_ptr(),
                                           targets.data_ptr(), C.byref(s), _ptr(which),
                                           _stream(dev)), "rbg_dataset_sample")
    return st, which


# --------------------------------------------------------------------- the walk taken apart
def prw_initialise_agents(keys: torch.Tensor, m: int, N: int,
                          grid: Optional[torch.Tensor] = None):
    """_initialise_agents for keys [B, 2] (grid [B, m, m] int32 or None = empty)
    -> (grid [B, m, m], start [B, N, 2])."""
    keys = check_keys(keys)
    B, dev = keys.shape[0], keys.device
    if grid is not None:
        _check(grid, torch.int32, (B, m, m), "grid")
    out = torch.empty((B, m, m), dtype=torch.int32, device=dev)
    start = torch.empty((B, N, 2), dtype=torch.int32, device=dev)
    nat.check(nat.lib().rbg_prw_initialise_agents(keys.data_ptr(), B, m, N, _ptr(grid),
                                                  out.data_ptr(), start.data_ptr(), _stream(dev)),
              "rbg_prw_initialise_agents")
    return out, start


def prw_step(keys: torch.Tensor, grid: torch.Tensor, target: Optional[torch.Tensor],
             position: torch.Tensor, want_actions: bool = False):
    """_step for B boards -> (next keys [B, 2], grid [B, m, m], position [B, N, 2], actions)."""
    keys = check_keys(keys)
    B, dev = keys.shape[0], keys.device
    m, N = grid.shape[-1], position.shape[1]
    _check(grid, torch.int32, (B, m, m), "grid")
    _check(position, torch.int32, (B, N, 2), "position")
    if target is not None:
        _check(target, torch.int32, (B, N, 2), "target")
    key_out = torch.empty_like(keys)
    grid_out, pos_out = torch.empty_like(grid), torch.empty_like(position)
    actions = torch.empty((B, N), dtype=torch.int32, device=dev) if want_actions else None
    nat.check(nat.lib().rbg_prw_step(keys.data_ptr(), B, m, N, grid.data_ptr(), _ptr(target),
                                     position.data_ptr(), key_out.data_ptr(), grid_out.data_ptr(),
                                     pos_out.data_ptr(), _ptr(actions), _stream(dev)),
              "rbg_prw_step")
    return key_out, grid_out, pos_out, actions


def prw_select_action(agent_keys: torch.Tensor, grid: torch.Tensor,
                      position: torch.Tensor) -> torch.Tensor:
    """_select_action with one key per agent: agent_keys [B, N, 2] -> actions [B, N]."""
    B, N = position.shape[0], position.shape[1]
    m, dev = grid.shape[-1], grid.device
    _check(agent_keys, torch.uint32, (B, N, 2), "agent_keys")
    _check(grid, torch.int32, (B, m, m), "grid")
    _check(position, torch.int32, (B, N, 2), "position")
    action = torch.empty((B, N), dtype=torch.int32, device=dev)
    nat.check(nat.lib().rbg_prw_select_action(agent_keys.data_ptr(), B, m, N, grid.data_ptr(),
                                              position.data_ptr(), action.data_ptr(),
                                              _stream(dev)), "rbg_prw_select_action")
    return action


def prw_cell_queries(grid: Optional[torch.Tensor], cells: torch.Tensor, m: int,
                     wire_id: Optional[torch.Tensor] = None, want=("adjacent",)) -> dict:
    """cells [B, Q] int32 -> the requested of adjacent / available [B, Q, 4] int32, is_free /
    not_doubling_back [B, Q] bool."""
    B, Q = cells.shape
    dev = cells.device
    _check(cells, torch.int32, (B, Q), "cells")
    if grid is not None:
        _check(grid, torch.int32, (B, m, m), "grid")
    if wire_id is not None:
        _check(wire_id, torch.int32, (B, Q), "wire_id")
    out = {}
    for name in want:
        if name in ("adjacent", "available"):
            out[name] = torch.empty((B, Q, 4), dtype=torch.int32, device=dev)
        elif name in ("is_free", "not_doubling_back"):
            out[name] = torch.empty((B, Q), dtype=torch.uint8, device=dev)
        else:
            raise ValueError(f"unknown cell query {name!r}")
    nat.check(nat.lib().rbg_prw_cell_queries(
        _ptr(grid), B, m, cells.data_ptr(), Q, _ptr(wire_id), _ptr(out.get("adjacent")),
        _ptr(out.get("available")), _ptr(out.get("is_free")), _ptr(out.get("not_doubling_back")),
        _stream(dev)), "rbg_prw_cell_queries")
    for name in ("is_free", "not_doubling_back"):
        if name in out:
            out[name] = out[name].bool()
    return out


def prw_step_agent(grid: torch.Tensor, target: Optional[torch.Tensor], position: torch.Tensor,
                   action: torch.Tensor, want_grids: bool = True):
    """_step_agent of every agent on its own copy of the grid -> (new position [B, N, 2],
    grids [B, N, m, m] or None, moved [B, N] bool)."""
    B, N = position.shape[0], position.shape[1]
    m, dev = grid.shape[-1], grid.device
    _check(grid, torch.int32, (B, m, m), "grid")
    _check(position, torch.int32, (B, N, 2), "position")
    _check(action, torch.int32, (B, N), "action")
    if target is not None:
        _check(target, torch.int32, (B, N, 2), "target")
    new_pos = torch.empty_like(position)
    grids = torch.empty((B, N, m, m), dtype=torch.int32, device=dev) if want_grids else None
    moved = torch.empty((B, N), dtype=torch.uint8, device=dev)
    nat.check(nat.lib().rbg_prw_step_agent(grid.data_ptr(), B, m, N, _ptr(target),
                                           position.data_ptr(), action.data_ptr(),
                                           new_pos.data_ptr(), _ptr(grids), moved.data_ptr(),
                                           _stream(dev)), "rbg_prw_step_agent")
    return new_pos, grids, moved.bool()


# ------------------------------------------------------------------------------------- Connector
def connector_reset(keys: torch.Tensor, m: int, N: int, state: Optional[StateBufs] = None,
                    timestep: Optional[TimestepBufs] = None,
                    gen: Optional[GeneratorSpec] = None):
    """Connector(generator).reset(keys) in one launch; gen=None is the ParallelRandomWalk."""
    keys = check_keys(keys)
    B, dev = keys.shape[0], keys.device
    st = state if state is not None else alloc_state(B, m, N, dev)
    ts = timestep if timestep is not None else alloc_timestep(B, m, N, dev)
    s, t = _state_struct(st, B, m, N), _timestep_struct(ts, B, m, N)
    nat.check(nat.lib().rbg_connector_reset_gen(_gen_ref(gen), keys.data_ptr(), B, m, N,
                                                C.byref(s), C.byref(t), None, 0, 0, _stream(dev)),
              "rbg_connector_reset")
    return st, ts


def connector_observe(state: StateBufs, timestep: Optional[TimestepBufs] = None):
    B, m, _ = state["grid"].shape
    N = state["start"].shape[1]
    dev = state["grid"].device
    ts = timestep if timestep is not None else alloc_timestep(B, m, N, dev)
    s, t = _state_struct(state, B, m, N), _timestep_struct(ts, B, m, N)
    nat.check(nat.lib().rbg_connector_observe(B, m, N, C.byref(s), C.byref(t), _stream(dev)),
              "rbg_connector_observe")
    return ts


def connector_step(state: StateBufs, action: torch.Tensor, time_limit: int = 50,
                   timestep_reward: float = -0.03, connected_reward: float = 0.1,
                   auto_reset: bool = False, state_out: Optional[StateBufs] = None,
                   timestep: Optional[TimestepBufs] = None,
                   gen: Optional[GeneratorSpec] = None):
    """state_out=None allocates a fresh state (functional); pass state_out=state to step in place.
    `gen` = the generator that restarts finished episodes when auto_reset is set."""
    B, m, _ = state["grid"].shape
    N = state["start"].shape[1]
    dev = state["grid"].device
    _check(action, torch.int32, (B, N), "action")
    so = state_out if state_out is not None else alloc_state(B, m, N, dev)
    ts = timestep if timestep is not None else alloc_timestep(B, m, N, dev)
    si, s_o = _state_struct(state, B, m, N), _state_struct(so, B, m, N)
    t = _timestep_struct(ts, B, m, N)
    p = nat.EnvParams(int(time_limit), float(timestep_reward), float(connected_reward),
                      int(bool(auto_reset)))
    nat.check(nat.lib().rbg_connector_step_gen(_gen_ref(gen), B, m, N, C.byref(si), C.byref(s_o),
                                               action.data_ptr(), None, 0, C.byref(p),
                                               C.byref(t), None, 0, 0, _stream(dev)),
              "rbg_connector_step")
    return so, ts


# ------------------------------------------------------------------------------------- prefetch
def alloc_stash(B: int, m: int, N: int, device, depth: int = 1, rolling: bool = False) -> dict:
    """Buffers of include/rbg_prw.h::rbg_stash (tag / ready_at / refill_key / hits zeroed);
    `depth` slots per env, slot s of env b in row s * B + b.  rolling: also the in-flight marks
    of refills that overlap each other and the step launches (EpochStash(serial=False))."""
    if depth not in range(1, 13):
        raise ValueError("stash depth must be in [1, 12]")
    R = depth * B
    return {"next": alloc_state(R, m, N, device),
            "tag": torch.zeros((R, 2), dtype=torch.uint32, device=device),
            "ready_at": torch.zeros((R,), dtype=torch.int32, device=device),
            "refill_key": torch.zeros((R, 2), dtype=torch.uint32, device=device),
            "hits": torch.zeros((B,), dtype=torch.int32, device=device),
            "misses": torch.zeros((B,), dtype=torch.int32, device=device),
            "scratch": torch.zeros((R + 1,), dtype=torch.int32, device=device),
            "inflight": (torch.zeros((R,), dtype=torch.int32, device=device) if rolling else None),
            "far_key": torch.zeros((B, 2), dtype=torch.uint32, device=device),
            "depth": depth}


def _stash_struct(stash: dict, B: int, m: int, N: int) -> nat.Stash:
    depth = int(stash.get("depth", 1))
    R = depth * B
    nxt = _state_struct(stash["next"], R, m, N)
    _check(stash["tag"], torch.uint32, (R, 2), "stash.tag")
    _check(stash["ready_at"], torch.int32, (R,), "stash.ready_at")
    _check(stash["refill_key"], torch.uint32, (R, 2), "stash.refill_key")
    _check(stash["hits"], torch.int32, (B,), "stash.hits")
    misses = stash.get("misses")
    if misses is not None:
        _check(misses, torch.int32, (B,), "stash.misses")
    scratch = stash.get("scratch")
    if scratch is not None:
        _check(scratch, torch.int32, (R + 1,), "stash.scratch")
    inflight, far = stash.get("inflight"), stash.get("far_key")
    if inflight is not None:
        _check(inflight, torch.int32, (R,), "stash.inflight")
    if far is not None:
        _check(far, torch.uint32, (B, 2), "stash.far_key")
    return nat.Stash(nxt, stash["tag"].data_ptr(), stash["ready_at"].data_ptr(),
                     stash["refill_key"].data_ptr(), stash["hits"].data_ptr(), _ptr(misses),
                     _ptr(scratch), _ptr(inflight), _ptr(far), depth)


def connector_reset_prefetch(keys: torch.Tensor, m: int, N: int, stash: dict, tick: int,
                             ready_delay: int, state: Optional[StateBufs] = None,
                             timestep: Optional[TimestepBufs] = None,
                             gen: Optional[GeneratorSpec] = None):
    keys = check_keys(keys)
    B, dev = keys.shape[0], keys.device
    st = state if state is not None else alloc_state(B, m, N, dev)
    ts = timestep if timestep is not None else alloc_timestep(B, m, N, dev)
    s, t, h = _state_struct(st, B, m, N), _timestep_struct(ts, B, m, N), _stash_struct(stash, B, m, N)
    nat.check(nat.lib().rbg_connector_reset_gen(
        _gen_ref(gen), keys.data_ptr(), B, m, N, C.byref(s), C.byref(t), C.byref(h), int(tick),
        int(ready_delay), _stream(dev)), "rbg_connector_reset_prefetch")
    return st, ts


def connector_step_prefetch(state: StateBufs, action: torch.Tensor, stash: dict, tick: int,
                            ready_delay: int, time_limit: int = 50,
                            timestep_reward: float = -0.03, connected_reward: float = 0.1,
                            state_out: Optional[StateBufs] = None,
                            timestep: Optional[TimestepBufs] = None,
                            gen: Optional[GeneratorSpec] = None):
    """Auto-resetting step that takes finished episodes' next State from `stash` when it can."""
    B, m, _ = state["grid"].shape
    N = state["start"].shape[1]
    dev = state["grid"].device
    _check(action, torch.int32, (B, N), "action")
    so = state_out if state_out is not None else alloc_state(B, m, N, dev)
    ts = timestep if timestep is not None else alloc_timestep(B, m, N, dev)
    si, s_o = _state_struct(state, B, m, N), _state_struct(so, B, m, N)
    t, h = _timestep_struct(ts, B, m, N), _stash_struct(stash, B, m, N)
    p = nat.EnvParams(int(time_limit), float(timestep_reward), float(connected_reward), 1)
    nat.check(nat.lib().rbg_connector_step_gen(
        _gen_ref(gen), B, m, N, C.byref(si), C.byref(s_o), action.data_ptr(), None, 0, C.byref(p),
        C.byref(t), C.byref(h), int(tick), int(ready_delay), _stream(dev)),
        "rbg_connector_step_prefetch")
    return so, ts


def stash_refill(stash: dict, m: int, gen: Optional[GeneratorSpec] = None) -> None:
    """Launch the refill on torch's current stream."""
    B, N = stash["hits"].shape[0], stash["next"]["start"].shape[1]
    dev = stash["tag"].device
    h = _stash_struct(stash, B, m, N)
    nat.check(nat.lib().rbg_stash_refill_gen(_gen_ref(gen), B, m, N, C.byref(h), _stream(dev)),
              "rbg_stash_refill")


_GEN_STREAMS: Dict[str, list] = {}


def generator_streams(device, n: int) -> list:
    """The process's generator streams on `device` (at most 3, created once and shared by every
    stash that refills beside its stepping stream).  HIP multiplexes streams onto 4 hardware
    queues in creation order: a stash that made streams of its own would, from the second stash
    on, put two of them -- or one of them and the stepping stream -- on one queue, where their
    launches run one after the other (measured: 18 us per step became 25-28)."""
    pool = _GEN_STREAMS.setdefault(str(torch.device(device)), [])
    while len(pool) < min(3, max(1, int(n))):
        pool.append(torch.cuda.Stream(device=device))
    return pool[:max(1, min(3, int(n)))]


class EpochStash:
    """Host side of the stash protocol in its epoch form: launches are numbered (ticks) and
    grouped in epochs of K; ONE refill is queued after the last launch of each epoch.

    serial=True (default): the refill is queued on the launches' own stream, so an entry
    requested in epoch e is usable from the first tick of epoch e + 1, and no events are needed.
    For the walk (a refill of ~30 us) this beats any overlap: a refill that shares the GPU with
    the step launches slows each of them down by more than it takes alone.

    serial=False ("rolling refills"): the refills run beside the stepping stream -- rows listed on
    one listing stream (rbg_stash_list, which also claims them so that overlapping refills never
    generate a row twice), generated on `streams` generator streams used in turn
    (rbg_stash_refill_listed) -- and an entry requested in epoch e is usable from the first tick
    of epoch e + `lead`, which the stepping stream makes sure of with one event wait per epoch.
    This is for the generators whose refill lasts as long as its slowest board whatever its row
    count (SeedExtension 1.4-1.8 ms, SequentialRandomWalk 0.4-0.5 ms) while leaving most of the
    chip idle: with one or two of them always in flight, generation keeps up with the ~85 boards
    per step that 4096 envs consume, at the price of ~6 us per step launch (two queues active).
    It needs lead time: an entry requested at position p of epoch e is generated by the refill
    queued behind that epoch and usable lead * K - p steps after its request, and the request
    rule asks `depth` episodes ahead -- so `depth` consecutive episodes must outlast lead * K
    steps (Connector sizes K by that, tools/probe_envgen.py --time-limit).  The
    requests of the reset launch are served by one refill on the stepping stream itself (usable
    from tick 1)."""

    def __init__(self, B: int, m: int, N: int, device, every: int, serial: bool = True,
                 depth: int = 1, gen: Optional[GeneratorSpec] = None,
                 adaptive: bool = False, lead: int = 2, streams: int = 2) -> None:
        """depth: slots per env (include/rbg_prw.h::rbg_stash): every entry is requested
        depth - 1 whole episodes before it is needed.
        gen: the generator whose producer refills the stash (None = ParallelRandomWalk).
        adaptive (serial only): watch the stash's miss counter and refill more often when
        episodes turn out shorter than the refill period -- a miss runs the generator inside the
        step launch, which for SequentialRandomWalk / SeedExtension stalls that launch for
        0.1-1 ms.  The period only moves between `every` / 8 and `every`; results never depend
        on it.  The counter travels to pinned host memory asynchronously (no synchronisation:
        a decision may lag by one refill).
        lead, streams (serial=False only): see the class docstring."""
        self.B, self.m, self.N, self.device = B, m, N, device
        self.gen, self._g = gen, _gen_ref(gen)
        self.serial = bool(serial)
        # when refills are queued (serial form; the first one follows the reset launch itself, so
        # that no early finisher of the first K steps runs the generator inline): schedule.py
        self.sched = RefillSchedule(every, B, adaptive=bool(adaptive) and self.serial)
        self.K0 = self.sched.K0
        self.stash = alloc_stash(B, m, N, device, depth, rolling=not self.serial)
        self.struct = _stash_struct(self.stash, B, m, N)
        self._lib = nat.lib()
        self.adaptive = self.sched.adaptive
        if self.adaptive:
            self._miss_dev = torch.zeros((1,), dtype=torch.int64, device=device)
            self._miss_host = torch.zeros((1,), dtype=torch.int64).pin_memory()
        self.fell_back = False
        self._watch, self._refills = MissWatch(), 0
        if not self.serial:
            # a stash that misses in rolling form (episodes much shorter than the lead it was
            # sized for) falls back to the serial, miss-adaptive form for good: _watch_misses
            self._miss_dev = torch.zeros((1,), dtype=torch.int64, device=device)
            self._miss_host = torch.zeros((1,), dtype=torch.int64).pin_memory()
            self.lead = max(2, int(lead))
            ring = self.lead + 1
            # (no stream of its own for the listing: HIP multiplexes streams onto 4 hardware
            # queues, and the stepping stream + up to 3 generator streams use them up; the list
            # launches are ordered among themselves by events instead)
            self._gen_streams = generator_streams(device, streams)
            self._launched = [torch.cuda.Event() for _ in range(ring)]
            self._listed = [torch.cuda.Event() for _ in range(ring)]
            self._done = [torch.cuda.Event() for _ in range(ring)]
            # every refill in flight lists its rows into a scratch area of its own
            self._scratch = [torch.zeros_like(self.stash["scratch"]) for _ in range(ring)]
            self._structs = []
            for sc in self._scratch:
                st = nat.Stash.from_buffer_copy(self.struct)
                st.scratch = sc.data_ptr()
                self._structs.append(st)

    @property
    def K(self) -> int:
        return self.sched.K

    @property
    def tick(self) -> int:
        return self.sched.tick

    @property
    def _refill_after(self) -> int:
        return self.sched.refill_after

    def _epoch(self) -> int:
        """Rolling form: tick 0 is the reset (served by a refill of its own), epoch e covers the
        ticks 1 + e K .. (e + 1) K."""
        return (self.tick - 1) // self.K

    def ready_delay(self) -> int:
        """Ticks from the current launch to the first one that may use what it requests."""
        if self.serial:
            return self.sched.ready_delay()
        if self.tick == 0:
            return 1
        return 1 + self.K * (self._epoch() + self.lead) - self.tick

    def before_launch(self, stream: Optional[torch.cuda.Stream] = None) -> None:
        """Right before a reset/step launch on `stream` (default: torch's current stream)."""
        if self.serial or self.tick == 0:
            return
        e = self._epoch()
        if (self.tick - 1) % self.K == 0 and e >= self.lead:
            # the first launch of epoch e may consume what the refill of epoch e - lead generated
            (stream or torch.cuda.current_stream(self.device)).wait_event(
                self._done[(e - self.lead) % len(self._done)])

    def _send_misses(self, stream: Optional[torch.cuda.Stream], stamp: int) -> None:
        """Queue a snapshot of the stash's miss counter, stamped, to pinned host memory (no
        synchronisation; schedule.MissWatch decodes it whenever it has arrived)."""
        with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
            torch.sum(self.stash["misses"], dim=0, keepdim=True, out=self._miss_dev)
            self._miss_dev.mul_(1 << MissWatch.STAMP_BITS).add_(stamp & MissWatch.STAMP_MASK)
            self._miss_host.copy_(self._miss_dev, non_blocking=True)

    def _adapt(self, stream: Optional[torch.cuda.Stream]) -> None:
        """Halve the period while misses keep coming, let it grow back after four calm refills."""
        got = self._watch.observe(int(self._miss_host[0]))  # whatever has arrived (no sync)
        if got is not None:
            self.sched.observe_misses(*got)
        self._refills += 1
        self._send_misses(stream, self._refills)

    def _watch_misses(self, stream: torch.cuda.Stream) -> None:
        """Rolling form, once per epoch: the stash's miss counter travels to pinned host memory
        without synchronising; if it grew by more than batch / 64 per epoch since the last
        snapshot the lead time this stash was sized for does not hold (short episodes: a trained
        policy), and every miss stalls a step launch for a board's latency -- so the stepping
        stream waits for the refills in flight once and the stash carries on in the serial,
        miss-adaptive form."""
        got = self._watch.observe(int(self._miss_host[0]))
        if got is not None and got[0] > max(1, self.B // 64) * got[1]:
            self.drain(stream)
            with torch.cuda.stream(stream):
                # every producer has finished: a row is consistent with its tag, so rows asked for
                # under the rolling form's long lead are usable as soon as their key comes up
                self.stash["ready_at"].zero_()
            self.serial = self.fell_back = True
            self.sched.adaptive = self.adaptive = True
            self.sched.K = self.sched.K0 = min(self.sched.K0, 48)
            self.sched.refill_after = self.sched.tick  # a refill right behind this launch
            self._refills = self._epoch() + 1  # (the serial form's stamps carry on from the last epoch's)
            return
        self._send_misses(stream, self._epoch() + 1)

    def after_launch(self, stream: Optional[torch.cuda.Stream] = None) -> None:
        """Right after it: the last launch of an epoch queues the refill."""
        if not self.serial and self.tick > 0 and (self.tick - 1) % self.K == self.K - 1:
            self._watch_misses(stream or torch.cuda.current_stream(self.device))
        if self.serial or self.tick == 0:
            if self.tick == 0 or self.sched.refill_due():
                sp = C.c_void_p(stream.cuda_stream) if stream is not None else _stream(self.device)
                nat.check(self._lib.rbg_stash_refill_gen(self._g, self.B, self.m, self.N,
                                                         C.byref(self.struct), sp),
                          "rbg_stash_refill")
                if self.adaptive:
                    self._adapt(stream)
                self.sched.refill_queued()
            self.sched.launched()
            return
        if (self.tick - 1) % self.K == self.K - 1:
            e = self._epoch()
            i = e % len(self._done)
            stream = stream or torch.cuda.current_stream(self.device)
            self._launched[i].record(stream)
            gs = self._gen_streams[e % len(self._gen_streams)]
            gs.wait_event(self._launched[i])
            if e > 0:  # list launches in order: each sees the claims of the one before
                gs.wait_event(self._listed[(e - 1) % len(self._listed)])
            nat.check(self._lib.rbg_stash_list(self.B, self.m, self.N, C.byref(self._structs[i]),
                                               C.c_void_p(gs.cuda_stream)), "rbg_stash_list")
            self._listed[i].record(gs)
            nat.check(self._lib.rbg_stash_refill_listed(self._g, self.B, self.m, self.N,
                                                        C.byref(self._structs[i]),
                                                        C.c_void_p(gs.cuda_stream)),
                      "rbg_stash_refill_listed")
            self._done[i].record(gs)
        self.sched.launched()

    def drain(self, stream: Optional[torch.cuda.Stream] = None) -> None:
        """Make `stream` wait for every refill in flight (before freeing the buffers)."""
        if self.serial:
            return
        stream = stream or torch.cuda.current_stream(self.device)
        for side in self._gen_streams:
            stream.wait_stream(side)


class StepLoop:
    """An in-place random-policy stepping loop with every launch descriptor built once
    ({ rbg_random_actions; rbg_connector_step[_prefetch] } per step + rbg_stash_refill): what a
    training loop that reuses its buffers looks like from the host.  The validated, allocating
    path is connector_step(); this class only removes per-call Python work (tensor checks,
    ctypes struct construction), which at B = 4096 costs more than the kernels themselves.

    refill_every = K > 0 turns on the next-episode stash (include/rbg_prw.h::rbg_stash) with the
    epoch protocol of EpochStash.  All main-stream launches go to the stream that was current at
    construction."""

    def __init__(self, state: StateBufs, timestep: TimestepBufs, policy_keys: torch.Tensor,
                 time_limit: int = 50, timestep_reward: float = -0.03,
                 connected_reward: float = 0.1, refill_every: int = 0,
                 fuse_policy: bool = False, auto_reset: bool = True,
                 serial_refill: bool = True, stash_depth: int = 1,
                 gen: Optional[GeneratorSpec] = None, refill_lead: int = 2,
                 refill_streams: int = 2) -> None:
        B, m, _ = state["grid"].shape
        N = state["start"].shape[1]
        dev = state["grid"].device
        self.B, self.m, self.N, self.dev = B, m, N, dev
        self.state, self.timestep = state, timestep
        self.keys = check_keys(policy_keys)
        self.action = torch.empty((B, N), dtype=torch.int32, device=dev)
        self._lib = nat.lib()
        self._s = _state_struct(state, B, m, N)
        self._t = _timestep_struct(timestep, B, m, N)
        self._p = nat.EnvParams(int(time_limit), float(timestep_reward), float(connected_reward),
                                1 if auto_reset else 0)
        self._main_stream = torch.cuda.current_stream(dev)
        self._main = C.c_void_p(self._main_stream.cuda_stream)
        self.K = max(0, int(refill_every))
        self.fuse_policy = bool(fuse_policy)  # policy + step in ONE launch (rollout of 1 step)
        self.gen, self._g = gen, _gen_ref(gen)
        self._epochs = (EpochStash(B, m, N, dev, self.K, serial=serial_refill, depth=stash_depth,
                                   gen=gen, lead=refill_lead, streams=refill_streams)
                        if self.K else None)
        self._hp = None
        if self.K:
            self.stash, self._h = self._epochs.stash, self._epochs.struct
            self._hp = C.byref(self._h)
        # the step launch as one struct (rbg_step_call): every pointer of this loop is fixed
        call = self._call = nat.StepCall()
        if self._g is not None:
            call.generator = C.pointer(gen.struct)
        call.B, call.m, call.N = B, m, N
        call.state_in = call.state_out = self._s
        call.timestep = self._t
        call.params = self._p
        if self.K:
            call.stash = C.pointer(self._h)
        call.stream = self._main
        self._call_ref = C.byref(call)
        self._step_call = self._lib.rbg_connector_step_call
        self._keys_ptr, self._action_ptr = self.keys.data_ptr(), self.action.data_ptr()
        self._mask_ptr = _ptr(timestep.get("action_mask"))

    @property
    def tick(self) -> int:
        return self._epochs.tick if self._epochs else 0

    def _ready_delay(self) -> int:
        return self._epochs.ready_delay()

    def _before_launch(self) -> None:
        self._epochs.before_launch(self._main_stream)

    def _after_launch(self) -> None:
        self._epochs.after_launch(self._main_stream)

    # -- launches ------------------------------------------------------------------------------
    def reset(self, keys: torch.Tensor) -> None:
        """env.reset(keys) into the loop's buffers."""
        keys = check_keys(keys)
        if self.K:
            self._before_launch()
        nat.check(self._lib.rbg_connector_reset_gen(
            self._g, keys.data_ptr(), self.B, self.m, self.N, C.byref(self._s), C.byref(self._t),
            self._hp, self.tick, self._ready_delay() if self.K else 0, self._main),
            "rbg_connector_reset")
        if self.K:
            self._after_launch()

    def step(self, t: int, action: Optional[torch.Tensor] = None) -> None:
        """Policy at global step t, then one auto-resetting env step, in place.  With `action`
        (int32 [B, N], e.g. recorded from an earlier pass) the policy launch is skipped."""
        call = self._call
        if action is not None:
            call.action = _check(action, torch.int32, (self.B, self.N), "action").data_ptr()
            call.policy_keys = None
        elif self.fuse_policy:  # policy + step in ONE launch
            call.action, call.policy_keys = None, self._keys_ptr
        else:
            rc = self._lib.rbg_random_actions(self._keys_ptr, self.B, self.N, int(t) & 0xFFFFFFFF,
                                              self._mask_ptr, self._action_ptr, self._main)
            if rc:
                nat.check(rc, "rbg_random_actions")
            call.action, call.policy_keys = self._action_ptr, None
        call.policy_t = int(t) & 0xFFFFFFFF
        ep = self._epochs
        if ep is not None:
            ep.before_launch(self._main_stream)
            call.tick, call.ready_delay = ep.tick, ep.ready_delay()
            rc = self._step_call(self._call_ref)
            ep.after_launch(self._main_stream)
        else:
            rc = self._step_call(self._call_ref)
        if rc:
            nat.check(rc, "rbg_connector_step")

    def drain(self) -> None:
        """Make the main stream wait for every refill in flight (before freeing the buffers)."""
        if self.K:
            self._epochs.drain(self._main_stream)



class GroupedStepLoop:
    """StepLoop over `groups` contiguous slices of the batch, each on a HIP stream of its own: a
    group's step t + 1 depends only on its own step t, so one group's per-launch fixed cost
    (dependent-launch floor, pipeline fill: ~5.7 of the 12 us of a 4096-env step launch,
    DESIGN.md 4.2) hides behind the other group's store stream.  Same buffers, same results as
    one StepLoop (tests).  The groups run ahead of torch's current stream: call join() before
    anything else reads the buffers -- a join costs an event round trip (~20 us), so this shape
    pays only for loops that read the buffers every many steps (random policy inside the step
    launch, trajectory statistics)."""

    def __init__(self, state: StateBufs, timestep: TimestepBufs, policy_keys: torch.Tensor,
                 groups: int = 2, **kw) -> None:
        B = state["grid"].shape[0]
        dev = state["grid"].device
        policy_keys = check_keys(policy_keys)
        self.B, self.device, self.groups = B, dev, int(groups)
        self.state, self.timestep = state, timestep
        base, extra = divmod(B, self.groups)
        self.bounds, lo = [], 0
        for g in range(self.groups):
            hi = lo + base + (1 if g < extra else 0)
            self.bounds.append((lo, hi))
            lo = hi
        # (from the process's pool of side streams -- see generator_streams -- as far as it goes)
        pool = generator_streams(dev, self.groups)
        self.streams = pool + [torch.cuda.Stream(device=dev) for _ in range(self.groups - len(pool))]
        self.loops = []
        cur = torch.cuda.current_stream(dev)
        for (lo, hi), st in zip(self.bounds, self.streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):  # StepLoop binds the stream that is current here
                self.loops.append(StepLoop({k: v[lo:hi] for k, v in state.items()},
                                           {k: v[lo:hi] for k, v in timestep.items()},
                                           policy_keys[lo:hi].contiguous(), **kw))

    def reset(self, keys: torch.Tensor) -> None:
        keys = check_keys(keys)
        cur = torch.cuda.current_stream(self.device)
        for (lo, hi), st, loop in zip(self.bounds, self.streams, self.loops):
            st.wait_stream(cur)  # the keys were produced on the caller's stream
            loop.reset(keys[lo:hi].contiguous())

    def step(self, t: int) -> None:
        for loop in self.loops:
            loop.step(t)

    def join(self) -> None:
        """torch's current stream waits for every group's work so far."""
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams:
            cur.wait_stream(st)

    def drain(self) -> None:
        for loop in self.loops:
            loop.drain()
        self.join()


class _Chunk:
    __slots__ = ("tensors", "views", "tbl", "addr", "flat", "storages", "base_ref", "base_use")


# use count of a storage (private torch API; without it chunks are never reused)
_storage_use_count = getattr(torch._C, "_storage_Use_Count", None)


class ConnectorSession:
    """Launch context of the allocating (functional) reset / step for one (B, m, N, device) and
    one set of env parameters: what Connector keeps between calls so that a step costs the host
    little more than allocating its outputs (at B = 4096 the checked path below costs more host
    time than the kernels take).  Output tensors are allocated here, so their pointer structs
    are built without re-validation, and the State returned last is remembered by tensor
    identity: feeding it to the next step -- the shape of every rollout loop -- skips validating
    the input too.  Any other State goes through the checked path.  refill_every = K > 0 adds
    the next-episode stash with the epoch protocol of EpochStash.

    donate=True gives up the functional style for speed, like jax's donate_argnums: step()
    advances the State it is given IN PLACE (and returns the same tensors), and the TimeStep
    lands in one of two buffer sets used alternately -- the timestep returned by a step stays
    intact during the next step and is overwritten by the one after.  No allocation per step."""

    def __init__(self, B: int, m: int, N: int, device, time_limit: int = 50,
                 timestep_reward: float = -0.03, connected_reward: float = 0.1,
                 auto_reset: bool = False, refill_every: int = 0, donate: bool = False,
                 aggregate: bool = False, gen: Optional[GeneratorSpec] = None,
                 chunk_steps: Optional[int] = None) -> None:
        """aggregate=True: the timestep carries reward_total / discount_max [B]
        (MultiToSingleWrapper inside the launch) instead of per-agent reward / discount.
        gen: the env's generator (None = ParallelRandomWalk).
        chunk_steps: steps per output chunk (default: as many as fit 1 GiB per field, at most 16).
        Every step's output tensors are views into a chunk, so a TimeStep kept alive (a replay
        buffer sampling every k-th step) pins the whole chunk -- up to chunk_steps times its own
        size; .clone() what is retained, or pass chunk_steps=1."""
        self.B, self.m, self.N, self.device = B, m, N, device
        self.gen, self._g = gen, _gen_ref(gen)
        self.donate, self.aggregate = bool(donate), bool(aggregate)
        self._ring, self._ring_i = None, 0
        self._lib = nat.lib()
        self._p = nat.EnvParams(int(time_limit), float(timestep_reward), float(connected_reward),
                                1 if auto_reset else 0)
        self.K = max(0, int(refill_every)) if auto_reset else 0
        self.epochs = None
        self._last = None  # (leaves in STATE_FIELDS order, nat.State) of the State returned last
        self.agent_ids = torch.arange(N, dtype=torch.int32, device=device).expand(B, N)
        # Outputs are carved from chunks that hold `chunk_steps` steps per field (one torch.empty
        # + one unbind per field and chunk instead of 15 torch.empty per step, which alone cost
        # ~25 us of host time).  Every step still gets memory of its own; a tensor kept alive
        # keeps its chunk alive, so chunks are capped at ~1 GiB per field (of 288 GB of HBM: at
        # 16x16/10, B = 4096 the observation takes 42 MB per step, and a cap of 256 MB meant a new
        # set of chunks -- 15 torch.empty + 15 unbind, 66 us -- every 6 steps: 11 us per step).
        i32, f32 = torch.int32, torch.float32
        self._specs = (
            ("grid", i32, (B, m, m)), ("step_count", i32, (B,)), ("start", i32, (B, N, 2)),
            ("target", i32, (B, N, 2)), ("position", i32, (B, N, 2)), ("key", torch.uint32, (B, 2)),
            ("step_type", torch.int8, (B,)),
            ("reward", None if aggregate else f32, (B, N)),
            ("discount", None if aggregate else f32, (B, N)),
            ("obs_grid", i32, (B, N, m, m)), ("action_mask", torch.bool, (B, N, 5)),
            ("obs_step_count", i32, (B,)), ("num_connections", i32, (B,)),
            ("ratio_connections", f32, (B,)), ("total_path_length", i32, (B,)),
            ("reward_total", f32 if aggregate else None, (B,)),
            ("discount_max", f32 if aggregate else None, (B,)))  # dtype None: output not wanted
        self.chunk_steps = (max(1, int(chunk_steps)) if chunk_steps else
                            max(1, min(16, (1 << 30) // (4 * B * N * m * m))))
        self._chunk, self._pool, self._cursor = None, [], 0
        self.pool_chunks = 3  # chunks kept for reuse (the current one, the one the State fed to
        #                       the next step still lives in, one spare)
        # the step launch as ONE struct (include/rbg_prw.h::rbg_step_call): the pointers that change
        # from step to step are copied in with two memmoves
        call = self._call = nat.StepCall()
        if self._g is not None:
            call.generator = C.pointer(gen.struct)
        call.B, call.m, call.N = B, m, N
        call.params = self._p
        self._call_ref = C.byref(call)
        self._addr_in = C.addressof(call) + nat.StepCall.state_in.offset
        self._addr_out = C.addressof(call) + nat.StepCall.state_out.offset
        assert nat.StepCall.timestep.offset == nat.StepCall.state_out.offset + 48
        self._step_call = self._lib.rbg_connector_step_call
        self._raw_stream = torch._C._cuda_getCurrentRawStream
        idx = torch.device(device).index
        self._dev_index = idx if idx is not None else torch.cuda.current_device()
        self._action_shape = torch.Size((B, N))

    def _new_chunk(self):
        """Memory, views and pointer table of `chunk_steps` steps' outputs."""
        S, dev = self.chunk_steps, self.device
        ch = _Chunk()
        ch.tensors = [None if dt is None else torch.empty((S,) + shape, dtype=dt, device=dev)
                      for _, dt, shape in self._specs]
        ch.views = list(zip(*[(None,) * S if c is None else c.unbind(0)
                              for c in ch.tensors]))  # [step][field]
        ch.tbl = np.zeros((S, 17), dtype=np.uint64)  # absent field: pointer 0 = NULL
        steps = np.arange(S, dtype=np.uint64)
        for j, c in enumerate(ch.tensors):
            if c is not None:
                ch.tbl[:, j] = np.uint64(c.data_ptr()) + steps * np.uint64(c.stride(0) * c.element_size())
        ch.addr = ch.tbl.ctypes.data
        # what "nobody but this session refers to the chunk" looks like (see _chunk_is_free)
        ch.flat = [v for step in ch.views for v in step if v is not None]
        ch.storages = [c.untyped_storage() for c in ch.tensors if c is not None]
        ch.base_ref = max(map(sys.getrefcount, ch.flat))
        ch.base_use = [_storage_use_count(st._cdata) for st in ch.storages] \
            if _storage_use_count else None
        return ch

    @staticmethod
    def _chunk_is_free(ch) -> bool:
        """True when no tensor of the chunk is referenced outside this session: neither one of
        the views it handed out (Python reference counts) nor anything derived from them (the
        storages' use counts: slices, reshapes, autograd's saved tensors).  Reusing such a chunk
        is what torch's caching allocator does with freed blocks -- same stream-ordering
        contract -- minus 15 allocations and 15 x chunk_steps view constructions."""
        if ch.base_use is None or max(map(sys.getrefcount, ch.flat)) > ch.base_ref:
            return False
        for st, base in zip(ch.storages, ch.base_use):
            if _storage_use_count(st._cdata) != base:
                return False
        return True

    def _fresh(self):
        """(state leaves, timestep leaves, address of the step's 17 output pointers, owner of that
        memory): rbg_state then rbg_timestep as they lie in rbg_step_call (state_out, timestep);
        leaves in STATE_FIELDS / TIMESTEP_FIELDS order.  Every step gets memory of its own; a
        chunk is reused only once nothing outside the session refers to it any more."""
        i = self._cursor
        ch = self._chunk
        if ch is None or i == self.chunk_steps:
            nxt = None
            for cand in self._pool:
                if cand is not ch and self._chunk_is_free(cand):
                    nxt = cand
                    break
            if nxt is None:
                nxt = self._new_chunk()
                if len(self._pool) < self.pool_chunks:
                    self._pool.append(nxt)
            ch = self._chunk = nxt
            i = 0
        self._cursor = i + 1
        v = ch.views[i]
        return v[:6], v[6:], ch.addr + i * 136, ch

    def _ring_next(self):
        """donate: the TimeStep lands in one of two buffer sets used alternately."""
        if self._ring is None:
            self._ring = []
            for _ in range(2):
                ts = alloc_timestep(self.B, self.m, self.N, self.device, aggregate=self.aggregate)
                v = tuple(ts.get(f) for f in TIMESTEP_FIELDS)
                row = np.array([_ptr(x) or 0 for x in v], dtype=np.uint64)
                self._ring.append((v, row, row.ctypes.data))
        self._ring_i ^= 1
        return self._ring[self._ring_i]

    def reset(self, keys: torch.Tensor):
        keys = check_keys(keys)
        if keys.shape[0] != self.B or keys.device != self.agent_ids.device:
            raise ValueError("keys do not match this session's batch size / device")
        if self.donate:
            bufs = alloc_state(self.B, self.m, self.N, self.device)
            st = tuple(bufs[f] for f in STATE_FIELDS)
            owner = np.array([x.data_ptr() for x in st], dtype=np.uint64)
            s_addr = owner.ctypes.data
            ts, _, t_addr = self._ring_next()
        else:
            st, ts, s_addr, owner = self._fresh()
            t_addr = s_addr + 48
        s, t = nat.State.from_address(s_addr), nat.Timestep.from_address(t_addr)
        ep_ref, tick, delay = None, 0, 0
        if self.K:
            if self.epochs is not None:
                self.epochs.drain()
            kw = ({k: v for k, v in self.gen.refill.items() if k in ("depth", "serial", "adaptive", "lead", "streams")}
                  if self.gen is not None else {})
            ep = self.epochs = EpochStash(self.B, self.m, self.N, self.device, self.K,
                                          gen=self.gen, **kw)
            self._call.stash = C.pointer(ep.struct)
            ep.before_launch()
            ep_ref, tick, delay = C.byref(ep.struct), ep.tick, ep.ready_delay()
        rc = self._lib.rbg_connector_reset_gen(
            self._g, keys.data_ptr(), self.B, self.m, self.N, C.byref(s), C.byref(t), ep_ref,
            tick, delay, _stream(self.device))
        if self.K:
            self.epochs.after_launch()
        if rc:
            nat.check(rc, "rbg_connector_reset")
        self._last = (st, s_addr, owner)
        return st, ts

    def step(self, leaves, action: torch.Tensor):
        """leaves = (grid, step_count, start, target, position, key) of a batched State."""
        last = self._last
        if (last is not None and leaves[0] is last[0][0] and leaves[1] is last[0][1]
                and leaves[2] is last[0][2] and leaves[3] is last[0][3]
                and leaves[4] is last[0][4] and leaves[5] is last[0][5]):
            in_addr, in_owner = last[1], last[2]
        else:
            held = {f: x.contiguous() for f, x in zip(STATE_FIELDS, leaves)}  # alive till launch
            in_owner = _state_struct(held, self.B, self.m, self.N)
            in_addr = C.addressof(in_owner)
            leaves = tuple(held[f] for f in STATE_FIELDS)
        if (action.dtype is not torch.int32 or action.shape != self._action_shape
                or not action.is_cuda or not action.is_contiguous()):
            _check(action, torch.int32, (self.B, self.N), "action")
        call = self._call
        C.memmove(self._addr_in, in_addr, 48)
        if self.donate:  # the State advances in place
            st, owner, s_addr = leaves, in_owner, in_addr
            ts, _, t_addr = self._ring_next()
            C.memmove(self._addr_out, in_addr, 48)
            C.memmove(self._addr_out + 48, t_addr, 88)
        else:
            st, ts, s_addr, owner = self._fresh()
            C.memmove(self._addr_out, s_addr, 136)
        call.action = action.data_ptr()
        call.stream = self._raw_stream(self._dev_index)
        ep = self.epochs
        if ep is not None:
            ep.before_launch()
            call.tick, call.ready_delay = ep.tick, ep.ready_delay()
            rc = self._step_call(self._call_ref)
            ep.after_launch()
        else:
            rc = self._step_call(self._call_ref)
        if rc:
            nat.check(rc, "rbg_connector_step")
        self._last = (st, s_addr, owner)
        return st, ts

    def drain(self) -> None:
        if self.epochs is not None:
            self.epochs.drain()


ROLLOUT_STATS = ("episodes", "connections_made", "unconnected_agent_steps", "final_connections",
                 "final_path_length")


def connector_rollout(state: StateBufs, policy_keys: torch.Tensor, t0: int, n_steps: int,
                      time_limit: int = 50, timestep_reward: float = -0.03,
                      connected_reward: float = 0.1, auto_reset: bool = True,
                      state_out: Optional[StateBufs] = None, stats: Optional[dict] = None,
                      timestep: Optional[TimestepBufs] = None, want_timestep: bool = False,
                      epochs: Optional["EpochStash"] = None,
                      gen: Optional[GeneratorSpec] = None):
    """n_steps of random-policy stepping in one launch.  Returns (state_out, stats, timestep);
    timestep is None unless a buffer dict is passed or want_timestep is set.  `epochs` (an
    EpochStash, best with depth=2 and every=1) serves the resets from the next-episode stash and
    refills it behind the launch; same results."""
    B, m, _ = state["grid"].shape
    N = state["start"].shape[1]
    dev = state["grid"].device
    policy_keys = check_keys(policy_keys)
    if policy_keys.shape[0] != B:
        raise ValueError("policy_keys and state disagree on B")
    so = state_out if state_out is not None else alloc_state(B, m, N, dev)
    st = stats if stats is not None else {k: torch.empty((B,), dtype=torch.int32, device=dev)
                                          for k in ROLLOUT_STATS}
    for k in ROLLOUT_STATS:
        if k in st:
            _check(st[k], torch.int32, (B,), f"stats.{k}")
    ts = timestep if timestep is not None else (alloc_timestep(B, m, N, dev) if want_timestep
                                                else None)
    si, s_o = _state_struct(state, B, m, N), _state_struct(so, B, m, N)
    sp = nat.RolloutStats(*[_ptr(st.get(k)) for k in ROLLOUT_STATS])
    p = nat.EnvParams(int(time_limit), float(timestep_reward), float(connected_reward),
                      int(bool(auto_reset)))
    tp = _timestep_struct(ts, B, m, N) if ts is not None else None
    use_stash = epochs is not None and auto_reset
    if use_stash:
        if (epochs.B, epochs.m, epochs.N) != (B, m, N):
            raise ValueError("epochs was built for another batch / grid / agent count")
        epochs.before_launch()
    rc = nat.lib().rbg_connector_rollout_gen(
        _gen_ref(gen), B, m, N, C.byref(si), C.byref(s_o), policy_keys.data_ptr(),
        int(t0) & 0xFFFFFFFF, int(n_steps), C.byref(p), C.byref(sp),
        C.byref(tp) if tp is not None else None, None, None,
        C.byref(epochs.struct) if use_stash else None, epochs.tick if use_stash else 0,
        epochs.ready_delay() if use_stash else 0, _stream(dev))
    if use_stash:
        epochs.after_launch()
    nat.check(rc, "rbg_connector_rollout")
    return so, st, ts


def alloc_trajectory(T: int, B: int, m: int, N: int, device, with_obs_grid: bool = True,
                     aggregate: bool = False) -> TimestepBufs:
    """Timestep buffers with a leading time axis [T, B, ...] + "action" [T, B, N]."""
    flat = alloc_timestep(T * B, m, N, device, with_obs_grid=with_obs_grid, aggregate=aggregate)
    traj = {k: v.view(T, B, *v.shape[1:]) for k, v in flat.items()}
    traj["action"] = torch.empty((T, B, N), dtype=torch.int32, device=device)
    return traj


def connector_rollout_record(state: StateBufs, policy_keys: torch.Tensor, t0: int, n_steps: int,
                             time_limit: int = 50, timestep_reward: float = -0.03,
                             connected_reward: float = 0.1, auto_reset: bool = True,
                             state_out: Optional[StateBufs] = None,
                             trajectory: Optional[TimestepBufs] = None,
                             stats: Optional[dict] = None, with_obs_grid: bool = True,
                             epochs: Optional["EpochStash"] = None,
                             gen: Optional[GeneratorSpec] = None):
    """connector_rollout that also records every step's timestep and action: returns
    (state_out, stats, trajectory) with trajectory[field] of shape [n_steps, B, ...] -- row s is
    what the s-th `VmapAutoResetWrapper(env).step` of the loop returns."""
    B, m, _ = state["grid"].shape
    N = state["start"].shape[1]
    dev = state["grid"].device
    T = int(n_steps)
    policy_keys = check_keys(policy_keys)
    if policy_keys.shape[0] != B:
        raise ValueError("policy_keys and state disagree on B")
    so = state_out if state_out is not None else alloc_state(B, m, N, dev)
    st = stats if stats is not None else {}
    for k in ROLLOUT_STATS:
        if k in st:
            _check(st[k], torch.int32, (B,), f"stats.{k}")
    traj = trajectory if trajectory is not None else alloc_trajectory(T, B, m, N, dev, with_obs_grid)
    flat = {}
    for k, v in traj.items():
        if tuple(v.shape[:2]) != (T, B) or not v.is_contiguous():
            raise ValueError(f"trajectory.{k} must be contiguous with shape [n_steps, B, ...]")
        flat[k] = v.view(T * B, *v.shape[2:])
    action = flat.pop("action", None)
    if action is not None:
        _check(action, torch.int32, (T * B, N), "trajectory.action")
    si, s_o = _state_struct(state, B, m, N), _state_struct(so, B, m, N)
    sp = nat.RolloutStats(*[_ptr(st.get(k)) for k in ROLLOUT_STATS])
    p = nat.EnvParams(int(time_limit), float(timestep_reward), float(connected_reward),
                      int(bool(auto_reset)))
    tp = _timestep_struct(flat, T * B, m, N)
    use_stash = epochs is not None and auto_reset
    if use_stash:
        if (epochs.B, epochs.m, epochs.N) != (B, m, N):
            raise ValueError("epochs was built for another batch / grid / agent count")
        epochs.before_launch()
    rc = nat.lib().rbg_connector_rollout_gen(
        _gen_ref(gen), B, m, N, C.byref(si), C.byref(s_o), policy_keys.data_ptr(),
        int(t0) & 0xFFFFFFFF, T,
        C.byref(p), C.byref(sp), None, C.byref(tp), _ptr(action),
        C.byref(epochs.struct) if use_stash else None, epochs.tick if use_stash else 0,
        epochs.ready_delay() if use_stash else 0, _stream(dev))
    if use_stash:
        epochs.after_launch()
    nat.check(rc, "rbg_connector_rollout_record")
    return so, st, traj


class RandomPolicy:
    """The masked-uniform random policy (rbg_random_actions; stands in for [3P] jumanji's
    RandomAgent) with its launch bound once: `policy(t, action_mask)` -> action int32 [B, N] in a
    buffer of its own (two, used alternately).  The checked one-off form is random_actions()."""

    def __init__(self, keys: torch.Tensor, num_agents: int) -> None:
        self.keys = check_keys(keys)
        self.B, self.N, self.device = keys.shape[0], int(num_agents), keys.device
        self._out = [torch.empty((self.B, self.N), dtype=torch.int32, device=self.device)
                     for _ in range(2)]
        self._i = 0
        self._fn = nat.lib().rbg_random_actions
        self._kp = self.keys.data_ptr()
        self._mask_shape = torch.Size((self.B, self.N, 5))
        idx = self.device.index
        self._dev_index = idx if idx is not None else torch.cuda.current_device()

    def __call__(self, t: int, action_mask: torch.Tensor) -> torch.Tensor:
        if (action_mask.dtype is not torch.bool or action_mask.shape != self._mask_shape
                or not action_mask.is_contiguous() or action_mask.device != self.device):
            _check(action_mask, torch.bool, (self.B, self.N, 5), "action_mask")
        self._i ^= 1
        out = self._out[self._i]
        rc = self._fn(self._kp, self.B, self.N, int(t) & 0xFFFFFFFF, action_mask.data_ptr(),
                      out.data_ptr(), torch._C._cuda_getCurrentRawStream(self._dev_index))
        if rc:
            nat.check(rc, "rbg_random_actions")
        return out


def random_actions(keys: torch.Tensor, t: int, action_mask: torch.Tensor,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    keys = check_keys(keys)
    B, N, _ = action_mask.shape
    _check(action_mask, torch.bool, (B, N, 5), "action_mask")
    if keys.shape[0] != B:
        raise ValueError("keys and action_mask disagree on B")
    act = out if out is not None else torch.empty((B, N), dtype=torch.int32, device=keys.device)
    _check(act, torch.int32, (B, N), "action")
    nat.check(nat.lib().rbg_random_actions(keys.data_ptr(), B, N, int(t) & 0xFFFFFFFF,
                                           action_mask.data_ptr(), act.data_ptr(),
                                           _stream(keys.device)), "rbg_random_actions")
    return act
