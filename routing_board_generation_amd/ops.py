"""Tensor-level launchers: torch device tensors in, C-ABI calls on torch's current HIP stream.

PyTorch is used for device memory and streams only; every value below is produced by the HIP
kernels of csrc/prw_kernels.hip.  All functions are batched ([B, ...]) and asynchronous.
"""
import ctypes as C
from typing import Dict, Optional

import torch

from . import _native as nat

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


def _gen_ref(gen: Optional["GeneratorSpec"], device=None):
    """Pointer argument for the *_gen entry points (None = ParallelRandomWalk).  `device` = the
    device of the batch the launch works on: a generator that points to device memory (the
    offline dataset's heads / targets) must live there -- the kernels dereference its raw
    pointers, and a dataset built on another rank's GPU would fault or read through peer access."""
    if gen is None or gen.kind == nat.GEN_PARALLEL_RW:
        return None
    if device is not None and gen.device is not None and gen.device != torch.device(device):
        raise ValueError(f"the generator's dataset lives on {gen.device}, the env batch on "
                         f"{torch.device(device)}: build the dataset generator with device=...")
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
    nat.check(nat.lib().rbg_connector_reset_gen(_gen_ref(gen, dev), keys.data_ptr(), B, m, N,
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
    nat.check(nat.lib().rbg_connector_step_gen(_gen_ref(gen, dev), B, m, N, C.byref(si), C.byref(s_o),
                                               action.data_ptr(), None, 0, C.byref(p),
                                               C.byref(t), None, 0, 0, _stream(dev)),
              "rbg_connector_step")
    return so, ts


# ------------------------------------------------------------------------------------- prefetch
def alloc_stash(B: int, m: int, N: int, device, depth: int = 1, rolling: bool = False) -> dict:
    """Buffers of include/rbg_prw.h::rbg_stash (tag / ready_at / refill_key / hits zeroed);
    `depth` slots per env, slot s of env b in row b * depth + s.  rolling: also the in-flight marks
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
        _gen_ref(gen, dev), keys.data_ptr(), B, m, N, C.byref(s), C.byref(t), C.byref(h), int(tick),
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
        _gen_ref(gen, dev), B, m, N, C.byref(si), C.byref(s_o), action.data_ptr(), None, 0, C.byref(p),
        C.byref(t), C.byref(h), int(tick), int(ready_delay), _stream(dev)),
        "rbg_connector_step_prefetch")
    return so, ts


def stash_refill(stash: dict, m: int, gen: Optional[GeneratorSpec] = None) -> None:
    """Launch the refill on torch's current stream."""
    B, N = stash["hits"].shape[0], stash["next"]["start"].shape[1]
    dev = stash["tag"].device
    h = _stash_struct(stash, B, m, N)
    nat.check(nat.lib().rbg_stash_refill_gen(_gen_ref(gen, dev), B, m, N, C.byref(h), _stream(dev)),
              "rbg_stash_refill")


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
        _gen_ref(gen, dev), B, m, N, C.byref(si), C.byref(s_o), policy_keys.data_ptr(),
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
        _gen_ref(gen, dev), B, m, N, C.byref(si), C.byref(s_o), policy_keys.data_ptr(),
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


# the stepping loops and the stash protocol live in session.py; their names stay reachable here
from .session import (ConnectorSession, EpochStash, GroupedStepLoop, StepLoop,  # noqa: E402,F401
                      generator_streams)
