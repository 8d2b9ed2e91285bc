"""ctypes front-end of the CPU oracle (oracle/prw_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under routing_board_generation_amd/ may import this module.

All arrays are numpy, C-contiguous; shapes mirror the batched HBM layout of the product
(see DESIGN.md): grid [B,m,m] i32, start/target/position [B,N,2] i32, key [B,2] u32, ...
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RBG_ORACLE_LIB: another build of the same sources (the sanitizer build of `make asan-check`)
_OVERRIDE = os.environ.get("RBG_ORACLE_LIB")
_LIB_PATH = _OVERRIDE or os.path.join(_HERE, "libprw_oracle.so")


def build(force: bool = False) -> str:
    if _OVERRIDE:
        return _LIB_PATH
    srcs = [os.path.join(_HERE, f) for f in ("prw_oracle.c", "seed_extension.c", "prw_philox.c", "Makefile")]
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libprw_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_uniform.restype = C.c_float
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u32(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.uint32))


def _i32(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.int32))


class AgentT(C.Structure):
    _fields_ = [("id", C.c_int32), ("start", C.c_int32 * 2), ("target", C.c_int32 * 2),
                ("position", C.c_int32 * 2)]


def _agents_to_c(agents):
    """agents: dict(id[N], start[N,2], target[N,2], position[N,2]) -> (AgentT * N)."""
    n = len(agents["id"])
    arr = (AgentT * n)()
    for i in range(n):
        arr[i].id = int(agents["id"][i])
        for d in range(2):
            arr[i].start[d] = int(agents["start"][i][d])
            arr[i].target[d] = int(agents["target"][i][d])
            arr[i].position[d] = int(agents["position"][i][d])
    return arr


def _agents_from_c(arr):
    n = len(arr)
    return {
        "id": np.array([arr[i].id for i in range(n)], np.int32),
        "start": np.array([[arr[i].start[0], arr[i].start[1]] for i in range(n)], np.int32),
        "target": np.array([[arr[i].target[0], arr[i].target[1]] for i in range(n)], np.int32),
        "position": np.array([[arr[i].position[0], arr[i].position[1]] for i in range(n)], np.int32),
    }


# ------------------------------------------------------------------------------------ PRNG
def prng_key(seed: int) -> np.ndarray:
    k = np.zeros(2, np.uint32)
    lib().oracle_prng_key(C.c_uint64(seed), _p(k))
    return k


def threefry2x32(key, x0, x1):
    o0, o1 = C.c_uint32(), C.c_uint32()
    lib().oracle_threefry2x32(C.c_uint32(int(key[0])), C.c_uint32(int(key[1])), C.c_uint32(x0),
                              C.c_uint32(x1), C.byref(o0), C.byref(o1))
    return o0.value, o1.value


def split(key, num: int = 2) -> np.ndarray:
    key = _u32(key)
    out = np.zeros((num, 2), np.uint32)
    lib().oracle_split(_p(key), num, _p(out))
    return out


def random_bits(key, n: int) -> np.ndarray:
    key = _u32(key)
    out = np.zeros(n, np.uint32)
    lib().oracle_random_bits(_p(key), n, _p(out))
    return out


def uniform(key) -> np.float32:
    key = _u32(key)
    return np.float32(lib().oracle_uniform(_p(key)))


def permutation(key, n: int) -> np.ndarray:
    key = _u32(key)
    out = np.zeros(n, np.int32)
    lib().oracle_permutation(_p(key), n, _p(out))
    return out


# ------------------------------------------------------------------------------------ pure helpers
def adjacent_cells(cell, rows, cols):
    out = np.zeros(4, np.int32)
    lib().oracle_adjacent_cells(int(cell), rows, cols, _p(out))
    return out


def available_cells(grid, cell):
    grid = _i32(grid)
    out = np.zeros(4, np.int32)
    lib().oracle_available_cells(_p(grid), grid.shape[0], grid.shape[1], int(cell), _p(out))
    return out


def is_cell_free(grid, cell):
    grid = _i32(grid)
    return bool(lib().oracle_is_cell_free(_p(grid), grid.shape[0], grid.shape[1], int(cell)))


def action_from_positions(p1, p2, cols):
    return int(lib().oracle_action_from_positions(int(p1), int(p2), cols))


def continue_stepping(grid, agents):
    grid = _i32(grid)
    arr = _agents_to_c(agents)
    return bool(lib().oracle_continue_stepping(_p(grid), arr, grid.shape[0], len(arr)))


def step_agent(agent_i, agents, grid, action):
    """vmapped element of _step_agent: returns (new_agent_dict_row, new_grid)."""
    grid = _i32(grid)
    arr = _agents_to_c(agents)
    new_agent = AgentT()
    new_grid = np.zeros_like(grid)
    lib().oracle_step_agent(C.byref(arr[agent_i]), _p(grid), grid.shape[0], int(action),
                            C.byref(new_agent), _p(new_grid))
    return ([new_agent.position[0], new_agent.position[1]], new_grid)


def not_doubling_back(grid, wire_id, cell):
    grid = _i32(grid)
    return bool(lib().oracle_is_cell_not_doubling_back(_p(grid), grid.shape[0], grid.shape[1],
                                                       int(wire_id), int(cell)))


def select_action(key, grid, agents, i):
    """ParallelRandomWalkBoard._select_action(key, grid, agents[i])."""
    key = _u32(key)
    grid = _i32(grid)
    arr = _agents_to_c(agents)
    return int(lib().oracle_select_action(_p(key), _p(grid), grid.shape[0], C.byref(arr[i])))


def initialise_agents(key, m, N, grid=None):
    key = _u32(key)
    grid = np.zeros((m, m), np.int32) if grid is None else _i32(grid).copy()
    arr = (AgentT * N)()
    lib().oracle_initialise_agents(_p(key), _p(grid), arr, m, N)
    return grid, _agents_from_c(arr)


def prw_step(key, grid, agents):
    """ParallelRandomWalkBoard._step: returns (next_key, grid, agents)."""
    key = _u32(key).copy()
    grid = _i32(grid).copy()
    arr = _agents_to_c(agents)
    lib().oracle_prw_step(_p(key), _p(grid), arr, grid.shape[0], len(arr))
    return key, grid, _agents_from_c(arr)


def step_agents_with_actions(grid, agents, actions):
    grid = _i32(grid).copy()
    arr = _agents_to_c(agents)
    actions = _i32(actions)
    lib().oracle_step_agents_with_actions(_p(grid), arr, grid.shape[0], len(arr), _p(actions))
    return grid, _agents_from_c(arr)


def generate_board(key, m, N):
    """ParallelRandomWalkBoard.generate_board -> heads [N,2], targets [N,2], solved [m,m], iters."""
    key = _u32(key)
    heads = np.zeros((N, 2), np.int32)
    targets = np.zeros((N, 2), np.int32)
    solved = np.zeros((m, m), np.int32)
    it = lib().oracle_generate_board(_p(key), m, N, _p(heads), _p(targets), _p(solved))
    return heads, targets, solved, int(it)


# ------------------------------------------------------------------------------------ batched
def alloc_state(B, m, N):
    return {
        "grid": np.zeros((B, m, m), np.int32),
        "step_count": np.zeros((B,), np.int32),
        "start": np.zeros((B, N, 2), np.int32),
        "target": np.zeros((B, N, 2), np.int32),
        "position": np.zeros((B, N, 2), np.int32),
        "key": np.zeros((B, 2), np.uint32),
    }


def alloc_timestep(B, m, N):
    return {
        "step_type": np.zeros((B,), np.int8),
        "reward": np.zeros((B, N), np.float32),
        "discount": np.zeros((B, N), np.float32),
        "obs_grid": np.zeros((B, N, m, m), np.int32),
        "action_mask": np.zeros((B, N, 5), np.uint8),
        "obs_step_count": np.zeros((B,), np.int32),
        "num_connections": np.zeros((B,), np.int32),
        "ratio_connections": np.zeros((B,), np.float32),
        "total_path_length": np.zeros((B,), np.int32),
    }


_STATE_ORDER = ("grid", "step_count", "start", "target", "position", "key")
_TS_ORDER = ("step_type", "reward", "discount", "obs_grid", "action_mask", "obs_step_count",
             "num_connections", "ratio_connections", "total_path_length")


def generate_batch(keys, m, N, generator="parallel_random_walk", threads=1, want_solved=True):
    """Generator.__call__ vmapped over keys [B,2] -> (state dict, solved [B,m,m], iters [B])."""
    keys = _u32(keys).reshape(-1, 2)
    B = keys.shape[0]
    st = alloc_state(B, m, N)
    solved = np.zeros((B, m, m), np.int32) if want_solved else None
    iters = np.zeros((B,), np.int32)
    gen = {"parallel_random_walk": 0, "uniform": 1}[generator]
    rc = lib().oracle_generate_batch(_p(keys), B, m, N, gen, *[_p(st[k]) for k in _STATE_ORDER],
                                     _p(solved), _p(iters), threads)
    if rc != 0:
        raise ValueError("oracle_generate_batch: bad arguments")
    return st, solved, iters


def generate_board_batch(keys, m, N, threads=1):
    keys = _u32(keys).reshape(-1, 2)
    B = keys.shape[0]
    heads = np.zeros((B, N, 2), np.int32)
    targets = np.zeros((B, N, 2), np.int32)
    solved = np.zeros((B, m, m), np.int32)
    iters = np.zeros((B,), np.int32)
    rc = lib().oracle_generate_board_batch(_p(keys), B, m, N, _p(heads), _p(targets), _p(solved),
                                           _p(iters), threads)
    if rc != 0:
        raise ValueError("oracle_generate_board_batch: bad arguments")
    return heads, targets, solved, iters


def connector_reset_batch(keys, m, N, threads=1):
    keys = _u32(keys).reshape(-1, 2)
    B = keys.shape[0]
    st, ts = alloc_state(B, m, N), alloc_timestep(B, m, N)
    rc = lib().oracle_connector_reset_batch(_p(keys), B, m, N, *[_p(st[k]) for k in _STATE_ORDER],
                                            *[_p(ts[k]) for k in _TS_ORDER], threads)
    if rc != 0:
        raise ValueError("oracle_connector_reset_batch: bad arguments")
    return st, ts


def connector_observe_batch(st, threads=1):
    B, m, _ = st["grid"].shape
    N = st["start"].shape[1]
    ts = alloc_timestep(B, m, N)
    rc = lib().oracle_connector_observe_batch(B, m, N, *[_p(st[k]) for k in _STATE_ORDER],
                                              *[_p(ts[k]) for k in _TS_ORDER], threads)
    if rc != 0:
        raise ValueError("oracle_connector_observe_batch: bad arguments")
    return ts


def connector_step_batch(st, action, time_limit=50, timestep_reward=-0.03, connected_reward=0.1,
                         auto_reset=False, threads=1):
    """Steps `st` IN PLACE (pass a copy to keep the old state) and returns the timestep dict."""
    B, m, _ = st["grid"].shape
    N = st["start"].shape[1]
    action = _i32(action).reshape(B, N)
    ts = alloc_timestep(B, m, N)
    rc = lib().oracle_connector_step_batch(
        B, m, N, *[_p(st[k]) for k in _STATE_ORDER], _p(action), int(time_limit),
        C.c_float(timestep_reward), C.c_float(connected_reward), int(bool(auto_reset)),
        *[_p(ts[k]) for k in _TS_ORDER], threads)
    if rc != 0:
        raise ValueError("oracle_connector_step_batch: bad arguments")
    return ts


def random_actions_batch(keys, t, action_mask):
    keys = _u32(keys).reshape(-1, 2)
    action_mask = np.ascontiguousarray(action_mask, dtype=np.uint8)
    B, N, _ = action_mask.shape
    action = np.zeros((B, N), np.int32)
    lib().oracle_random_actions_batch(_p(keys), B, N, C.c_uint32(t), _p(action_mask), _p(action))
    return action


def copy_state(st):
    return {k: v.copy() for k, v in st.items()}


ROLLOUT_STATS = ("episodes", "connections_made", "unconnected_agent_steps", "final_connections",
                 "final_path_length")


def connector_rollout_batch(st, policy_keys, t0, n_steps, time_limit=50, timestep_reward=-0.03,
                            connected_reward=0.1, auto_reset=True, threads=1):
    """n_steps x { random_actions_batch; connector_step_batch } with the per-env counters of
    rbg_rollout_stats.  Steps `st` IN PLACE; returns (stats dict, timestep of the last step)."""
    B = st["grid"].shape[0]
    stats = {k: np.zeros((B,), np.int32) for k in ROLLOUT_STATS}
    mask = connector_observe_batch(st, threads=threads)["action_mask"]
    ts = None
    for s in range(n_steps):
        act = random_actions_batch(policy_keys, t0 + s, mask)
        was = (st["position"] == st["target"]).all(-1)
        probe = copy_state(st)  # the stepped state BEFORE any auto-reset, for the counters
        connector_step_batch(probe, act, time_limit, timestep_reward, connected_reward,
                             auto_reset=False, threads=threads)
        now = (probe["position"] == probe["target"]).all(-1)
        ts = connector_step_batch(st, act, time_limit, timestep_reward, connected_reward,
                                  auto_reset=auto_reset, threads=threads)
        last = ts["step_type"] == 2
        stats["connections_made"] += (~was & now).sum(-1).astype(np.int32)
        stats["unconnected_agent_steps"] += (~was).sum(-1).astype(np.int32)
        stats["episodes"] += last.astype(np.int32)
        stats["final_connections"] += np.where(last, ts["num_connections"], 0).astype(np.int32)
        stats["final_path_length"] += np.where(last, ts["total_path_length"], 0).astype(np.int32)
        mask = ts["action_mask"]
    return stats, ts


def randint(key, lo, hi):
    """[3P] jax.random.randint(key, (), lo, hi) (jax/_src/random.py::_randint), 32-bit path."""
    k = split(key, 2)
    hi_bits = int(random_bits(k[0], 1)[0])
    lo_bits = int(random_bits(k[1], 1)[0])
    span = (hi - lo) & 0xFFFFFFFF
    span = 1 if hi <= lo else span
    mult = (65536 % span)
    mult = ((mult * mult) & 0xFFFFFFFF) % span
    off = (((hi_bits % span) * mult) & 0xFFFFFFFF) + (lo_bits % span)
    return lo + ((off & 0xFFFFFFFF) % span)


def dataset_sample_batch(keys, heads, targets, m):
    """BoardDatasetGeneratorJAX.__call__ vmapped: -> (state dict, which [B])."""
    keys = _u32(keys).reshape(-1, 2)
    B, N = keys.shape[0], heads.shape[1]
    st = alloc_state(B, m, N)
    which = np.zeros((B,), np.int32)
    for b in range(B):
        k0 = split(keys[b])[0]
        w = randint(k0, 0, heads.shape[0])
        which[b] = w
        st["key"][b] = k0
        st["start"][b] = heads[w]
        st["target"][b] = targets[w]
        st["position"][b] = heads[w]
        for i in range(N):
            st["grid"][b, heads[w, i, 0], heads[w, i, 1]] = 2 + 3 * i
        for i in range(N):
            st["grid"][b, targets[w, i, 0], targets[w, i, 1]] = 3 + 3 * i
    return st, which


def seq_rw_generator_batch(keys, m, N, threads=1):
    """SequentialRandomWalkGenerator.__call__ vmapped -> (state, solved [B,m,m], attempts [B])."""
    keys = _u32(keys).reshape(-1, 2)
    B = keys.shape[0]
    st = alloc_state(B, m, N)
    solved = np.zeros((B, m, m), np.int32)
    attempts = np.zeros((B,), np.int32)
    rc = lib().oracle_seq_rw_generator_batch(_p(keys), B, m, N, *[_p(st[k]) for k in _STATE_ORDER],
                                             _p(solved), _p(attempts), threads)
    if rc != 0:
        raise ValueError("oracle_seq_rw_generator_batch: bad arguments")
    return st, solved, attempts


def seq_rw_rounding_audit(keys, m, N, threads=1):
    """Float32 rounding audit of the integer Gumbel-top-1 rule over the SequentialRandomWalk
    generator (prw_oracle.c::oracle_seq_rw_rounding_audit) -> dict of per-board int32 arrays."""
    keys = _u32(keys).reshape(-1, 2)
    B = keys.shape[0]
    out = {k: np.zeros((B,), np.int32) for k in ("draws", "differs", "within_4ulp", "min_ulps")}
    rc = lib().oracle_seq_rw_rounding_audit(_p(keys), B, m, N, *[_p(v) for v in out.values()],
                                            threads)
    if rc != 0:
        raise ValueError("oracle_seq_rw_rounding_audit: bad arguments")
    return out


def randint_vec(key, n, lo, hi):
    """[3P] jax.random.randint(key, (n,), lo, hi), the C restatement used by SeedExtension."""
    key = _u32(key)
    out = np.zeros(n, np.int32)
    lib().oracle_randint(_p(key), n, int(lo), int(hi), _p(out))
    return out


def seed_extension_batch(keys, m, N, split_first=True, randomness=0.0, two_sided=True,
                         extension_iterations=1, extension_steps=1e23, threads=1,
                         return_live=False):
    """vmap(SeedExtensionGenerator(m, N).__call__) (split_first) or
    vmap(SeedExtensionBoard(m, m, N).generate_starts_ends) (not split_first) over keys [B,2]
    -> (state dict, solved [B,m,m], sweeps [B]) (+ live sweeps [B] with return_live: the sweeps
    that changed the board, which is what the HIP kernel runs and reports)."""
    keys = _u32(keys).reshape(-1, 2)
    B = keys.shape[0]
    st = alloc_state(B, m, N)
    solved = np.zeros((B, m, m), np.int32)
    sweeps = np.zeros((B,), np.int32)
    live = np.zeros((B,), np.int32)
    rc = lib().oracle_seed_extension_batch(
        _p(keys), B, m, N, int(bool(split_first)), C.c_float(randomness), int(bool(two_sided)),
        int(extension_iterations), C.c_double(extension_steps),
        *[_p(st[k]) for k in _STATE_ORDER], _p(solved), _p(sweeps), _p(live), threads)
    if rc != 0:
        raise ValueError("oracle_seed_extension_batch: bad arguments (rc=%d)" % rc)
    return (st, solved, sweeps, live) if return_live else (st, solved, sweeps)


def philox4x32_10(key, ctr) -> np.ndarray:
    key, ctr = _u32(key), _u32(ctr)
    out = np.zeros(4, np.uint32)
    lib().oracle_philox4x32_10(_p(key), _p(ctr), _p(out))
    return out


def prw_philox_batch(keys, m, N, threads=1):
    """vmap(ParallelRandomWalkGenerator(m, N, stream="philox").__call__) over keys [B,2]
    -> (state dict, solved [B,m,m], iters [B]).  Project-defined stream (oracle/prw_philox.c)."""
    keys = _u32(keys).reshape(-1, 2)
    B = keys.shape[0]
    st = alloc_state(B, m, N)
    solved = np.zeros((B, m, m), np.int32)
    iters = np.zeros((B,), np.int32)
    rc = lib().oracle_prw_philox_batch(_p(keys), B, m, N, *[_p(st[k]) for k in _STATE_ORDER],
                                       _p(solved), _p(iters), threads)
    if rc != 0:
        raise ValueError("oracle_prw_philox_batch: bad arguments")
    return st, solved, iters
