"""ctypes binding of librbg_prw.so (include/rbg_prw.h) -- the only compute path of this package.

There is deliberately NO fallback: if the HIP library is missing or a launch fails, the call
raises.  Build the library with `python __graft_entry__.py` (or `build_native()` below); the
built .so lives in-tree under routing_board_generation_amd/lib/.
"""
import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
LIB_PATH = os.path.join(_PKG, "lib", "librbg_prw.so")
SRC_PATH = os.path.join(_PKG, "csrc", "prw_kernels.hip")
HEADER_PATH = os.path.join(_ROOT, "include", "rbg_prw.h")

ABI_VERSION = 9
# -amdgpu-sched-strategy=max-ilp: the walk's straggler boards run one wave per SIMD, where an
# instruction that depends on its predecessor issues after ~8 cycles and an independent one after
# ~4 (tools/microbench/dep_latency.hip); the default scheduler (occupancy-first) leaves the
# availability compares and the Threefry rounds as dependent runs (35.3 -> 31.7 us)
# (an internal LLVM option: OPTIONAL_FLAGS are dropped, with a warning, by a hipcc that rejects
# them -- they change scheduling, never results; -ffp-contract=off is mandatory)
REQUIRED_FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-ffp-contract=off",
                  "-std=c++17"]
OPTIONAL_FLAGS = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
HIPCC_FLAGS = REQUIRED_FLAGS + OPTIONAL_FLAGS


class NativeLibraryError(RuntimeError):
    pass


class State(C.Structure):
    _fields_ = [("grid", C.c_void_p), ("step_count", C.c_void_p), ("start", C.c_void_p),
                ("target", C.c_void_p), ("position", C.c_void_p), ("key", C.c_void_p)]


class Timestep(C.Structure):
    _fields_ = [("step_type", C.c_void_p), ("reward", C.c_void_p), ("discount", C.c_void_p),
                ("obs_grid", C.c_void_p), ("action_mask", C.c_void_p),
                ("obs_step_count", C.c_void_p), ("num_connections", C.c_void_p),
                ("ratio_connections", C.c_void_p), ("total_path_length", C.c_void_p),
                ("reward_total", C.c_void_p), ("discount_max", C.c_void_p)]


class RolloutStats(C.Structure):
    _fields_ = [("episodes", C.c_void_p), ("connections_made", C.c_void_p),
                ("unconnected_agent_steps", C.c_void_p), ("final_connections", C.c_void_p),
                ("final_path_length", C.c_void_p)]


class Stash(C.Structure):
    _fields_ = [("next", State), ("tag", C.c_void_p), ("ready_at", C.c_void_p),
                ("refill_key", C.c_void_p), ("hits", C.c_void_p), ("misses", C.c_void_p),
                ("scratch", C.c_void_p), ("inflight", C.c_void_p), ("far_key", C.c_void_p),
                ("depth", C.c_int32)]


class Generator(C.Structure):
    """include/rbg_prw.h::rbg_generator"""
    _fields_ = [("kind", C.c_int32), ("randomness", C.c_float), ("two_sided", C.c_int32),
                ("extension_iterations", C.c_int32), ("max_extension_steps", C.c_int32),
                ("n_boards", C.c_int32), ("heads", C.c_void_p), ("targets", C.c_void_p)]


(GEN_PARALLEL_RW, GEN_UNIFORM, GEN_SEQUENTIAL_RW, GEN_SEED_EXTENSION, GEN_DATASET,
 GEN_PARALLEL_RW_PHILOX) = range(6)


class EnvParams(C.Structure):
    _fields_ = [("time_limit", C.c_int32), ("timestep_reward", C.c_float),
                ("connected_reward", C.c_float), ("auto_reset", C.c_int32)]


class StepCall(C.Structure):
    """include/rbg_prw.h::rbg_step_call"""
    _fields_ = [("generator", C.POINTER(Generator)), ("B", C.c_int32), ("m", C.c_int32),
                ("N", C.c_int32), ("tick", C.c_int32), ("ready_delay", C.c_int32),
                ("policy_t", C.c_uint32), ("state_in", State), ("state_out", State),
                ("timestep", Timestep), ("action", C.c_void_p), ("policy_keys", C.c_void_p),
                ("params", EnvParams), ("stash", C.POINTER(Stash)), ("stream", C.c_void_p)]


class StashRollCall(C.Structure):
    """include/rbg_prw.h::rbg_stash_roll_call"""
    _fields_ = [("generator", C.POINTER(Generator)), ("B", C.c_int32), ("m", C.c_int32),
                ("N", C.c_int32), ("claim_ready", C.c_int32), ("skip_ready", C.c_int32),
                ("stamp", C.c_uint32), ("stash", C.POINTER(Stash)), ("step_stream", C.c_void_p),
                ("gen_stream", C.c_void_p), ("launched", C.c_void_p), ("await_done", C.c_void_p),
                ("after_listed", C.c_void_p), ("listed", C.c_void_p), ("done", C.c_void_p),
                ("miss_acc", C.c_void_p), ("miss_word", C.c_void_p), ("miss_host", C.c_void_p)]


# name -> (restype, argtypes); mirrors include/rbg_prw.h one to one
SIGNATURES = {
    "rbg_abi_version": (C.c_int, []),
    "rbg_version": (C.c_char_p, []),
    "rbg_last_error": (C.c_char_p, []),
    "rbg_split": (C.c_int, [C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]),
    "rbg_split_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "rbg_prw_generate": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    "rbg_prw_generate_philox": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rbg_seq_rw_generate": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "rbg_seed_extension_generate": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rbg_uniform_generate": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "rbg_dataset_sample": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p, C.POINTER(State), C.c_void_p, C.c_void_p]),
    "rbg_connector_reset": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(State),
                                      C.POINTER(Timestep), C.c_void_p]),
    "rbg_connector_reset_prefetch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int,
                                               C.POINTER(State), C.POINTER(Timestep),
                                               C.POINTER(Stash), C.c_int, C.c_int, C.c_void_p]),
    "rbg_connector_step_prefetch": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(State),
                                              C.POINTER(State), C.c_void_p, C.POINTER(EnvParams),
                                              C.POINTER(Timestep), C.POINTER(Stash), C.c_int,
                                              C.c_int, C.c_void_p]),
    "rbg_stash_refill": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(Stash), C.c_void_p]),
    "rbg_connector_observe": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(State),
                                        C.POINTER(Timestep), C.c_void_p]),
    "rbg_connector_step": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(State),
                                     C.POINTER(State), C.c_void_p, C.POINTER(EnvParams),
                                     C.POINTER(Timestep), C.c_void_p]),
    "rbg_connector_rollout": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(State),
                                        C.POINTER(State), C.c_void_p, C.c_uint32, C.c_int,
                                        C.POINTER(EnvParams), C.POINTER(RolloutStats),
                                        C.POINTER(Timestep), C.c_void_p]),
    "rbg_connector_rollout_prefetch": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(State),
                                                 C.POINTER(State), C.c_void_p, C.c_uint32,
                                                 C.c_int, C.POINTER(EnvParams),
                                                 C.POINTER(RolloutStats), C.POINTER(Timestep),
                                                 C.POINTER(Stash), C.c_int, C.c_int, C.c_void_p]),
    "rbg_connector_rollout_record": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(State),
                                               C.POINTER(State), C.c_void_p, C.c_uint32, C.c_int,
                                               C.POINTER(EnvParams), C.POINTER(RolloutStats),
                                               C.POINTER(Timestep), C.POINTER(Timestep),
                                               C.c_void_p, C.POINTER(Stash), C.c_int, C.c_int,
                                               C.c_void_p]),
    "rbg_connector_reset_gen": (C.c_int, [C.POINTER(Generator), C.c_void_p, C.c_int, C.c_int,
                                          C.c_int, C.POINTER(State), C.POINTER(Timestep),
                                          C.POINTER(Stash), C.c_int, C.c_int, C.c_void_p]),
    "rbg_connector_step_gen": (C.c_int, [C.POINTER(Generator), C.c_int, C.c_int, C.c_int,
                                         C.POINTER(State), C.POINTER(State), C.c_void_p,
                                         C.c_void_p, C.c_uint32, C.POINTER(EnvParams),
                                         C.POINTER(Timestep), C.POINTER(Stash), C.c_int, C.c_int,
                                         C.c_void_p]),
    "rbg_connector_step_call": (C.c_int, [C.POINTER(StepCall)]),
    "rbg_stash_list": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(Stash), C.c_int, C.c_int,
                                 C.c_void_p]),
    "rbg_stash_roll": (C.c_int, [C.POINTER(StashRollCall)]),
    "rbg_stash_refill_listed": (C.c_int, [C.POINTER(Generator), C.c_int, C.c_int, C.c_int,
                                          C.POINTER(Stash), C.c_void_p]),
    "rbg_stash_refill_gen": (C.c_int, [C.POINTER(Generator), C.c_int, C.c_int, C.c_int,
                                       C.POINTER(Stash), C.c_void_p]),
    "rbg_connector_rollout_gen": (C.c_int, [C.POINTER(Generator), C.c_int, C.c_int, C.c_int,
                                            C.POINTER(State), C.POINTER(State), C.c_void_p,
                                            C.c_uint32, C.c_int, C.POINTER(EnvParams),
                                            C.POINTER(RolloutStats), C.POINTER(Timestep),
                                            C.POINTER(Timestep), C.c_void_p, C.POINTER(Stash),
                                            C.c_int, C.c_int, C.c_void_p]),
    "rbg_random_actions": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "rbg_prw_initialise_agents": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int] +
                                  [C.c_void_p] * 4),
    "rbg_prw_step": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 8),
    "rbg_prw_select_action": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 4),
    "rbg_prw_cell_queries": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int] +
                             [C.c_void_p] * 6),
    "rbg_prw_step_agent": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 7),
}


def hipcc_compile(extra_args, flags=None) -> None:
    """hipcc with HIPCC_FLAGS (or `flags`) + extra_args; if that fails and OPTIONAL_FLAGS were in
    it, once more without them (a scheduling preference must not break the build)."""
    import warnings

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = list(HIPCC_FLAGS if flags is None else flags)
    try:
        subprocess.check_call([hipcc] + flags + list(extra_args))
    except subprocess.CalledProcessError:
        plain = [f for f in flags if f not in OPTIONAL_FLAGS]
        if plain == flags:
            raise
        warnings.warn(f"hipcc failed with {OPTIONAL_FLAGS}; building without them "
                      "(same results, the walk launch is ~10 % slower)")
        subprocess.check_call([hipcc] + plain + list(extra_args))


def build_native(force: bool = False) -> str:
    """Compile csrc/prw_kernels.hip for gfx950 into lib/librbg_prw.so (hipcc cross-compiles
    without a GPU)."""
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    csrc = os.path.dirname(SRC_PATH)
    sources = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [HEADER_PATH, __file__]
    stale = (not os.path.exists(LIB_PATH)
             or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(f) for f in sources))
    if force or stale:
        hipcc_compile(["-o", LIB_PATH, SRC_PATH])
    return LIB_PATH


_lib = None


def lib():
    """The loaded library; raises NativeLibraryError when it is missing or of the wrong ABI."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
                "(there is no CPU or PyTorch fallback for this path)")
        try:
            handle = C.CDLL(LIB_PATH)
        except OSError as e:
            raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (restype, argtypes) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise NativeLibraryError(f"{LIB_PATH} does not export {name}") from e
            fn.restype = restype
            fn.argtypes = argtypes
        if handle.rbg_abi_version() != ABI_VERSION:
            raise NativeLibraryError(
                f"ABI mismatch: library {handle.rbg_abi_version()} != binding {ABI_VERSION}")
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().rbg_last_error().decode()
        kind = "argument error" if rc < 0 else f"hipError {rc}"
        raise NativeLibraryError(f"{what} failed ({kind}): {msg}")
