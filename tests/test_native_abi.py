"""CPU-side checks of the C-ABI shared library: it loads, exports every symbol declared in
include/rbg_prw.h, and rejects bad arguments before touching the GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    from routing_board_generation_amd import _native

    _native.build_native()
    return _native


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "rbg_prw.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rbg_[a-z_0-9]+)\s*\(", text)))


def test_header_functions_are_exported_and_bound(native):
    declared = _declared_functions()
    assert len(declared) >= 11
    handle = C.CDLL(native.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in rbg_prw.h but not exported"
    assert sorted(native.SIGNATURES) == declared, "ctypes binding out of sync with the header"


def test_version_and_abi(native):
    lib = native.lib()
    assert lib.rbg_abi_version() == native.ABI_VERSION
    assert b"gfx950" in lib.rbg_version()


def test_struct_layouts_match_header(native, tmp_path):
    assert C.sizeof(native.State) == 6 * C.sizeof(C.c_void_p)
    assert C.sizeof(native.Timestep) == 11 * C.sizeof(C.c_void_p)
    assert C.sizeof(native.EnvParams) == 16
    # every struct of the header as the C compiler lays it out against the ctypes mirror
    import subprocess

    structs = {"rbg_state": native.State, "rbg_timestep": native.Timestep,
               "rbg_env_params": native.EnvParams, "rbg_stash": native.Stash,
               "rbg_rollout_stats": native.RolloutStats, "rbg_generator": native.Generator,
               "rbg_step_call": native.StepCall, "rbg_stash_roll_call": native.StashRollCall}
    lines = []
    for cname, ct in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in ct._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rbg_prw.h"\nint main(void){'
                   + "".join(lines) + "return 0;}")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    got = dict(ln.split() for ln in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, ct in structs.items():
        assert int(got[cname]) == C.sizeof(ct), cname
        for fname, _ in ct._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(ct, fname).offset, f"{cname}.{fname}"


def test_argument_errors_do_not_need_a_gpu(native):
    lib = native.lib()
    assert lib.rbg_prw_generate(None, 1, 16, 10, 1, None, None, None, None, None, None, None, None, None) == -1
    assert b"NULL" in lib.rbg_last_error()
    one = C.c_void_p(16)  # never dereferenced: shape checks come first
    assert lib.rbg_prw_generate(one, 1, 41, 10, 1, None, None, one, one, None, None, None, None, None) == -2
    assert lib.rbg_prw_generate(one, 1, 16, 65, 1, None, None, one, one, None, None, None, None, None) == -2
    assert lib.rbg_prw_generate(one, 1, 3, 10, 1, None, None, one, one, None, None, None, None, None) == -2
    assert lib.rbg_prw_generate(one, -1, 16, 10, 1, None, None, one, one, None, None, None, None, None) == -2
    assert lib.rbg_uniform_generate(one, 1, 4, 9, None, one, one, None, None, None, None) == -2
    assert lib.rbg_split(0, 0, 4, None, None) == -1
    with pytest.raises(native.NativeLibraryError):
        native.check(-2, "rbg_prw_generate")


def test_argument_errors_of_the_generator_and_stash_entry_points(native):
    """The ABI 8 / 9 entry points (rbg_generator, rbg_step_call, rolling refills) check their
    arguments before any launch: no GPU needed, nothing is dereferenced on the device."""
    nat = native
    lib = nat.lib()
    one = 16  # stands for a device pointer; never dereferenced by the checks
    assert lib.rbg_connector_step_call(None) == -1 and b"NULL call" in lib.rbg_last_error()
    call = nat.StepCall()
    call.B, call.m, call.N = 1, 16, 10
    call.params = nat.EnvParams(0, -0.03, 0.1, 1)
    assert lib.rbg_connector_step_call(C.byref(call)) == -3 and b"time_limit" in lib.rbg_last_error()
    call.params = nat.EnvParams(50, -0.03, 0.1, 1)
    call.m = 41
    assert lib.rbg_connector_step_call(C.byref(call)) == -2
    call.m = 16
    assert lib.rbg_connector_step_call(C.byref(call)) == -1  # incomplete state
    # generator descriptors
    state = nat.State(*[one] * 6)
    stash = nat.Stash(state, one, one, one, None, None, None, None, None, 1)
    for gen, m, N, rc, msg in (
            (nat.Generator(99, 0.0, 1, 1, 1, 0, None, None), 16, 10, -3, b"unknown generator kind"),
            (nat.Generator(nat.GEN_UNIFORM, 0.0, 1, 1, 1, 0, None, None), 4, 9, -2, b"2N exceeds"),
            (nat.Generator(nat.GEN_SEED_EXTENSION, 0.0, 1, 1, 1, 0, None, None), 4, 5, -2,
             b"seed cells"),
            (nat.Generator(nat.GEN_SEED_EXTENSION, 0.0, 1, -1, 1, 0, None, None), 16, 10, -3,
             b"iteration"),
            (nat.Generator(nat.GEN_DATASET, 0.0, 1, 1, 1, 0, one, one), 16, 10, -2, b"n_boards"),
            (nat.Generator(nat.GEN_DATASET, 0.0, 1, 1, 1, 7, None, one), 16, 10, -1,
             b"NULL boards")):
        assert lib.rbg_stash_refill_gen(C.byref(gen), 8, m, N, C.byref(stash), None) == rc, msg
        assert msg in lib.rbg_last_error(), lib.rbg_last_error()
    seq = nat.Generator(nat.GEN_SEQUENTIAL_RW, 0.0, 1, 1, 1, 0, None, None)
    assert lib.rbg_stash_refill_gen(C.byref(seq), 8, 2, 1, C.byref(stash), None) == -2  # m >= 3
    # the stash itself
    assert lib.rbg_stash_refill_gen(None, 8, 16, 10, None, None) == -1
    deep = nat.Stash(state, one, one, one, None, None, None, None, None, 13)
    assert lib.rbg_stash_refill_gen(None, 8, 16, 10, C.byref(deep), None) == -3
    assert b"depth" in lib.rbg_last_error()
    assert lib.rbg_stash_list(8, 16, 10, C.byref(stash), 1, 1, None) == -1    # no scratch area
    assert lib.rbg_stash_refill_listed(None, 8, 16, 10, C.byref(stash), None) == -1
    assert b"scratch" in lib.rbg_last_error()
    # ABI 9: a list with in-flight claims needs its two ready ticks; one call per epoch boundary
    claims = nat.Stash(state, one, one, one, None, None, one, one, None, 2)
    assert lib.rbg_stash_list(8, 16, 10, C.byref(claims), 0, 5, None) == -3
    assert lib.rbg_stash_list(8, 16, 10, C.byref(claims), 9, 5, None) == -3
    assert b"claim_ready" in lib.rbg_last_error()
    assert lib.rbg_stash_roll(None) == -1 and b"NULL call" in lib.rbg_last_error()
    roll = nat.StashRollCall()
    roll.B, roll.m, roll.N = 8, 16, 10
    assert lib.rbg_stash_roll(C.byref(roll)) == -1 and b"NULL stash" in lib.rbg_last_error()
    roll.stash = C.pointer(stash)
    roll.launched = roll.listed = roll.done = one
    assert lib.rbg_stash_roll(C.byref(roll)) == -1 and b"in-flight" in lib.rbg_last_error()
    roll.stash = C.pointer(claims)
    roll.miss_word = one
    assert lib.rbg_stash_roll(C.byref(roll)) == -1 and b"miss snapshot" in lib.rbg_last_error()
    # ticks are int32: a launch whose tick + ready_delay would wrap is refused, never truncated
    call.state_in = call.state_out = state
    call.timestep = nat.Timestep(*[one] * 9, None, None)
    call.action = one
    call.stash = C.pointer(stash)
    for tick, delay in ((-1, 1), (5, -1), (2 ** 31 - 1, 1), (2 ** 30, 2 ** 30)):
        call.tick, call.ready_delay = tick, delay
        assert lib.rbg_connector_step_call(C.byref(call)) == -3, (tick, delay)
        assert b"rebase" in lib.rbg_last_error()
    # a rollout launch handles two slots per env
    three = nat.Stash(state, one, one, one, None, None, None, None, None, 3)
    params = nat.EnvParams(50, -0.03, 0.1, 1)
    rc = lib.rbg_connector_rollout_gen(None, 8, 16, 10, C.byref(state), C.byref(state), one, 0, 5,
                                       C.byref(params), None, None, None, None, C.byref(three), 0,
                                       1, None)
    assert rc == -3 and b"2 stash slots" in lib.rbg_last_error()
    assert lib.rbg_connector_rollout_gen(None, 8, 16, 10, C.byref(state), C.byref(state), one, 0, 0,
                                         C.byref(params), None, None, None, None, None, 0, 0,
                                         None) == -3  # n_steps < 1


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: no file of the package may reference it."""
    pkg = os.path.join(ROOT, "routing_board_generation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower(), (
                    f"{os.path.join(dirpath, f)} mentions the oracle")


def test_cpu_tensors_are_rejected_loudly():
    import torch

    from routing_board_generation_amd import ops

    keys = torch.zeros((4, 2), dtype=torch.uint32)
    with pytest.raises(ValueError, match="no CPU path"):
        ops.prw_generate(keys, 16, 10, split_first=True)


def test_no_kernel_spills_to_scratch(native, tmp_path):
    """Every kernel of the library keeps its state in registers: the gfx950 assembly (hipcc -S,
    device side only; no GPU needed) declares no private segment and no spilled VGPRs."""
    import re
    import subprocess

    asm = tmp_path / "kernels.s"
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = [f for f in native.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.check_call([hipcc] + flags + ["--cuda-device-only", "-S", "-o", str(asm),
                                             native.SRC_PATH], stderr=subprocess.DEVNULL)
    text = asm.read_text()
    kernels = re.findall(r"\.name:\s+(\S+)\n(?:.*\n){0,12}?\s+\.private_segment_fixed_size:\s+(\d+)",
                         text)
    assert len(kernels) >= 20, f"only {len(kernels)} kernel descriptors found"
    bad = [(n, s) for n, s in kernels if int(s) != 0]
    assert not bad, f"kernels with scratch: {bad}"
    spills = re.findall(r"\.vgpr_spill_count:\s+(\d+)", text)
    assert spills and all(int(x) == 0 for x in spills), spills
    assert "scratch_store" not in text and "scratch_load" not in text
