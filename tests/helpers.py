"""Helpers shared by the GPU parity tests."""
import numpy as np


def to_np(t):
    return t.detach().cpu().numpy()


def state_to_np(state):
    """package State (batched) -> dict of numpy arrays in the oracle's layout."""
    return {"grid": to_np(state.grid), "step_count": to_np(state.step_count),
            "start": to_np(state.agents.start), "target": to_np(state.agents.target),
            "position": to_np(state.agents.position), "key": to_np(state.key)}


def timestep_to_np(ts):
    return {"step_type": to_np(ts.step_type), "reward": to_np(ts.reward),
            "discount": to_np(ts.discount), "obs_grid": to_np(ts.observation.grid),
            "action_mask": to_np(ts.observation.action_mask).astype(np.uint8),
            "obs_step_count": to_np(ts.observation.step_count),
            "num_connections": to_np(ts.extras["num_connections"]),
            "ratio_connections": to_np(ts.extras["ratio_connections"]),
            "total_path_length": to_np(ts.extras["total_path_length"])}


def assert_dicts_equal(got, exp, what=""):
    for k in exp:
        g, e = np.asarray(got[k]), np.asarray(exp[k])
        assert g.shape == e.shape, f"{what}{k}: shape {g.shape} != {e.shape}"
        if e.dtype.kind == "f":
            # rewards/discounts/ratios are single IEEE float32 operations: bit-exact
            same = g.view(np.uint32) == e.view(np.uint32)
        else:
            same = g == e
        if not same.all():
            bad = np.argwhere(~same)[:5].tolist()
            raise AssertionError(f"{what}{k}: {(~same).sum()} mismatches, first at {bad}")


def same_wire_neighbour_count(solved):
    """[B,m,m] -> number of 4-neighbours belonging to the same wire, per cell."""
    wire = np.where(solved > 0, (solved - 1) // 3, -1)
    cnt = np.zeros_like(solved)
    pad = np.pad(wire, ((0, 0), (1, 1), (1, 1)), constant_values=-2)
    m = solved.shape[1]
    for dr, dc in ((0, 1), (2, 1), (1, 0), (1, 2)):
        nb = pad[:, dr:dr + m, dc:dc + m]
        cnt += ((nb == wire) & (wire >= 0)).astype(solved.dtype)
    return cnt


def check_solved_boards(solved, start, target):
    """Vectorised verify_wire_validity (NPI/utils/post_processor_utils_numpy.py:90-121):
    path cells touch exactly 2 cells of their wire, heads/targets exactly 1; a wire whose agent
    never moved is a lone TARGET cell."""
    B, m, _ = solved.shape
    N = start.shape[1]
    cnt = same_wire_neighbour_count(solved)
    kind = np.where(solved > 0, (solved - 1) % 3, -1)
    wire = np.where(solved > 0, (solved - 1) // 3, 0)
    never_moved = (start == target).all(-1)                       # [B, N]
    nm_cell = np.take_along_axis(never_moved, wire.reshape(B, -1), axis=1).reshape(B, m, m)
    assert (solved >= 0).all() and (solved <= 3 * N).all()
    assert (cnt[kind == 0] == 2).all(), "a path cell without exactly two same-wire neighbours"
    ends = (kind > 0) & ~nm_cell
    assert (cnt[ends] == 1).all(), "a head/target without exactly one same-wire neighbour"
    lone = (kind > 0) & nm_cell
    assert (cnt[lone] == 0).all()
    b = np.arange(B)[:, None]
    i = np.arange(N)[None, :]
    assert (solved[b, target[..., 0], target[..., 1]] == 3 + 3 * i).all()
    head_ok = (solved[b, start[..., 0], start[..., 1]] == 2 + 3 * i) | never_moved
    assert head_ok.all()
    # exactly one head and one target per wire (or a single target when it never moved)
    for v, expect in ((2, ~never_moved), (0, np.ones_like(never_moved))):
        k = (v - 1) % 3
        for w in range(N):
            n_cells = ((kind == k) & (wire == w) & (solved > 0)).sum((1, 2))
            assert (n_cells == expect[:, w].astype(int)).all()
