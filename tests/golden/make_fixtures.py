#!/usr/bin/env python3
"""Extract golden vectors (DATA only) from the reference's own test file and executed notebook.

Run once in the build container (needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_fixtures.py

Writes tests/golden/reference_goldens.json.  Nothing from the reference is imported or executed:
the test file is parsed with `ast` for its module-level array literals, and the notebook is read as
JSON for the *printed outputs* of its executed cells.

Sources
  * routing_board_generation/board_generation_methods/jax_implementation/board_generation/
    test_parallel_random_walk_board.py:37-148  (grids and Agent pytrees used by the reference tests)
  * package_evaluation/profiling_generators.ipynb  (printed `State(...)` of each generator for
    PRNGKey(0) at 10x10/5 agents, and of the vmapped generators over split(PRNGKey(0), 100))
"""
import ast
import json
import os
import re

REF = "/root/reference"
TEST_FILE = os.path.join(
    REF,
    "routing_board_generation/board_generation_methods/jax_implementation/board_generation/"
    "test_parallel_random_walk_board.py",
)
NOTEBOOK = os.path.join(REF, "package_evaluation/profiling_generators.ipynb")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_goldens.json")


# ----------------------------------------------------------------------------- test-file literals
def _literal(node):
    """Evaluate jnp.array([...]) / jnp.arange(n) / plain literals appearing in the test module."""
    if isinstance(node, ast.Call):
        fn = node.func.attr if isinstance(node.func, ast.Attribute) else getattr(node.func, "id", "")
        if fn == "array":
            return _literal(node.args[0])
        if fn == "arange":
            return list(range(_literal(node.args[0])))
        if fn == "zeros":
            shape = _literal(node.args[0])
            return [[0] * shape[1] for _ in range(shape[0])]
        if fn == "Agent":
            return {kw.arg: _literal(kw.value) for kw in node.keywords}
        raise ValueError(f"unhandled call {fn}")
    value = ast.literal_eval(node)
    return _to_int(value)


def _to_int(v):
    if isinstance(v, (list, tuple)):
        return [_to_int(x) for x in v]
    if isinstance(v, float):
        assert v == int(v)
        return int(v)
    return v


def extract_test_literals():
    tree = ast.parse(open(TEST_FILE).read())
    wanted = {
        "valid_starting_grid",
        "valid_starting_grid_after_1_step",
        "valid_end_grid",
        "valid_end_grid2",
        "grid_to_test_available_cells",
        "grids_after_1_agent_step",
        "agents_finished",
        "agents_reshaped_for_generator",
        "agents_starting",
        "agents_starting_move_1_step_up",
    }
    out = {}
    for node in tree.body:
        if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name):
            name = node.targets[0].id
            if name in wanted:
                out[name] = _literal(node.value)
    missing = wanted - set(out)
    assert not missing, missing
    return out


# ----------------------------------------------------------------------------- notebook outputs
def _cell_output(nb, idx):
    for o in nb["cells"][idx].get("outputs", []):
        if "data" in o and "text/plain" in o["data"]:
            return "".join(o["data"]["text/plain"])
    raise ValueError(f"cell {idx} has no text/plain output")


def _array_after(text, field):
    """Return the python literal printed after `field=Array(` (bracket-matched)."""
    m = re.search(re.escape(field) + r"=Array\(", text)
    assert m, field
    i = m.end()
    if text[i] != "[":  # scalar
        j = text.index(",", i)
        return int(text[i:j])
    depth = 0
    for j in range(i, len(text)):
        if text[j] == "[":
            depth += 1
        elif text[j] == "]":
            depth -= 1
            if depth == 0:
                break
    return text[i : j + 1]


def _eval(lit):
    return ast.literal_eval(lit)


def _strip_ellipsis(v):
    """numpy's summarised print keeps 3 leading + 3 trailing items per truncated axis."""
    if isinstance(v, list):
        return [_strip_ellipsis(x) for x in v if x is not Ellipsis]
    return v


def parse_state(text, batched):
    st = {}
    grid = _eval(_array_after(text, "grid"))
    if batched:
        st["grid_window"] = _strip_ellipsis(grid)  # boards/rows/cols [0,1,2,-3,-2,-1]
    else:
        st["grid"] = grid
    for f in ("start", "target", "position"):
        st[f] = _eval(_array_after(text, f))
    st["key"] = _eval(_array_after(text, "key"))
    return st


def extract_notebook():
    nb = json.load(open(NOTEBOOK))
    single = {
        "uniform": 4,
        "sequential_random_walk": 7,
        "seed_extension": 10,
        "parallel_random_walk": 13,
    }
    batch = {"uniform": 18, "seed_extension": 23, "parallel_random_walk": 26}
    out = {"single_10x10_5_PRNGKey0": {}, "batch100_10x10_5_split_PRNGKey0": {}}
    for name, idx in single.items():
        out["single_10x10_5_PRNGKey0"][name] = parse_state(_cell_output(nb, idx), batched=False)
    for name, idx in batch.items():
        out["batch100_10x10_5_split_PRNGKey0"][name] = parse_state(
            _cell_output(nb, idx), batched=True
        )
    return out


def main():
    data = {
        "_provenance": {
            "test_file": "routing_board_generation/board_generation_methods/jax_implementation/"
            "board_generation/test_parallel_random_walk_board.py:37-148",
            "notebook": "package_evaluation/profiling_generators.ipynb (executed cell outputs)",
            "note": "data only; produced by tests/golden/make_fixtures.py",
        },
        "test_file": extract_test_literals(),
        "notebook": extract_notebook(),
    }
    with open(OUT, "w") as f:
        json.dump(data, f, separators=(",", ":"))
    b = data["notebook"]["batch100_10x10_5_split_PRNGKey0"]["parallel_random_walk"]
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
    print("batch PRW start shape", len(b["start"]), len(b["start"][0]), "key", len(b["key"]))
    print("grid window", len(b["grid_window"]), len(b["grid_window"][0]), len(b["grid_window"][0][0]))


if __name__ == "__main__":
    main()
