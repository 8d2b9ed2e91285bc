"""A bounded slice of the randomised parity soak (tests/soak_parity.py) inside `-m gpu`: random
grid sizes, agent counts, batch sizes, options, stash schedules and seeds -- every generator of
setup_train.py:112-152 (+ the opt-in Philox walk), the offline sampler, the coupled env path
through the per-step API and through fused / recording rollouts -- HIP against the CPU oracle for
about 20 seconds, from a fixed seed.  The long runs (profiles/*_soak_parity.json) are the
builder's; this slice is what a fresh box's test run carries.  A failure prints the case line,
from which the case is reproducible (`python tests/soak_parity.py --seed ...`)."""
import json
import types

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,kinds", [(20260401, "prw,uniform,seq_rw,seed_ext,philox,dataset"),
                                        (20260402, "env,rollout")])
def test_randomised_soak_slice(seed, kinds):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (there is no CPU fallback to hide behind)")
    import soak_parity

    args = types.SimpleNamespace(minutes=10.0 / 60.0, seed=seed, out="", kinds=kinds,
                                 max_boards=3000)
    summary, failures = soak_parity.run(args, quiet=True)
    print(json.dumps(summary))
    assert not failures, "\n".join(json.dumps(f) for f in failures[:5])
    # the slice did run: every kind it names several times
    for kind in kinds.split(","):
        assert summary["totals"].get(kind, {}).get("cases", 0) >= 2, summary
