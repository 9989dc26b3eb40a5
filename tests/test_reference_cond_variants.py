"""GPU: the fused HIP step against reference-produced known answers for the condition types and options that round 1
only checked against the oracle — cond=cwh, cond=partial (helpers/task.py:61-110) and time_difference (base.py:218-226).
Fixture: tests/golden/rico25_cond_variants.npz (oracle/make_golden.py::cond_variant_cases; the same file pins the oracle
in tests/test_oracle_golden.py).  Criterion as in test_hip_parity.py: greedy tokens bit-exact in the exact mode,
reference-margin-bounded in the fast mode."""
import os

import numpy as np
import pytest
import torch

from test_hip_parity import MARGIN_BOUND, _assert_traj, engine  # noqa: F401  (shared helpers)
from test_hip_parity import cuda  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _sub(g, prefix):
    return {k[len(prefix):]: g[k] for k in g.files if k.startswith(prefix)}


@pytest.mark.parametrize("precision", ["exact", "fast"])
@pytest.mark.parametrize("variant", ["cwh", "partial", "td"])
def test_step_teacher_forced_cond_variants(cuda, golden_dir, variant, precision):  # noqa: F811
    e = engine("rico25", precision)
    g = np.load(os.path.join(golden_dir, "rico25_cond_variants.npz"))
    t = _sub(g, variant + "_")
    cond = None
    if variant != "td":
        cond = {"seq": t["cond_seq"].astype(np.int64), "mask": t["cond_mask"], "type": variant}
    T = e.T
    before = torch.from_numpy(t["states_before"].astype(np.int32))
    ref_next = torch.from_numpy(t["greedy_next"].astype(np.int32))
    margin = torch.from_numpy(t["greedy_margin"])
    bad, worst = 0, 0.0
    for i, tm in enumerate(t["steps"]):
        tm = int(tm)
        tp = min(max(tm - int(T * 0.15), 0), T - 1) if variant == "td" else tm  # base.py:218-226
        out = e.sample_step(before[i], tm, {"name": "deterministic"}, t_post=tp, cond=cond, step=i).cpu()
        mism = out != ref_next[i]
        if mism.any():
            bad += int(mism.sum())
            worst = max(worst, margin[i][mism].max().item())
    _assert_traj(variant, precision, bad, ref_next.numel(), worst)
