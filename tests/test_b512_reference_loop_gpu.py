"""The reference's OWN full loops at BASELINE config 2's batch (512 layouts, T = 100) against the engines (until r06 the
B = 512 loops were checked against the oracle and the exact engine, the reference itself only at B = 2 - 4).

tests/golden/rico25_b512_reference_loops.npz (oracle/make_b512_golden.py, 20 min of the reference on the CPU):
  * config 2 verbatim — `sample(batch_size=512, deterministic)` from all-[MASK];
  * the reference's greedy continuation (t = 49 .. 0) of the state its own `random` run stands in after 50 steps — greedy decisions at every
    one of 50 steps, not only the last two;
  * per layout the smallest top-2 margin the reference's sampler saw.
A layout may differ from the reference only if that margin is below MARGIN (a differently-ordered fp32 sum can flip such a decision, and the
layout's trajectory with it); `exact`, `split` and the two verified modes are all held to the same bar — the verified modes promise the
reference-precision engine's tokens."""
import os

import numpy as np
import pytest
import torch

from oracle import spec as SP
from oracle import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "rico25_b512_reference_loops.npz")
MARGIN = 1e-4
MAX_DIFFERING_LAYOUTS = 5        # of 512 (expected: 0 - 2, all with a margin ~1e-5 or below)


@pytest.mark.parametrize("precision", ["exact", "split", "fast_verified", "mixed_verified", "hybrid_verified"])
def test_reference_full_loops_at_batch_512(precision):
    from layout_dm_amd.diffusion import HipMaskAndReplaceDiffusion, timestep_schedule

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device (no CPU fallback exists)")
    g = np.load(GOLDEN)
    spec = SP.SPECS["rico25"]
    B = g["final_greedy"].shape[0]
    m = HipMaskAndReplaceDiffusion(n_category=spec.n_category, precision=precision, max_batch=B)
    m.load_state_dict(synth.synth_state_dict(spec, seed=int(g["weight_seed"]), perturb=True))

    def check(name, out, ref, margin):
        diff = (out != ref).any(dim=1)
        n = int(diff.sum())
        worst = float(margin[diff].max()) if n else 0.0
        print(f"[b512/{precision}/{name}] layouts differing from the reference's own loop: {n} / {B}"
              + (f" (largest reference margin among them {worst:.2e}, {int((out != ref).sum())} tokens)" if n else ""))
        assert n <= MAX_DIFFERING_LAYOUTS and worst < MARGIN, (name, n, worst)

    # ---- BASELINE config 2 verbatim: greedy from all-[MASK]
    out = m.sample(batch_size=B, sampling_cfg={"name": "deterministic", "num_timesteps": 100})
    check("config 2 verbatim", out, torch.from_numpy(g["final_greedy"].astype(np.int64)), torch.from_numpy(g["min_margin_greedy"]))
    # ---- greedy continuation of the reference's mid-trajectory states
    tm, tp = timestep_schedule(100, 100)
    assert tm[50] == 49
    mid = torch.from_numpy(g["mid_state"].astype(np.int32)).to(m.engine.device)
    if m.verified is not None:
        fin, _ = m.verified.sample_loop(mid.clone(), tm[50:], tp[50:])
        st = m.verified.last_stats
        print(f"[b512/{precision}] verification: marked {st['marked_layout_steps']}, re-checked {st['exact_layout_steps']}, corrected "
              f"{st['mismatch_layout_steps']}, audit mismatches {st['audit_mismatch_layout_steps']}")
        assert st["audit_mismatch_layout_steps"] == 0
    else:
        fin, _ = m.engine.sample_loop(mid.clone(), tm[50:], tp[50:], {"name": "deterministic"})
    check("from the reference's state after 50 steps", fin.cpu().long(), torch.from_numpy(g["final_from_mid"].astype(np.int64)),
          torch.from_numpy(g["min_margin_from_mid"]))
    m.close()
