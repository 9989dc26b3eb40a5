"""TEST INFRASTRUCTURE ONLY — the reference's OWN full loops at BASELINE config 2's batch ("B = 512 / 1 024 are checked
against the oracle restatement for single steps and against the exact engine for loops, never against the reference's own full loop at that size").

    python -m oracle.make_b512_golden            (build container, /root/reference present; ~20 min on 8 cores)

Runs the real reference (oracle/ref_harness.py) on the synthetic Rico25 checkpoint of the other goldens (oracle/make_golden.py load_synth) and leaves
tests/golden/rico25_b512_reference_loops.npz:
  * `final_greedy`  (512, 125): `sample(batch_size=512, sampling_cfg=deterministic)` from all-[MASK], T = 100 — BASELINE config 2 verbatim
    (base.py:293-371);
  * `mid_state`     (512, 125): the state of a `random` run of the same call (torch.manual_seed(SEED)) after 50 of its 100 steps, and
    `final_from_mid` (512, 125): the reference's own greedy continuation of it — `_sample_single_step` (base.py:205-291) for t = 49 .. 0, exactly
    what sample()'s loop body does — so that the greedy decisions are NOT confined to the last two steps (greedy decoding of this diffusion keeps
    every token [MASK] until t <= 2: SURVEY App. G);
  * `min_margin_*`  (512,): per layout, the smallest top-2 gap of the log-probabilities the reference handed to its sampler (helpers/sampling.py)
    at any step and token: a reduced-precision — or merely differently-ordered fp32 — implementation may legitimately differ on a layout whose
    margin is below its own error.
tests/test_b512_reference_loop_gpu.py holds the engines to it; tests/test_oracle_vs_reference.py re-runs a slice with the live reference."""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

from . import make_golden as MG
from . import ref_harness as rh
from . import spec as SP

OUT = os.path.join(MG.OUT, "rico25_b512_reference_loops.npz")
SEED = 20260601
B = 512


def _spy_margins(fn):
    """Run fn() with the reference's sampler spied on; returns (result, per-layout min top-2 gap over every call)."""
    import trainer.models.categorical_diffusion.base as ref_base

    orig = ref_base.sample
    worst = {}

    def spy(logits, sampling_cfg):
        top2 = logits.detach().topk(2, dim=1).values
        gap = (top2[:, 0] - top2[:, 1]).amin(dim=-1)          # (B,)
        worst["m"] = gap if "m" not in worst else torch.minimum(worst["m"], gap)
        return orig(logits, sampling_cfg)

    ref_base.sample = spy
    try:
        out = fn()
    finally:
        ref_base.sample = orig
    return out, worst["m"]


def greedy_from(m, spec, state, t_first):
    """The reference's greedy continuation of `state` (B, S) through t = t_first .. 0 (the body of sample()'s loop, skip_step = 0)."""
    from trainer.models.categorical_diffusion.util import index_to_log_onehot, log_onehot_to_index

    det = rh.sampling_cfg("deterministic")
    log_z = index_to_log_onehot(state, spec.n_class)
    for t in range(t_first, -1, -1):
        tt = torch.full((state.shape[0],), t, dtype=torch.long)
        log_z = m._sample_single_step(log_z, tt, skip_step=0, sampling_cfg=det, cond=None)
    return log_onehot_to_index(log_z)


def main(batch=B, out=OUT):
    torch.set_num_threads(os.cpu_count() or 8)
    spec = SP.SPECS["rico25"]
    m, _tok = rh.build_reference_model("rico25")
    MG.load_synth(m, spec)
    t0 = time.time()
    with torch.no_grad():
        final_greedy, mg = _spy_margins(lambda: m.sample(batch_size=batch, cond=None, sampling_cfg=rh.sampling_cfg("deterministic")))
        print(f"greedy loop: {time.time() - t0:.0f} s", flush=True)
        torch.manual_seed(SEED)
        inter = m.sample(batch_size=batch, cond=None, sampling_cfg=rh.sampling_cfg("random"), get_intermediate_results=True)
        mid = inter[49].clone()                                 # state after the 50th step (t = 50): the next step is t = 49
        print(f"random loop: {time.time() - t0:.0f} s", flush=True)
        final_mid, mm = _spy_margins(lambda: greedy_from(m, spec, mid, 49))
        print(f"greedy continuation: {time.time() - t0:.0f} s", flush=True)
    np.savez_compressed(out, final_greedy=final_greedy.numpy().astype(np.int16), min_margin_greedy=mg.numpy().astype(np.float32),
                        mid_state=mid.numpy().astype(np.int16), final_from_mid=final_mid.numpy().astype(np.int16),
                        min_margin_from_mid=mm.numpy().astype(np.float32), seed=np.int64(SEED), weight_seed=np.int64(MG.WEIGHT_SEED))
    print(f"wrote {out} ({os.path.getsize(out) / 1e3:.0f} KB); masked tokens left: {int((final_greedy == spec.mask_id).sum())} / "
          f"{int((final_mid == spec.mask_id).sum())}; layouts with a margin below 1e-4: {int((mg < 1e-4).sum())} / {int((mm < 1e-4).sum())}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else B, sys.argv[2] if len(sys.argv) > 2 else OUT)
