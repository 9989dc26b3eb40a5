"""GPU: BASELINE config 5's acceptance metric — "FID vs reference" — as ONE test.

The reference's own `sample()` (CPU, sampling=random, torch.multinomial) drew two independent sets of 1 024 layouts from the
trained-like "mid" synthetic checkpoint (oracle/make_reference_samples.py -> tests/golden/rico25_mid_reference_samples.npz;
tests/test_oracle_vs_reference.py re-generates the first chunk of each seed bit for bit).  Here the SAME checkpoint is
sampled through the HIP path (exact and fast numerics, Philox draws), every set goes ids -> {bbox, label, mask}
(ldm_decode_layouts) -> FIDNetV3 features (ldm_fid_features, synthetic extractor weights) -> FID + precision / recall /
density / coverage (helpers/metric.py:37-59, eval.py:203-220), and

    FID(ours, ref seed A or B)  <=  1.5 x FID(ref seed A, ref seed B)        (the seed-to-seed spread at this sample size)
    |PRDC(ours | ref A) - PRDC(ref B | ref A)|  <=  0.05 + 3 x |PRDC(ref B | ref A) - PRDC(ref A | ref B)|

i.e. our samples are as close to the reference's as the reference's are to each other."""
import os

import numpy as np
import pytest
import torch

from oracle import fid as OF
from oracle import spec as SP
from oracle import synth

pytestmark = pytest.mark.gpu
FIXTURE = "rico25_mid_reference_samples.npz"


@pytest.mark.parametrize("precision", ["exact", "fast"])
def test_fid_vs_reference_samples(golden_dir, precision):
    from layout_dm_amd.diffusion import HipMaskAndReplaceDiffusion
    from layout_dm_amd.fid import FIDNetV3, scores_vs_reference_samples

    path = os.path.join(golden_dir, FIXTURE)
    if not os.path.exists(path):
        pytest.fail(f"{FIXTURE} missing: run python -m oracle.make_reference_samples in the build container")
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device (no CPU fallback exists)")
    g = np.load(path)
    ref = torch.from_numpy(g["tokens"].astype(np.int32))            # (2, n, S)
    n = ref.shape[1]
    spec = SP.RICO25
    m = HipMaskAndReplaceDiffusion(n_category=spec.n_category, precision=precision, max_batch=n)
    m.load_state_dict(synth.trained_like_state_dict(spec, str(g["point"]), seed=int(g["weight_seed"])))
    ours = m.sample(batch_size=n, sampling_cfg={"name": "random", "temperature": 1.0}, seed=12345, return_device_tensor=True)
    fid_model = FIDNetV3(num_label=spec.n_category, max_bbox=spec.max_elem)
    fid_model.load_state_dict({k: torch.from_numpy(v) for k, v in OF.synth_fid_state_dict(spec.n_category, seed=0,
                                                                                          max_bbox=spec.max_elem).items()})
    sc = scores_vs_reference_samples(m.engine, fid_model, ref[0], ref[1], ours)
    spread = sc["ref_b_vs_ref_a"]["fid"]
    print(f"[FID vs reference / {precision}] n = {n} per set; FID(ref B, ref A) = {spread:.4f}; FID(ours, ref A) = "
          f"{sc['ours_vs_ref_a']['fid']:.4f}; FID(ours, ref B) = {sc['ours_vs_ref_b']['fid']:.4f}")
    for k in ("precision", "recall", "density", "coverage"):
        print(f"    {k:9s} ref B | ref A {sc['ref_b_vs_ref_a'][k]:.4f}   ref A | ref B {sc['ref_a_vs_ref_b'][k]:.4f}   "
              f"ours | ref A {sc['ours_vs_ref_a'][k]:.4f}   ours | ref B {sc['ours_vs_ref_b'][k]:.4f}")
    assert spread > 0
    assert sc["ours_vs_ref_a"]["fid"] <= 1.5 * spread and sc["ours_vs_ref_b"]["fid"] <= 1.5 * spread
    for k in ("precision", "recall", "density", "coverage"):
        tol = 0.05 + 3 * abs(sc["ref_b_vs_ref_a"][k] - sc["ref_a_vs_ref_b"][k])
        assert abs(sc["ours_vs_ref_a"][k] - sc["ref_b_vs_ref_a"][k]) <= tol, (k, sc)
        assert abs(sc["ours_vs_ref_b"][k] - sc["ref_a_vs_ref_b"][k]) <= tol, (k, sc)
    # sanity of the harness itself: a DIFFERENT distribution (the init-like checkpoint) is further away than the spread
    m2 = HipMaskAndReplaceDiffusion(n_category=spec.n_category, precision="fast", max_batch=n)
    m2.load_state_dict(synth.trained_like_state_dict(spec, "wide", seed=5))
    other = m2.sample(batch_size=n, sampling_cfg={"name": "random", "temperature": 1.0}, seed=1, return_device_tensor=True)
    far = scores_vs_reference_samples(m.engine, fid_model, ref[0], ref[1], other)["ours_vs_ref_a"]["fid"]
    print(f"    (a different checkpoint: FID = {far:.4f})")
    assert far > 1.5 * spread
    m.engine.close()
    m2.engine.close()
