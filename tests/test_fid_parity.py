"""GPU parity of the FID feature extractor (ldm_fid_features, kernels_fid.hip) — SURVEY §8f row 3:
features == the REAL reference's FIDNetV3.extract_features (golden) and == the oracle restatement on a full batch;
tolerance: fp32 accumulation-order noise (1e-4 absolute on features of magnitude ~3), stated here."""
import os

import numpy as np
import pytest
import torch

from oracle import fid as OF

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _model(num_label, max_bbox=25):
    from layout_dm_amd.fid import FIDNetV3

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device (no CPU fallback exists)")
    m = FIDNetV3(num_label=num_label, max_bbox=max_bbox)
    sd = OF.synth_fid_state_dict(num_label, seed=0, max_bbox=max_bbox)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.eval(), sd


def test_fid_features_vs_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "fid_v3.npz"))
    m, _ = _model(int(g["num_label"]))
    f = m.extract_features(torch.from_numpy(g["bbox"]), torch.from_numpy(g["label"]), torch.from_numpy(g["padding_mask"]))
    assert f.is_cuda and f.shape == (6, 256)
    err = np.abs(f.cpu().numpy() - g["features"]).max()
    print(f"[fid] max |feature - reference| = {err:.3e}")
    assert err <= TOL


@pytest.mark.parametrize("num_label,N", [(25, 25), (5, 25), (25, 9)])
def test_fid_features_vs_oracle_full_batch(num_label, N):
    """B=512 layouts (the sampling batch), ragged element counts incl. empty layouts, fewer slots than max_bbox;
    then the FID of two feature sets through the same host formula."""
    from layout_dm_amd.fid import compute_fid

    m, sd = _model(num_label)
    bbox, label, pm = OF.synth_layouts(num_label, 512, N, seed=3)
    f = m.extract_features(torch.from_numpy(bbox), torch.from_numpy(label), torch.from_numpy(pm)).cpu().numpy()
    ref = OF.extract_features(sd, bbox, label, pm).numpy()
    assert np.abs(f - ref).max() <= TOL
    # batch-composition independence, bit for bit
    one = m.extract_features(torch.from_numpy(bbox[7:8]), torch.from_numpy(label[7:8]), torch.from_numpy(pm[7:8]))
    assert np.array_equal(one.cpu().numpy()[0], f[7])
    # FID of the device features == FID of the oracle's, and a perturbed set is further away than an identical one
    bbox2 = np.clip(bbox + 0.05, 0, 1).astype(np.float32)
    f2 = m.extract_features(torch.from_numpy(bbox2), torch.from_numpy(label), torch.from_numpy(pm)).cpu().numpy()
    ref2 = OF.extract_features(sd, bbox2, label, pm).numpy()
    fid_dev, fid_ref = compute_fid(f, f2), compute_fid(ref, ref2)
    assert abs(fid_dev - fid_ref) <= 1e-3 * max(1.0, abs(fid_ref))
    assert compute_fid(f, f) < 1e-6 < fid_dev
    assert m.extract_features(torch.zeros(0, N, 4), torch.zeros(0, N, dtype=torch.long),
                              torch.zeros(0, N, dtype=torch.bool)).shape == (0, 256)


def test_fid_rejects_bad_geometry():
    from layout_dm_amd.fid import FIDNetV3

    with pytest.raises(RuntimeError):
        FIDNetV3(num_label=25, max_bbox=40)          # one workgroup holds token + <= 31 elements
    m, _ = _model(25)
    with pytest.raises(RuntimeError):
        m.extract_features(torch.zeros(1, 26, 4), torch.zeros(1, 26, dtype=torch.long), torch.zeros(1, 26, dtype=torch.bool))


@pytest.mark.parametrize("n_real,n_fake,dim,k", [(700, 500, 256, 5), (64, 333, 48, 3), (257, 257, 256, 7)])
def test_prdc_vs_oracle(n_real, n_fake, dim, k):
    """ldm_prdc (kernels_prdc.hip) == the restatement of prdc.compute_prdc on random feature clouds that overlap only
    partly (so that none of the four numbers is trivially 0 or 1); identical sets give precision = recall = coverage = 1."""
    from layout_dm_amd.fid import compute_generative_model_scores, compute_prdc

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device (no CPU fallback exists)")
    rng = np.random.default_rng(n_real + dim)
    real = rng.standard_normal((n_real, dim)).astype(np.float32)
    fake = (rng.standard_normal((n_fake, dim)) * 1.15 + 0.08).astype(np.float32)
    got, ref = compute_prdc(real, fake, nearest_k=k), OF.compute_prdc(real, fake, nearest_k=k)
    print(f"[prdc n={n_real}x{n_fake} dim={dim} k={k}] device {got}  oracle {ref}")
    for key in ("precision", "recall", "density", "coverage"):
        assert 0.0 < ref[key] or key in ("precision", "recall")
        assert abs(got[key] - ref[key]) <= 2.0 / min(n_real, n_fake) + 1e-6, (key, got[key], ref[key])  # <= 2 samples on a '<' edge
    same = compute_prdc(real, real, nearest_k=k)
    assert same["precision"] == 1.0 and same["recall"] == 1.0 and same["coverage"] == 1.0
    if k == 5:
        full = compute_generative_model_scores([torch.from_numpy(real[:300]), torch.from_numpy(real[300:])], torch.from_numpy(fake))
        assert set(full) == {"precision", "recall", "density", "coverage", "fid"} and full["fid"] > 0
