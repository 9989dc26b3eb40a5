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
