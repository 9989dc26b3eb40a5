"""Alignment / overlap metrics of generated layouts (trainer/helpers/metric.py:98-203; eval.py:153-155,203-205).

CPU: the oracle restatement (oracle/restatement.py layout_metrics) and the host build of the kernel's one source
(csrc/ldm_layout_metrics_core.h) against tests/golden/layout_metrics.npz — produced by the reference's own compute_alignment /
compute_overlap on random decoded layouts and on the edge cases its code branches on (empty layout, single element,
identical boxes, zero-area boxes, touching boxes, stale boxes in padded slots).
GPU: ldm_layout_metrics through the C-ABI against the same fixture and, at the bench's batch size, against the oracle; the
decode kernel's device output feeds it directly.  fp32 sums in another order than torch's: rtol 1e-5 (observed 2e-7)."""
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import restatement as R
from oracle.make_golden import metrics_layouts

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("alignment-ACLayoutGAN", "alignment-LayoutGAN++", "alignment-NDN", "overlap-ACLayoutGAN", "overlap-LayoutGAN++",
        "overlap-LayoutGAN")
RTOL, ATOL = 1e-5, 1e-7


def _close(a, b, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    bad = np.abs(a - b) > ATOL + RTOL * np.abs(b)
    assert not bad.any(), (what, a[bad][:4], b[bad][:4])


def test_oracle_restatement_vs_reference_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "layout_metrics.npz"))
    bbox, mask = metrics_layouts()
    assert np.array_equal(bbox, g["bbox"]) and np.array_equal(mask, g["mask"])      # the fixture's inputs are reproducible
    o = R.layout_metrics(g["bbox"], g["mask"])
    assert tuple(o) == KEYS
    for k in KEYS:
        _close(o[k], g[k], k)
    # known answers: an empty layout scores 0 everywhere; two touching boxes do not overlap; identical x-coordinates align
    assert all(g[k][0] == 0 for k in KEYS)
    assert g["overlap-LayoutGAN"][4] == 0 and g["alignment-NDN"][4] == 1.0   # |xl - xl'| = .5, |xr - xl'| = 0 -> per element .5


@pytest.fixture(scope="module")
def host_exe(tmp_path_factory):
    exe = tmp_path_factory.mktemp("metrics") / "cpu_metrics_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", os.path.join(ROOT, "tests", "cpu_metrics_check.cpp"), "-o",
                    str(exe)], check=True, cwd=ROOT)
    return str(exe)


def _run_host(exe, tmp_path, bbox, mask):
    B, S = mask.shape
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(np.array([B, S], np.int32).tobytes())
        f.write(np.ascontiguousarray(bbox, np.float32).tobytes())
        f.write(np.ascontiguousarray(mask, np.uint8).tobytes())
    subprocess.run([exe, str(inp), str(outp)], check=True)
    return np.fromfile(outp, np.float32).reshape(B, 6)


def test_kernel_source_on_the_host_vs_reference_fixture_and_oracle(host_exe, tmp_path, golden_dir):
    g = np.load(os.path.join(golden_dir, "layout_metrics.npz"))
    out = _run_host(host_exe, tmp_path, g["bbox"], g["mask"])
    for i, k in enumerate(KEYS):
        _close(out[:, i], g[k], k)
    # ragged sizes: S = 1, S = 7, S = 64, un-quantised boxes, random holes in the mask
    rng = np.random.default_rng(3)
    for S in (1, 7, 64):
        bbox = rng.random((9, S, 4)).astype(np.float32)
        mask = rng.random((9, S)) < 0.6
        out = _run_host(host_exe, tmp_path, bbox, mask)
        o = R.layout_metrics(bbox, mask)
        for i, k in enumerate(KEYS):
            _close(out[:, i], o[k], (S, k))


@pytest.fixture(scope="module")
def cuda():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device (no CPU fallback exists)")
    return torch.device("cuda", 0)


@pytest.mark.gpu
def test_device_metrics_vs_reference_fixture(cuda, golden_dir):
    from layout_dm_amd import metrics as M

    g = np.load(os.path.join(golden_dir, "layout_metrics.npz"))
    bbox, mask = torch.from_numpy(g["bbox"]), torch.from_numpy(g["mask"])
    al, ov = M.compute_alignment(bbox, mask), M.compute_overlap(bbox, mask)
    assert tuple(al) + tuple(ov) == KEYS and all(not v.is_cuda and v.dtype == torch.float32 for v in al.values())
    for k, v in {**al, **ov}.items():
        _close(v.numpy(), g[k], k)
    # device tensors in, device tensors out
    al_d = M.compute_alignment(bbox.to(cuda), mask.to(cuda))
    assert all(v.is_cuda for v in al_d.values()) and torch.equal(al_d["alignment-NDN"].cpu(), al["alignment-NDN"])
    with pytest.raises(RuntimeError):
        M.layout_metrics(torch.zeros(2, 300, 4), torch.ones(2, 300, dtype=torch.bool))     # S > 256
    with pytest.raises(ValueError):
        M.layout_metrics(torch.zeros(2, 5, 3), torch.ones(2, 5, dtype=torch.bool))


@pytest.mark.gpu
def test_device_metrics_on_decoded_samples_vs_oracle(cuda):
    """The consumer chain of a sampling call, all on the device: tokens -> decode_layouts_k -> layout_metrics_k, 512 layouts,
    against the oracle on the same boxes (and S = 1 / 64 / 200 on random boxes)."""
    from layout_dm_amd import metrics as M
    from layout_dm_amd.binding import Engine
    from oracle import spec as SP

    spec = SP.RICO25
    e = Engine(n_category=spec.n_category, precision="fast", max_batch=8)
    g = torch.Generator().manual_seed(0)
    B = 512
    tokens = torch.empty(B, spec.seq_len, dtype=torch.long)
    for a in range(spec.n_attr):
        ids = torch.as_tensor(spec.full_ids(a))
        tokens[:, a::spec.n_attr] = ids[torch.randint(0, len(ids) - 1, (B, spec.max_elem), generator=g)]
    n = torch.randint(0, spec.max_elem + 1, (B,), generator=g)
    tokens.view(B, spec.max_elem, spec.n_attr)[torch.arange(spec.max_elem)[None] >= n[:, None]] = spec.pad_id
    dec = e.decode(tokens.int().to(cuda))
    out = M.layout_metrics(dec["bbox"], dec["mask"])
    o = R.layout_metrics(dec["bbox"].cpu().numpy(), dec["mask"].cpu().numpy())
    for i, k in enumerate(KEYS):
        _close(out[:, i].cpu().numpy(), o[k], k)
    rng = np.random.default_rng(1)
    for S in (1, 64, 200):
        bbox = rng.random((5, S, 4)).astype(np.float32)
        mask = rng.random((5, S)) < 0.7
        out = M.layout_metrics(torch.from_numpy(bbox), torch.from_numpy(mask)).cpu().numpy()
        o = R.layout_metrics(bbox, mask)
        for i, k in enumerate(KEYS):
            _close(out[:, i], o[k], (S, k))
