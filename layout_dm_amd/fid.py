"""FID feature extractor + Fréchet distance on the MI355X.

Mirrors what the reference's evaluation uses of `trainer/fid/model.py`:
  * `FIDNetV3(num_label, d_model=256, nhead=4, num_layers=4, max_bbox=50)` (model.py:123-151) with
    `.load_state_dict`, `.eval()`, `.to()`, and `.extract_features(bbox, label, padding_mask) -> (B, 256)`
    (model.py:147-152) — the only method eval.py calls on it;
  * `load_fidnet_v3(dataset, weight_dir, device)` (model.py:182-193): same checkpoint path / format;
  * `compute_fid(feats_real, feats_fake)`: the "fid" entry of `compute_generative_model_scores`
    (helpers/metric.py:37-59) = pytorch_fid 0.2.1 `calculate_frechet_distance` on the feature means / covariances.
The forward runs in libldm_hip.so (ldm_fid_features, kernels_fid.hip); there is no CPU / eager fallback.  The decoder
half of FIDNetV3 (reconstruction heads used only to TRAIN the extractor, model.py:154-180) is out of scope.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np
import torch

from .binding import _stream_ptr, load_library


def _declare(lib):
    vp, i32 = C.c_void_p, C.c_int
    lib.ldm_fid_create.argtypes = [i32, i32, i32, i32, i32, i32, C.POINTER(vp)]
    lib.ldm_fid_create.restype = i32
    lib.ldm_fid_destroy.argtypes = [vp]
    lib.ldm_fid_destroy.restype = None
    lib.ldm_fid_last_error.argtypes = [vp]
    lib.ldm_fid_last_error.restype = C.c_char_p
    lib.ldm_fid_load_weight.argtypes = [vp, C.c_char_p, vp, C.POINTER(C.c_int64), i32]
    lib.ldm_fid_load_weight.restype = i32
    lib.ldm_fid_finalize.argtypes = [vp]
    lib.ldm_fid_finalize.restype = i32
    lib.ldm_fid_features.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp]
    lib.ldm_fid_features.restype = i32
    return lib


class FIDNetV3:
    def __init__(self, num_label: int, d_model: int = 256, nhead: int = 4, num_layers: int = 4, max_bbox: int = 50,
                 device: Optional[int] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("layout_dm_amd needs a ROCm GPU (MI355X); there is no CPU path")
        self.lib = _declare(load_library())
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        self.num_label, self.max_bbox, self.d_model = num_label, max_bbox, d_model
        h = C.c_void_p()
        rc = self.lib.ldm_fid_create(num_label, max_bbox, d_model, nhead, num_layers, self.device_index, C.byref(h))
        if rc != 0:
            raise RuntimeError(f"ldm_fid_create failed ({rc}): {self.lib.ldm_fid_last_error(None).decode()}")
        self._h = h

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.lib.ldm_fid_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None):
            self.lib.ldm_fid_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # nn.Module-ish surface used by eval.py / load_fidnet_v3
    def to(self, *_a, **_k):
        return self

    def eval(self):
        return self

    def load_state_dict(self, state_dict: Dict[str, "torch.Tensor | np.ndarray"], strict: bool = True):
        for k, v in state_dict.items():
            if isinstance(v, torch.Tensor):
                if not v.dtype.is_floating_point:
                    continue  # enc_transformer.token_mask (bool buffer)
                v = v.detach().cpu().numpy()
            a = np.ascontiguousarray(np.asarray(v), dtype=np.float32)
            shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
            self._check(self.lib.ldm_fid_load_weight(self._h, k.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim),
                        f"ldm_fid_load_weight({k})")
        self._check(self.lib.ldm_fid_finalize(self._h), "ldm_fid_finalize")
        return self

    @torch.no_grad()
    def extract_features(self, bbox: torch.Tensor, label: torch.Tensor, padding_mask: torch.Tensor) -> torch.Tensor:
        """model.py:147-152.  bbox (B,N,4) float, label (B,N) long, padding_mask (B,N) bool (True = padded).
        Returns the (B, 256) features on the device."""
        bbox = bbox.to(device=self.device, dtype=torch.float32).contiguous()
        label = label.to(device=self.device, dtype=torch.int64).contiguous()
        pm = padding_mask.to(device=self.device, dtype=torch.uint8).contiguous()
        B, N = label.shape
        assert bbox.shape == (B, N, 4) and pm.shape == (B, N)
        if B and (int(label.min()) < 0 or int(label.max()) >= self.num_label):
            # nn.Embedding raises on such an index (fid/model.py:128); the kernel would clamp it silently
            raise IndexError(f"label out of range [0, {self.num_label}): min {int(label.min())}, max {int(label.max())}")
        feat = torch.empty((B, self.d_model), dtype=torch.float32, device=self.device)
        self._check(self.lib.ldm_fid_features(self._h, bbox.data_ptr(), label.data_ptr(), pm.data_ptr(), B, N,
                                              feat.data_ptr(), _stream_ptr(self.device)), "ldm_fid_features")
        torch.cuda.current_stream(self.device).synchronize()  # inputs may be temporaries of the dtype/device casts
        return feat

    def forward(self, *a, **k):
        raise NotImplementedError("only extract_features is implemented (the decoder half trains the extractor)")

    __call__ = forward


def load_fidnet_v3(dataset, weight_dir: str, device=None) -> FIDNetV3:
    """model.py:182-193: <weight_dir>/<name>-max<max_seq_length>/model_best.pth.tar, key "state_dict"."""
    prefix = f"{dataset.name}-max{dataset.max_seq_length}"
    ckpt_path = os.path.join(weight_dir, prefix, "model_best.pth.tar")
    dev = device.index if isinstance(device, torch.device) else device
    model = FIDNetV3(num_label=dataset.num_classes, max_bbox=dataset.max_seq_length, device=dev)
    try:  # the reference opens the checkpoint through fsspec (model.py:187-189): gs:// and s3:// weight dirs work there
        import fsspec

        with fsspec.open(ckpt_path, "rb") as f:
            x = torch.load(f, map_location="cpu", weights_only=False)
    except ImportError:
        x = torch.load(ckpt_path, map_location="cpu", weights_only=False)
    model.load_state_dict(x["state_dict"])
    return model.eval()


def frechet_distance(mu1, sigma1, mu2, sigma2, eps: float = 1e-6) -> float:
    """pytorch_fid 0.2.1 fid_score.calculate_frechet_distance (the dependency pinned by the reference's pyproject,
    imported at helpers/metric.py:11): ||mu1 - mu2||^2 + Tr(S1) + Tr(S2) - 2 Tr(sqrt(S1 S2)), with the same
    singular-product offset and imaginary-component handling.
    (A third-party algorithm, Apache-2.0, restated here because the score IS this formula with these two numerical guards —
    results must equal the reference's to the last digit; the statements follow the published function closely for that reason.)"""
    from scipy import linalg

    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    assert mu1.shape == mu2.shape and sigma1.shape == sigma2.shape
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(covmean).all():
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError("Imaginary component {}".format(np.max(np.abs(covmean.imag))))
        covmean = covmean.real
    return float(diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean))


def _as_array(f) -> np.ndarray:
    if isinstance(f, (list, tuple)):  # list of batch-processed features (helpers/metric.py:25-32)
        return np.concatenate([x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x) for x in f])
    return f.detach().cpu().numpy() if isinstance(f, torch.Tensor) else np.asarray(f)


def compute_prdc(real_features, fake_features, nearest_k: int = 5, device=None) -> dict:
    """prdc.compute_prdc(real_features, fake_features, nearest_k) (prdc ^0.2, imported at helpers/metric.py:10) on the
    device: k-NN radii of both sets and the four counts over the real x fake distance matrix (kernels_prdc.hip)."""
    lib = load_library()
    lib.ldm_prdc.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
    lib.ldm_prdc.restype = C.c_int
    if not torch.cuda.is_available():
        raise RuntimeError("layout_dm_amd needs a ROCm GPU (MI355X); there is no CPU path")
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
    with torch.cuda.device(dev):
        r = torch.as_tensor(_as_array(real_features), dtype=torch.float32).to(dev).contiguous()
        f = torch.as_tensor(_as_array(fake_features), dtype=torch.float32).to(dev).contiguous()
        assert r.dim() == 2 and f.dim() == 2 and r.shape[1] == f.shape[1]
        out = (C.c_float * 4)()
        rc = lib.ldm_prdc(r.data_ptr(), r.shape[0], f.data_ptr(), f.shape[0], r.shape[1], int(nearest_k), out,
                          int(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"ldm_prdc failed ({rc}): need 1 <= nearest_k <= 7, more than nearest_k and at most 65 536 samples "
                           "per set (the pairwise-distance workspace is n^2 floats)")
    return {"precision": float(out[0]), "recall": float(out[1]), "density": float(out[2]), "coverage": float(out[3])}


def compute_generative_model_scores(feats_real, feats_fake) -> dict:
    """compute_generative_model_scores (helpers/metric.py:37-59): precision, recall, density, coverage (nearest_k = 5) and
    FID of two sets of FIDNetV3 features."""
    results = compute_prdc(feats_real, feats_fake, nearest_k=5)
    results["fid"] = compute_fid(feats_real, feats_fake)
    return results


def compute_fid(feats_real, feats_fake) -> float:
    """The "fid" entry of compute_generative_model_scores (helpers/metric.py:37-59)."""
    def arr(f):
        if isinstance(f, (list, tuple)):
            return np.concatenate([x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x) for x in f])
        return f.detach().cpu().numpy() if isinstance(f, torch.Tensor) else np.asarray(f)

    fr, ff = arr(feats_real), arr(feats_fake)
    return frechet_distance(np.mean(fr, axis=0), np.cov(fr, rowvar=False), np.mean(ff, axis=0), np.cov(ff, rowvar=False))


def token_set_features(engine, fid_model: FIDNetV3, tokens, batch: int = 512) -> torch.Tensor:
    """(n, S) token ids -> (n, 256) FIDNetV3 features, all on the device: ids -> {bbox, label, mask} through
    ldm_decode_layouts (what LayoutDM.sample returns, models/layoutdm.py:77-88) -> extract_features on (bbox, label,
    ~mask), i.e. the path trainer/eval.py:203-220 takes from a saved result pickle to its feature list."""
    tokens = torch.as_tensor(tokens)
    feats = []
    for i in range(0, tokens.shape[0], batch):
        dec = engine.decode(tokens[i:i + batch].to(torch.int32))
        feats.append(fid_model.extract_features(dec["bbox"].float(), dec["label"], ~dec["mask"]))
    return torch.cat(feats)


def scores_vs_reference_samples(engine, fid_model: FIDNetV3, tokens_ref_a, tokens_ref_b, tokens_ours) -> dict:
    """BASELINE config 5's acceptance metric ("FID vs reference") as one call: the generative-model scores
    (helpers/metric.py:37-59) of OUR sample set against a reference sample set, next to the same scores between two
    independent reference sample sets — the seed-to-seed spread that says what "the same distribution" measures as at
    this sample size."""
    fa, fb, fo = (token_set_features(engine, fid_model, t) for t in (tokens_ref_a, tokens_ref_b, tokens_ours))
    return {"ref_b_vs_ref_a": compute_generative_model_scores(fa, fb), "ref_a_vs_ref_b": compute_generative_model_scores(fb, fa),
            "ours_vs_ref_a": compute_generative_model_scores(fa, fo), "ours_vs_ref_b": compute_generative_model_scores(fb, fo),
            "n": {"ref_a": int(fa.shape[0]), "ref_b": int(fb.shape[0]), "ours": int(fo.shape[0])}}
