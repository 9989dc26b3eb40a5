"""TEST INFRASTRUCTURE ONLY — CPU restatement of FIDNetV3.extract_features (trainer/fid/model.py:123-164), the layout
feature extractor of the reference's FID / precision-recall evaluation, plus the deterministic
synthetic checkpoint the parity tests use.  Pinned by tests/golden/fid_v3.npz (features produced by the REAL
reference class on that checkpoint: oracle/make_golden.py fid_cases)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

D, HEADS, FF, LAYERS = 256, 4, 128, 4  # model.py:124,131-136


def synth_fid_state_dict(num_label: int, seed: int = 0, max_bbox: int = 25):
    """The encoder-half keys of FIDNetV3.state_dict() with seeded numpy values (weights ~N(0, 0.05), biases ~N(0, 0.1),
    LayerNorm affine around 1 / 0 — nothing trivially zero or one)."""
    rng = np.random.default_rng(seed + 4242)
    w = lambda *s: (rng.standard_normal(s) * 0.05).astype(np.float32)
    b = lambda n: (rng.standard_normal(n) * 0.1).astype(np.float32)
    g = lambda n: (1.0 + rng.standard_normal(n) * 0.1).astype(np.float32)
    sd = {"emb_label.weight": (rng.standard_normal((num_label, D)) * 0.5).astype(np.float32),
          "fc_bbox.weight": (rng.standard_normal((D, 4)) * 0.5).astype(np.float32), "fc_bbox.bias": b(D),
          "enc_fc_in.weight": w(D, 2 * D), "enc_fc_in.bias": b(D),
          "enc_transformer.token": rng.standard_normal((1, 1, D)).astype(np.float32)}
    for i in range(LAYERS):
        p = f"enc_transformer.core.layers.{i}."
        sd[p + "self_attn.in_proj_weight"] = w(3 * D, D)
        sd[p + "self_attn.in_proj_bias"] = b(3 * D)
        sd[p + "self_attn.out_proj.weight"] = w(D, D)
        sd[p + "self_attn.out_proj.bias"] = b(D)
        sd[p + "linear1.weight"] = w(FF, D)
        sd[p + "linear1.bias"] = b(FF)
        sd[p + "linear2.weight"] = w(D, FF)
        sd[p + "linear2.bias"] = b(D)
        sd[p + "norm1.weight"], sd[p + "norm1.bias"] = g(D), b(D)
        sd[p + "norm2.weight"], sd[p + "norm2.bias"] = g(D), b(D)
    return sd


def synth_layouts(num_label: int, B: int, N: int, seed: int = 0):
    """Random layouts: n ~ U{0..N} valid elements (one layout with none, one full), boxes in [0,1]."""
    rng = np.random.default_rng(seed + 99)
    n = rng.integers(0, N + 1, size=B)
    n[0], n[-1] = 0, N
    mask = np.arange(N)[None] < n[:, None]
    bbox = rng.random((B, N, 4)).astype(np.float32)
    label = rng.integers(0, num_label, size=(B, N)).astype(np.int64)
    return bbox, label, ~mask  # padding_mask: True = padded (eval.py hands ~mask)


def extract_features(sd, bbox, label, padding_mask, dtype=torch.float32):
    """model.py:147-152 + TransformerWithToken.forward (l.26-41) + torch.nn.TransformerEncoderLayer defaults
    (post-norm, ReLU, eps 1e-5; dropout inactive in eval)."""
    W = {k: torch.as_tensor(v).to(dtype) for k, v in sd.items()}
    bbox, label = torch.as_tensor(bbox).to(dtype), torch.as_tensor(label).long()
    pm = torch.as_tensor(padding_mask).bool()
    B, N = label.shape
    b = F.linear(bbox, W["fc_bbox.weight"], W["fc_bbox.bias"])                          # l.148
    l = W["emb_label.weight"][label]                                                       # l.149
    x = torch.relu(F.linear(torch.cat([b, l], dim=-1), W["enc_fc_in.weight"], W["enc_fc_in.bias"]))  # l.150-151
    x = torch.cat([W["enc_transformer.token"].view(1, 1, D).expand(B, 1, D), x], dim=1)    # l.33-34 (batch-first here)
    keep = torch.cat([torch.ones(B, 1, dtype=torch.bool), ~pm], dim=1)                    # l.36-37
    S = N + 1
    for i in range(LAYERS):
        p = f"enc_transformer.core.layers.{i}."
        qkv = F.linear(x, W[p + "self_attn.in_proj_weight"], W[p + "self_attn.in_proj_bias"])
        q, k, v = (t.view(B, S, HEADS, D // HEADS).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
        sc = (q @ k.transpose(-1, -2)) / (D // HEADS) ** 0.5
        sc = sc.masked_fill(~keep[:, None, None, :], float("-inf"))
        a = (torch.softmax(sc, dim=-1) @ v).transpose(1, 2).reshape(B, S, D)
        a = F.linear(a, W[p + "self_attn.out_proj.weight"], W[p + "self_attn.out_proj.bias"])
        x = F.layer_norm(x + a, (D,), W[p + "norm1.weight"], W[p + "norm1.bias"], 1e-5)
        f = F.linear(torch.relu(F.linear(x, W[p + "linear1.weight"], W[p + "linear1.bias"])),
                     W[p + "linear2.weight"], W[p + "linear2.bias"])
        x = F.layer_norm(x + f, (D,), W[p + "norm2.weight"], W[p + "norm2.bias"], 1e-5)
    return x[:, 0]                                                                         # l.152


def compute_prdc(real_features, fake_features, nearest_k: int = 5):
    """Restatement of prdc.compute_prdc (package prdc ^0.2 pinned by the reference's pyproject.toml:34, called at
    helpers/metric.py:52-54; NOT vendored under /root/reference and not installed here, so this follows the published
    algorithm — Naeem et al., ICML 2020 — as the package implements it): Euclidean pairwise distances in float64,
    radius = the (nearest_k + 1)-th smallest entry of a set's own distance rows (the smallest is the zero
    self-distance), strict '<' everywhere.  parity unpinned by the reference itself (it ships no vectors for it)."""
    r = np.asarray(real_features, dtype=np.float64)
    f = np.asarray(fake_features, dtype=np.float64)

    def pdist(a, b):
        return np.sqrt(np.maximum(((a[:, None, :] - b[None, :, :]) ** 2).sum(-1), 0.0))

    def radii(x):
        d = pdist(x, x)
        return np.partition(d, nearest_k, axis=-1)[:, :nearest_k + 1].max(axis=-1)

    rr, rf = radii(r), radii(f)
    d = pdist(r, f)
    return {"precision": float((d < rr[:, None]).any(axis=0).mean()),
            "recall": float((d < rf[None, :]).any(axis=1).mean()),
            "density": float((1.0 / nearest_k) * (d < rr[:, None]).sum(axis=0).mean()),
            "coverage": float((d.min(axis=1) < rr).mean())}
