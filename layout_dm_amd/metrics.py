"""Alignment / overlap metrics of generated layouts on the MI355X — drop-in for `compute_alignment` / `compute_overlap`
of the reference's evaluation (trainer/helpers/metric.py:98-203; eval.py:153-155,203-205 calls them on every generated
batch and sums each entry over the layouts).

    from layout_dm_amd.metrics import compute_alignment, compute_overlap     # same signatures, same dictionary keys

Inputs are what `LayoutDM.sample` / `Engine.decode` return: `bbox` (B,S,4) (xc, yc, w, h), `mask` (B,S) bool.  Tensors
already on the device stay there (the decode kernel's output feeds the metrics kernel directly); CPU tensors are copied
over.  One launch of `layout_metrics_k` (one wavefront per layout) computes all six scores; results come back as float32
tensors on the input's device.  No CPU fallback: without the extension or a GPU this raises.
"""
from __future__ import annotations

from typing import Dict

import torch

from .binding import _stream_ptr, load_library

KEYS = ("alignment-ACLayoutGAN", "alignment-LayoutGAN++", "alignment-NDN",
        "overlap-ACLayoutGAN", "overlap-LayoutGAN++", "overlap-LayoutGAN")


def layout_metrics(bbox: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """(B,6) float32 on the GPU: the six scores of every layout in the order of KEYS (ldm_layout_metrics)."""
    if not torch.cuda.is_available():
        raise RuntimeError("layout_dm_amd.metrics needs a ROCm GPU (MI355X); there is no CPU path")
    lib = load_library()
    dev = bbox.device if bbox.is_cuda else torch.device("cuda", torch.cuda.current_device())
    b = bbox.to(device=dev, dtype=torch.float32).contiguous()
    m = mask.to(device=dev, dtype=torch.uint8).contiguous()
    if b.dim() != 3 or b.shape[-1] != 4 or m.shape != b.shape[:2]:
        raise ValueError(f"bbox must be (B,S,4) and mask (B,S); got {tuple(bbox.shape)}, {tuple(mask.shape)}")
    B, S = m.shape
    out = torch.empty((B, len(KEYS)), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.ldm_layout_metrics(b.data_ptr(), m.data_ptr(), B, S, out.data_ptr(), _stream_ptr(dev))
    if rc != 0:
        raise RuntimeError(f"ldm_layout_metrics failed ({rc}): 1 <= S <= 256 elements per layout" if rc == -1
                           else f"ldm_layout_metrics failed ({rc})")
    torch.cuda.current_stream(dev).synchronize()   # (b / m may be temporaries)
    return out


def _as_dict(out: torch.Tensor, lo: int, like: torch.Tensor) -> Dict[str, torch.Tensor]:
    out = out if like.is_cuda else out.cpu()
    return {k: out[:, lo + i].contiguous() for i, k in enumerate(KEYS[lo:lo + 3])}


def compute_alignment(bbox: torch.Tensor, mask: torch.Tensor) -> Dict[str, torch.Tensor]:
    """helpers/metric.py:98-149."""
    return _as_dict(layout_metrics(bbox, mask), 0, bbox)


def compute_overlap(bbox: torch.Tensor, mask: torch.Tensor) -> Dict[str, torch.Tensor]:
    """helpers/metric.py:152-203."""
    return _as_dict(layout_metrics(bbox, mask), 3, bbox)
