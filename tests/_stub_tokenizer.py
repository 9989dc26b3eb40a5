"""Test helper: the attributes of the reference's LayoutSequenceTokenizer / BboxTokenizer (helpers/layout_tokenizer.py:123-186,
helpers/bbox_tokenizer.py:28-115) that the drop-in's host logic reads, for linear bins.  tests/test_boundary_vs_reference.py checks the
same host logic against the reference's REAL tokenizer where the reference tree is present."""
import numpy as np
import torch


class StubBboxTokenizer:
    """The BboxTokenizer attributes the drop-in reads (helpers/bbox_tokenizer.py:28-115), linear bins: cluster centres for
    the refinement prior and the relation plan, encode() for the canvas box (logit_adjustment.py:38-41).
    (tests/test_boundary_vs_reference.py checks these against the reference's real tokenizer where it is present.)"""
    shared_bbox_vocab, bbox_quantization = "x-y-w-h", "linear"
    var_names = ["x", "y", "w", "h"]
    _var_order = ["x", "y", "w", "h"]

    def __init__(self, n_bin):
        d = 1.0 / n_bin
        mk = lambda a: type("M", (), {"cluster_centers_": a.reshape(-1, 1)})()
        self.clustering_models = {f"x-{n_bin}": mk(np.linspace(0, 1 - d, n_bin)), f"y-{n_bin}": mk(np.linspace(0, 1 - d, n_bin)),
                                  f"w-{n_bin}": mk(np.linspace(d, 1, n_bin)), f"h-{n_bin}": mk(np.linspace(d, 1, n_bin))}
        self.n_bin = n_bin

    def encode(self, bbox):
        d = 1.0 / self.n_bin
        q = torch.zeros_like(bbox)
        q[..., :2] = torch.clamp(bbox[..., :2], 0.0, 1.0 - d)
        q[..., 2:] = torch.clamp(bbox[..., 2:], d, 1.0) - d
        return (self.n_bin * q).round().long() + torch.arange(4) * self.n_bin


class StubTokenizer:
    def __init__(self, spec):
        self.spec = spec
        self.N_category, self.N_bbox_per_var = spec.n_category, spec.n_bin
        self.max_seq_length, self.N_var_per_element = spec.max_elem, spec.n_attr
        self.N_total, self.max_token_length = spec.n_class, spec.seq_len
        self.var_names = ["c", "x", "y", "w", "h"]
        self.special_tokens = ["pad", "mask"]
        self.bbox_tokenizer = StubBboxTokenizer(spec.n_bin)

    def id_to_name(self, i):
        return {self.spec.pad_id: "pad", self.spec.mask_id: "mask"}[i]

    def name_to_id(self, n):
        return {"pad": self.spec.pad_id, "mask": self.spec.mask_id}[n]
