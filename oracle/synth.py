"""TEST INFRASTRUCTURE ONLY — deterministic synthetic checkpoints with the reference's
state-dict key layout and the reference's init distributions
(trainer/models/base_model.py:108-116: Linear/Embedding ~ N(0, 0.02), biases 0,
LayerNorm gamma 1 / beta 0; trainer/models/common/nn_lib.py:109-110: pos-emb U[0,1)).

Values come from numpy's PCG64 so they are identical in the build container and on
the GPU box (torch's CPU normal_ stream is not guaranteed to be ISA-independent).
`perturb=True` randomises biases / LN affines so that no term of the forward pass is
trivially 0 or 1 — used for golden vectors and parity tests.
"""
from __future__ import annotations

import numpy as np

from .spec import ModelSpec, schedule_buffers

PREFIX = "model.module."  # CustomDataParallel wrapper, models/layoutdm.py:52


def synth_state_dict(spec: ModelSpec, seed: int = 0, perturb: bool = False, prefix: str = PREFIX,
                     weight_std: float = 0.02, q_type: str = "constrained"):
    """Returns {key: np.ndarray(float32)} with the 100 reference keys."""
    rng = np.random.default_rng(seed)
    D, F, C, T = spec.d_model, spec.d_ff, spec.n_class, spec.n_step

    def normal(*shape):
        return (rng.standard_normal(shape) * weight_std).astype(np.float32)

    def bias(n):
        if perturb:
            return (rng.standard_normal(n) * 0.1).astype(np.float32)
        return np.zeros(n, np.float32)

    def gamma(n):
        if perturb:
            return (1.0 + rng.standard_normal(n) * 0.1).astype(np.float32)
        return np.ones(n, np.float32)

    sd = {}
    sd["Lt_history"] = np.zeros(T, np.float32)
    sd["Lt_count"] = np.zeros(T, np.float32)
    sd.update(schedule_buffers(spec, q_type))
    tr = "transformer."
    sd[tr + "cat_emb.weight"] = normal(C, D)
    sd[tr + "pos_emb.elem_emb"] = rng.random((spec.max_elem, D)).astype(np.float32)
    sd[tr + "pos_emb.attr_emb"] = rng.random((spec.n_attr, D)).astype(np.float32)
    for i in range(spec.n_layer):
        b = f"{tr}backbone.layers.{i}."
        sd[b + "self_attn.in_proj_weight"] = normal(3 * D, D)
        sd[b + "self_attn.in_proj_bias"] = bias(3 * D)
        sd[b + "self_attn.out_proj.weight"] = normal(D, D)
        sd[b + "self_attn.out_proj.bias"] = bias(D)
        sd[b + "linear1.weight"] = normal(F, D)
        sd[b + "linear1.bias"] = bias(F)
        sd[b + "linear2.weight"] = normal(D, F)
        sd[b + "linear2.bias"] = bias(D)
        sd[b + "norm1.emb.weight"] = normal(T, D)
        sd[b + "norm1.linear.weight"] = normal(2 * D, D)
        sd[b + "norm1.linear.bias"] = bias(2 * D)
        sd[b + "norm2.weight"] = gamma(D)
        sd[b + "norm2.bias"] = bias(D)
    sd[tr + "head.0.weight"] = gamma(D)
    sd[tr + "head.0.bias"] = bias(D)
    sd[tr + "head.1.weight"] = normal(C, D)
    return {prefix + k: v for k, v in sd.items()}


# "trained-like" weight distributions (VERDICT r3 next #1a): the reference's init (sigma = 0.02, gamma = 1, beta = 0) is ONE
# point of the space a checkpoint can sit in, and the easy one for reduced-precision arithmetic: logits of magnitude 2,
# attention scores below 1.  A trained LayoutDM has wider Linear weights, LayerNorm gains spread around 1, a few outlier
# channels in the residual writes and large AdaLN timestep embeddings.  No trained checkpoint is available offline
# (SURVEY section 8c), so these three points stand in for it; make_golden.py runs the REAL reference on each of them.
TRAINED_LIKE = {
    #            sigma of every Linear / Embedding weight; measured on the reference: max |logit|, max |attention score|
    "init":   dict(weight_std=0.02),      # logits ~2,  scores ~1   (same family as synth_state_dict(perturb=True))
    "mid":    dict(weight_std=0.06),      # logits ~4,  scores ~10  (what trained transformers of this size look like)
    "wide":   dict(weight_std=0.15),      # logits ~10, scores >200 (saturated softmax rows: the stress point)
}


def trained_like_state_dict(spec: ModelSpec, point: str, seed: int = 0, prefix: str = PREFIX, ln_spread: float = 0.5,
                            n_outlier: int = 4, outlier_gain: float = 8.0, adaln_gain: float = 5.0):
    """synth_state_dict(perturb=True) at TRAINED_LIKE[point]'s sigma, then: LayerNorm gains ~ U[1 - ln_spread,
    1 + ln_spread] (norm2, head LN), `n_outlier` output channels of every residual write (out_proj, linear2) scaled by
    `outlier_gain`, AdaLN timestep embeddings scaled by `adaln_gain`."""
    sd = synth_state_dict(spec, seed=seed, perturb=True, prefix="", **TRAINED_LIKE[point])
    rng = np.random.default_rng(seed + 1000)
    for k in sorted(sd):
        v = sd[k]
        if k.endswith("norm2.weight") or k.endswith("head.0.weight"):
            sd[k] = (1.0 + ln_spread * rng.uniform(-1.0, 1.0, v.shape)).astype(np.float32)
        elif k.endswith("norm1.emb.weight"):
            sd[k] = (v * adaln_gain).astype(np.float32)
        elif k.endswith("linear2.weight") or k.endswith("out_proj.weight"):
            ch = rng.choice(v.shape[0], n_outlier, replace=False)
            v = v.copy()
            v[ch] *= outlier_gain
            sd[k] = v
    return {prefix + k: v for k, v in sd.items()}


def strip_prefix(sd):
    """Accept either LayoutDM ('model.module.') or bare diffusion-module keys."""
    out = {}
    for k, v in sd.items():
        for p in (PREFIX, "module.", "model."):
            if k.startswith(p):
                k = k[len(p):]
                break
        out[k] = v
    return out


def synth_cond_c(spec: ModelSpec, batch: int, seed: int = 0):
    """Synthetic cond=c inputs shaped exactly like helpers/task.py:94-110 output:
    categories kept, other attrs of valid elements = mask_id, padded elements = pad_id;
    mask=True on category slots of valid elements and on every slot of padded elements."""
    rng = np.random.default_rng(seed + 77)
    S, A = spec.seq_len, spec.n_attr
    seq = np.full((batch, S), spec.pad_id, np.int64)
    mask = np.ones((batch, S), bool)
    n_elem = rng.integers(1, spec.max_elem + 1, size=batch)
    for b in range(batch):
        n = int(n_elem[b])
        cats = rng.integers(0, spec.n_category, size=n)
        for e in range(n):
            seq[b, e * A] = cats[e]
            seq[b, e * A + 1:(e + 1) * A] = spec.mask_id
            mask[b, e * A + 1:(e + 1) * A] = False
    return {"seq": seq, "mask": mask, "type": "c", "num_element": n_elem}
