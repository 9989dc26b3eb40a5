"""TEST INFRASTRUCTURE ONLY — CPU restatement (torch-CPU / numpy) of LayoutDM's
discrete-diffusion sampling hot path.  It is the *checker* for the HIP path
(tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg); the product package
`layout_dm_amd/` must never import it.

Parity status: PINNED — tests/test_oracle_golden.py checks every function here against
fixtures under tests/golden/ that were produced by the reference itself
(oracle/make_golden.py, which imports /root/reference via oracle/ref_harness.py), and
tests/test_oracle_vs_reference.py re-checks live whenever /root/reference is present.
(The reference ships no tests / golden vectors of its own: SURVEY.md §4.)

Every function cites the reference lines it restates; paths are relative to
/root/reference/src/trainer/trainer/.  State is kept as integer tokens (B,S) instead of
the reference's (B,C,S) log-one-hot tensors; `index_to_log_onehot` converts when a
function needs the reference's representation.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from .spec import LOG_EPS, VAR_NAMES, ModelSpec
from .synth import strip_prefix

# --------------------------------------------------------------------------- weights


def as_torch_weights(state_dict, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Reference state dict (any of the accepted prefixes) -> {bare key: CPU tensor}."""
    out = {}
    for k, v in strip_prefix(state_dict).items():
        t = torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v)
        out[k] = t.detach().cpu().to(dtype) if t.is_floating_point() else t.detach().cpu()
    return out


# --------------------------------------------------------------------------- denoiser


def adaln_table(W, spec: ModelSpec, dtype=torch.float32) -> torch.Tensor:
    """[T][L][2D] table of (scale, shift): Linear(SiLU(Embedding[t]))
    (models/transformer_utils.py:67-69,80-81).  t is batch-uniform during sampling
    (categorical_diffusion/base.py:351-353) so the table depends on (t, layer) only."""
    rows = []
    for i in range(spec.n_layer):
        b = f"transformer.backbone.layers.{i}.norm1."
        e = W[b + "emb.weight"].to(dtype)  # (T, D)
        e = e * torch.sigmoid(e)  # SiLU
        rows.append(e @ W[b + "linear.weight"].to(dtype).T + W[b + "linear.bias"].to(dtype))
    return torch.stack(rows, dim=1)  # (T, L, 2D)


def _ln(x, eps=1e-5):
    # nn.LayerNorm: biased variance, eps inside the sqrt
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps)


def denoiser_logits(W, spec: ModelSpec, tokens: torch.Tensor, t: int, dtype=torch.float32,
                    return_hidden: bool = False):
    """CategoricalTransformer.forward (models/common/nn_lib.py:191-237) +
    ElementPositionalEmbedding (nn_lib.py:112-127) + TransformerEncoder/Block/AdaLayerNorm
    (models/transformer_utils.py:72-83,165-210,226-246), eval mode (dropout = identity).

    tokens (B,S) int64, t python int (uniform over the batch) -> logits (B,S,C).
    """
    D, H, dh = spec.d_model, spec.n_head, spec.d_head
    B, S = tokens.shape
    g = lambda k: W[k].to(dtype)
    tr = "transformer."
    s_idx = torch.arange(S)
    pos = g(tr + "pos_emb.elem_emb")[s_idx // spec.n_attr] + g(tr + "pos_emb.attr_emb")[s_idx % spec.n_attr]
    x = g(tr + "cat_emb.weight")[tokens] + pos  # nn_lib.py:204,220
    hidden = {}
    for i in range(spec.n_layer):
        b = f"{tr}backbone.layers.{i}."
        e = g(b + "norm1.emb.weight")[t]
        e = e * torch.sigmoid(e)
        ss = g(b + "norm1.linear.weight") @ e + g(b + "norm1.linear.bias")
        scale, shift = ss[:D], ss[D:]  # chunk(2): transformer_utils.py:81
        x = _ln(x) * (1 + scale) + shift  # x REPLACED by its normed value (l.175)
        if return_hidden:
            hidden[f"l{i}.norm1"] = x
        qkv = x @ g(b + "self_attn.in_proj_weight").T + g(b + "self_attn.in_proj_bias")
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        q = q.view(B, S, H, dh).transpose(1, 2)
        k = k.view(B, S, H, dh).transpose(1, 2)
        v = v.view(B, S, H, dh).transpose(1, 2)
        att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(dh), dim=-1)
        a = (att @ v).transpose(1, 2).reshape(B, S, D)
        if return_hidden:
            hidden[f"l{i}.attn"] = a
        x = x + a @ g(b + "self_attn.out_proj.weight").T + g(b + "self_attn.out_proj.bias")
        if return_hidden:
            hidden[f"l{i}.x1"] = x
        h = _ln(x) * g(b + "norm2.weight") + g(b + "norm2.bias")
        h = torch.relu(h @ g(b + "linear1.weight").T + g(b + "linear1.bias"))
        x = x + h @ g(b + "linear2.weight").T + g(b + "linear2.bias")
        if return_hidden:
            hidden[f"l{i}.x2"] = x
    y = _ln(x) * g(tr + "head.0.weight") + g(tr + "head.0.bias")
    logits = y @ g(tr + "head.1.weight").T  # no bias: nn_lib.py:186-189
    if return_hidden:
        return logits, hidden
    return logits


# --------------------------------------------------------------------------- log-space helpers


def log_add_exp(a, b):  # util.py:19-21
    m = torch.max(a, b)
    return m + torch.log(torch.exp(a - m) + torch.exp(b - m))


def index_to_log_onehot(tokens: torch.Tensor, C: int) -> torch.Tensor:  # util.py:34-40
    oh = F.one_hot(tokens, C).permute(0, 2, 1)
    return torch.log(oh.float().clamp(min=1e-30))


def predict_start_from_logits(logits: torch.Tensor) -> torch.Tensor:
    """Tail of predict_start (categorical_diffusion/base.py:131-144): drop the MASK
    logit, log_softmax over the remaining C-1 classes in float64, cast to f32, append
    a -70 row for MASK, clamp to [-70, 0].  (B,S,C) -> (B,C,S)."""
    out = logits[:, :, :-1].permute(0, 2, 1)
    log_pred = F.log_softmax(out.double(), dim=1).float()
    B, _, S = log_pred.shape
    log_pred = torch.cat([log_pred, torch.full((B, 1, S), -70.0)], dim=1)
    return torch.clamp(log_pred, -70, 0)


def q_posterior(W, spec: ModelSpec, log_x_start: torch.Tensor, tokens: torch.Tensor, t: int,
                q_type: str = "constrained") -> torch.Tensor:
    """q_type="vanilla": VanillaMaskAndReplaceDiffusion.q_posterior (categorical_diffusion/vanilla.py:112-151 with
    q_pred 92-110 and q_pred_one_timestep 74-90) — the same recursion over ONE vocabulary of all C classes
    (MASK last, PAD an ordinary class) with the un-prefixed schedule buffers.  Otherwise:

    ConstrainedMaskAndReplaceDiffusion.q_posterior
    (categorical_diffusion/constrained.py:135-206, helpers q_pred 112-133 and
    q_pred_one_timestep 92-110; Converter gather/scatter helpers/layout_tokenizer.py:540-557).

    log_x_start (B,C,S) f32 = predict_start output; tokens (B,S) = x_t; t python int.
    Returns log p_theta(x_{t-1}|x_t) over the FULL vocabulary, (B,C,S), dead classes =
    log(1e-30).  Token form: log_x_t one-hot rows are synthesised per attribute.
    """
    B, C, S = log_x_start.shape
    A, T = spec.n_attr, spec.n_step
    assert 0 <= t < T  # constrained.py:139
    u = (t - 1 + (T + 1)) % (T + 1)  # constrained.py:114
    out = torch.full((B, C, S), LOG_EPS, dtype=torch.float32)  # p_to_f_log fill, layout_tokenizer.py:544
    groups = [(f"{key}_", torch.as_tensor(spec.full_ids(a)), slice(a, None, A)) for a, key in enumerate(VAR_NAMES)]
    if q_type == "vanilla":
        groups = [("", torch.arange(C), slice(None))]
    for prefix, full, pos in groups:  # full: partial -> full id
        K = full.numel()
        buf = lambda n: W[f"{prefix}{n}"].float()
        la, lb, lc = buf("log_at")[t], buf("log_bt")[t], buf("log_ct")[t]
        LA, LB, LC = buf("log_cumprod_at")[t], buf("log_cumprod_bt")[t], buf("log_cumprod_ct")[t]
        LAu, LBu, LCu = buf("log_cumprod_at")[u], buf("log_cumprod_bt")[u], buf("log_cumprod_ct")[u]
        L1Cu = buf("log_1_min_cumprod_ct")[u]

        tok_a = tokens[:, pos]  # (B,E) full ids
        lxs = log_x_start[:, :, pos][:, full, :]  # f_to_p_log: (B,K,E)
        # one-hot log_x_t in the partial vocabulary (util.py:34-40 then gather)
        part = (tok_a.unsqueeze(1) == full.view(1, K, 1))
        lxt = torch.log(part.float().clamp(min=1e-30))
        is_mask = (tok_a == spec.mask_id).unsqueeze(1)  # (B,1,E)

        # q(xt|x0) rows 0..K-2 (constrained.py:166-173)
        log_qt = log_add_exp(lxt[:, :-1, :] + LA, LB)
        log_qt = torch.where(is_mask, LC.expand_as(log_qt), log_qt)
        # q(xt|xt-1) (constrained.py:175-185)
        q1 = log_add_exp(lxt[:, :-1, :] + la, lb)
        q1 = torch.cat([q1, torch.full((B, 1, q1.shape[2]), LOG_EPS)], dim=1)
        ct_vec = torch.cat([lc.expand(B, K - 1, q1.shape[2]), torch.zeros(B, 1, q1.shape[2])], dim=1)
        q1 = torch.where(is_mask, ct_vec, q1)
        # eq.5 of VQ-Diffusion (constrained.py:188-197)
        q = lxs[:, :-1, :] - log_qt
        q = torch.cat([q, torch.full((B, 1, q.shape[2]), LOG_EPS)], dim=1)
        lse = torch.logsumexp(q, dim=1, keepdim=True)
        q = q - lse
        r = torch.cat([
            log_add_exp(q[:, :-1, :] + LAu, LBu),
            log_add_exp(q[:, -1:, :] + L1Cu, LCu),
        ], dim=1)
        ev = torch.clamp(r + q1 + lse, -70, 0)
        # p_to_f_log scatter + interleave (constrained.py:198-204)
        sub = out[:, :, pos]
        sub[:, full, :] = ev
    return out


def apply_cond(spec: ModelSpec, logp: torch.Tensor, cond: Optional[dict], t: Optional[int] = None) -> torch.Tensor:
    """Constraint injection of _sample_single_step (categorical_diffusion/base.py:243-284).  cond=relation's gradient
    update (logit_adjustment.py:88-126, between the refinement prior and the [PAD] disable) runs when the cond dict
    carries its inputs under "relation" = {graph, centres, canvas_bins, lr, num_update} and the model timestep `t` is
    given; without them the relation step is skipped (the update then belongs to the caller)."""
    if not cond:
        return logp
    B, C, S = logp.shape
    seq = torch.as_tensor(cond["seq"])
    if "mask" in cond:
        strong = torch.as_tensor(cond["mask"]).view(B, 1, S)
        logp = torch.where(strong, index_to_log_onehot(seq, C), logp)
    if cond.get("type") == "refinement":
        wm = torch.as_tensor(cond["weak_mask"])
        logp = torch.where(wm, logp + torch.as_tensor(cond["weak_logits"]), logp)
    if cond.get("type") == "relation" and cond.get("relation") is not None and t is not None:
        r = cond["relation"]
        logp = relation_update(spec, logp, seq, r["graph"], r["centres"], r["canvas_bins"], float(r["lr"]),
                               int(r["num_update"]), int(t))
    if cond.get("type") in ("c", "cwh", "refinement", "relation"):
        pos = torch.arange(S).view(1, S)
        pad_mask = (pos % spec.n_attr != 0) & (seq != spec.pad_id)  # (B,S)
        logp = logp.clone()
        logp[:, spec.pad_id, :] = torch.where(pad_mask, torch.tensor(LOG_EPS), logp[:, spec.pad_id, :])
    return logp


# --------------------------------------------------------------------------- sampler

PHILOX_M0, PHILOX_M1 = 0xD2511F53, 0xCD9E8D57
PHILOX_W0, PHILOX_W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32(counter, key, rounds: int = 10):
    """Philox4x32-10 (Salmon et al. 2011), numpy uint64 arithmetic, vectorised over
    the leading axis of `counter` (N,4) / `key` (N,2).  NOT part of the reference (which
    uses torch.multinomial): this is OUR sampler's RNG, restated here so stochastic
    draws of the HIP path can be checked draw-for-draw."""
    c = np.asarray(counter, dtype=np.uint64).copy()
    k = np.asarray(key, dtype=np.uint64).copy()
    M32 = np.uint64(0xFFFFFFFF)
    for _ in range(rounds):
        p0 = np.uint64(PHILOX_M0) * c[:, 0]
        p1 = np.uint64(PHILOX_M1) * c[:, 2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & M32
        hi1, lo1 = p1 >> np.uint64(32), p1 & M32
        c = np.stack([hi1 ^ c[:, 1] ^ k[:, 0], lo1, hi0 ^ c[:, 3] ^ k[:, 1], lo0], axis=1) & M32
        k = np.stack([(k[:, 0] + np.uint64(PHILOX_W0)) & M32, (k[:, 1] + np.uint64(PHILOX_W1)) & M32], axis=1)
    return c.astype(np.uint32)


def token_uniforms(seed: int, first_layout: int, B: int, S: int, step: int, n: int = 1) -> np.ndarray:
    """Uniforms in (0,1) for token (layout, pos) at reverse step `step` (0-based loop
    index): Philox counter = (pos, step, layout_lo, layout_hi), key = (seed_lo, seed_hi).
    u = ((x >> 9) + 0.5) * 2^-23, exact in float32 and strictly inside (0,1) — the same
    expression the kernel evaluates (kernels_post.hip u01).  Returns (B,S,min(n,4))."""
    lay = (np.arange(B, dtype=np.uint64) + np.uint64(first_layout))
    ctr = np.zeros((B, S, 4), np.uint64)
    ctr[..., 0] = np.arange(S, dtype=np.uint64)[None, :]
    ctr[..., 1] = np.uint64(step)
    ctr[..., 2] = (lay & np.uint64(0xFFFFFFFF))[:, None]
    ctr[..., 3] = (lay >> np.uint64(32))[:, None]
    key = np.zeros((B * S, 2), np.uint64)
    key[:, 0] = np.uint64(seed & 0xFFFFFFFF)
    key[:, 1] = np.uint64((seed >> 32) & 0xFFFFFFFF)
    r = philox4x32(ctr.reshape(-1, 4), key).reshape(B, S, 4)
    u = ((r >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -23)
    return u[..., :n]


def filter_logits(logp: torch.Tensor, cfg: dict) -> torch.Tensor:
    """The non-random part of helpers/sampling.py:81-116: temperature, top-k, top-p
    (incl. the quirk that the threshold-crossing class is dropped too, l.101-108)."""
    name = cfg["name"]
    lg = logp / cfg.get("temperature", 1.0)
    if name == "top_k":
        v, _ = torch.topk(lg, cfg["top_k"], 1)
        lg = lg.clone()
        lg[lg < v[:, [-1]]] = -float("inf")
    elif name == "top_p":
        top_p = cfg["top_p"]
        C = lg.size(1)
        s_lg, s_idx = torch.sort(lg, descending=True, dim=1)
        cum = torch.cumsum(F.softmax(s_lg, dim=1), dim=1)
        idx = torch.arange(C).view(1, C, 1)
        s_lg[(cum > top_p) & (idx > 0)] = -float("inf")
        lg = s_lg.gather(dim=1, index=s_idx.argsort(dim=1))
    elif name in ("random", "gumbel"):
        pass
    else:
        raise NotImplementedError(name)
    return lg


def sample_probs(logp: torch.Tensor, cfg: dict) -> torch.Tensor:
    """Class probabilities the reference hands to torch.multinomial
    (helpers/sampling.py:119-127) for the non-gumbel stochastic samplers. (B,C,S)."""
    return F.softmax(filter_logits(logp, cfg), dim=1)


def sample_tokens(logp: torch.Tensor, cfg: dict, uniforms: Optional[np.ndarray] = None,
                  generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """helpers/sampling.py:81-130.  deterministic -> argmax over classes (first max).
    Stochastic: with `uniforms` (B,S) the draw is the inverse-CDF rule of OUR kernel
    (smallest class c with cumsum(p)[c] > u * sum(p)), else torch.multinomial like the
    reference.  Returns (B,S) int64."""
    if cfg["name"] == "deterministic":
        return torch.argmax(logp, dim=1)
    if cfg["name"] == "gumbel" and uniforms is None:
        u = torch.rand(logp.shape, generator=generator)
        logp = logp / cfg.get("temperature", 1.0) + (-torch.log(-torch.log(u + 1e-30) + 1e-30))
        probs = F.softmax(logp, dim=1)
    else:
        probs = sample_probs(logp, cfg)
    B, C, S = probs.shape
    if uniforms is None:
        flat = probs.permute(0, 2, 1).reshape(B * S, C)
        return torch.multinomial(flat, 1, generator=generator).view(B, S)
    cdf = torch.cumsum(probs.double(), dim=1)
    thr = torch.as_tensor(uniforms, dtype=torch.float64).view(B, 1, S) * cdf[:, -1:, :]
    idx = (cdf <= thr).sum(dim=1)
    return idx.clamp(max=C - 1)


# --------------------------------------------------------------------------- loop


def timestep_list(T_model: int, T_eval: int):
    """categorical_diffusion/base.py:310-315."""
    assert T_eval <= T_model
    return [int(i * T_model / T_eval) for i in range(T_eval - 1, -1, -1)]


def single_step(W, spec, tokens, t, cfg, cond=None, skip_step: int = 0, uniforms=None, generator=None,
                dtype=torch.float32, return_all=False, q_type: str = "constrained"):
    """_sample_single_step (categorical_diffusion/base.py:205-291) in token form."""
    logits = denoiser_logits(W, spec, tokens, t, dtype=dtype).float()
    log_x0 = predict_start_from_logits(logits)
    noise_t = t
    td = cfg.get("time_difference", 0.0)
    if td > 0.0:
        noise_t = min(max(t - int(spec.n_step * td), 0), spec.n_step - 1)
    if skip_step > 0 and noise_t > skip_step:
        noise_t = noise_t - skip_step
    logp = q_posterior(W, spec, log_x0, tokens, noise_t, q_type=q_type)
    logp = apply_cond(spec, logp, cond, t)          # the relation update sees the MODEL timestep (base.py:262)
    nxt = sample_tokens(logp, cfg, uniforms=uniforms, generator=generator)
    if return_all:
        return nxt, logits, logp
    return nxt


def sample_loop(W, spec: ModelSpec, batch_size: int, cfg: dict, cond: Optional[dict] = None,
                seed: Optional[int] = None, first_layout: int = 0, generator=None,
                dtype=torch.float32, get_intermediate_results=False, q_type: str = "constrained"):
    """BaseMaskAndReplaceDiffusion.sample (categorical_diffusion/base.py:293-371).
    With `seed` the stochastic draws use the Philox inverse-CDF rule of the HIP kernel
    (keyed by global layout index => independent of batch split); otherwise
    torch.multinomial like the reference."""
    T_eval = cfg.get("num_timesteps", spec.n_step)
    steps = timestep_list(spec.n_step, T_eval)
    if cond:
        tokens = torch.as_tensor(cond["seq"]).clone().long()
        if tokens.shape[0] == 1 and batch_size > 1:  # duplicate_cond, helpers/task.py:235-248
            tokens = tokens.repeat(batch_size, 1)
            cond = {k: (torch.as_tensor(v).repeat([batch_size] + [1] * (torch.as_tensor(v).dim() - 1))
                        if isinstance(v, (np.ndarray, torch.Tensor)) and torch.as_tensor(v).dim() > 0 and torch.as_tensor(v).shape[0] == 1 else v)
                    for k, v in cond.items()}
    else:
        tokens = torch.full((batch_size, spec.seq_len), spec.mask_id, dtype=torch.long)
    prev = spec.n_step
    inter = []
    for i, t in enumerate(steps):
        u = None
        if seed is not None and cfg["name"] != "deterministic":
            u = token_uniforms(seed, first_layout, batch_size, spec.seq_len, i)[..., 0]
        tokens = single_step(W, spec, tokens, t, cfg, cond, skip_step=prev - t - 1, uniforms=u,
                             generator=generator, dtype=dtype, q_type=q_type)
        prev = t
        if get_intermediate_results:
            inter.append(tokens.clone())
    return inter if get_intermediate_results else tokens


# ------------------------------------------------------------------------------------------------
# result packaging: ids -> {bbox, label, mask}
def decode_layouts(spec: ModelSpec, tokens, centres=None) -> Dict[str, torch.Tensor]:
    """LayoutSequenceTokenizer.decode (helpers/layout_tokenizer.py:255-266) with
    _filter_invalid_labels_and_bboxes (l.106-114; no bos/eos => _filter_eos is all-False, l.116-121) and
    BboxTokenizer.decode (helpers/bbox_tokenizer.py:117-168) for var_order c-x-y-w-h / shared_bbox_vocab
    x-y-w-h.  centres: None -> bbox_quantization=linear (float32 boxes, l.141-146); (4,n_bin) float64 ->
    kmeans/percentile (float64 boxes clamped to [0,1], l.148-166)."""
    ids = torch.as_tensor(tokens).long().view(-1, spec.max_elem, spec.n_attr)
    label = ids[..., 0].clone()
    bbox = ids[..., 1:] - spec.n_category
    n_bbox = 4 * spec.n_bin
    valid = (label >= 0) & (label < spec.n_category) & ((bbox >= 0) & (bbox < n_bbox)).all(dim=-1)
    arr = bbox - torch.arange(4) * spec.n_bin           # KEY_MULT_DICT["x-y-w-h"]: y 1, w 2, h 3
    arr = arr.clamp(0, spec.n_bin - 1)
    if centres is None:
        d = 1 / spec.n_bin
        out = torch.zeros(arr.shape, dtype=torch.float32)
        out[..., :2] = arr[..., :2].float() * d
        out[..., 2:] = (arr[..., 2:] + 1).float() * d
    else:
        c = torch.as_tensor(centres, dtype=torch.float64).view(4, spec.n_bin)
        out = torch.stack([c[j][arr[..., j]] for j in range(4)], dim=-1).clamp(0.0, 1.0)
    invalid = ~valid
    label[invalid] = 0
    out[invalid] = 0.0
    return {"bbox": out, "label": label, "mask": valid}


# ------------------------------------------------------------------------------------------------
# cond=relation: logit adjustment by gradient descent on the relational-constraint losses
REL_SIZE_ALPHA = 0.1                                   # data/util.py:30
REL_SIZE = {"unknown": 0, "smaller": 1, "equal": 2, "larger": 3}          # data/util.py:14-18
REL_LOC = {"unknown": 4, "left": 5, "top": 6, "right": 7, "bottom": 8, "center": 9}   # data/util.py:21-27


def relation_expected_boxes(spec: ModelSpec, logp: torch.Tensor, cond_seq: torch.Tensor, centres, canvas_bins):
    """_stochastic_convert, mode="average" (categorical_diffusion/logit_adjustment.py:16-85): per valid node
    (canvas first, then every element whose conditioned category is not PAD) the softmax-expectation of the
    cluster centres over that coordinate's 32-bin sub-vocabulary.  Returns (N_nodes, 4) in x,y,w,h order."""
    B, C, S = logp.shape
    A, N = spec.n_attr, spec.n_bin
    mask = torch.cat([torch.ones(B, 1, dtype=torch.bool), torch.as_tensor(cond_seq)[:, ::A] != spec.pad_id], dim=1)
    per_coord = []
    for i in range(A - 1):
        sl = slice(spec.n_category + i * N, spec.n_category + (i + 1) * N)
        canvas = torch.full((B, N, 1), LOG_EPS, dtype=logp.dtype)   # index_to_log_onehot of the canvas id, util.py:34-40
        canvas[:, int(canvas_bins[i]), 0] = 0.0
        per_coord.append(torch.cat([canvas, logp[:, sl, (i + 1)::A]], dim=2))      # (B, N, E+1)
    logits = torch.stack(per_coord, dim=-1).permute(0, 2, 1, 3)[mask]              # (nodes, N, 4)
    prob = F.softmax(logits, dim=1)
    c = torch.as_tensor(centres, dtype=torch.float64).view(4, N).t().unsqueeze(0).to(prob.dtype)   # (1, N, 4)
    return (prob * c).sum(dim=1)


def relation_costs(bbox: torch.Tensor, y, edge_index, edge_attr, batch, n_graph: int) -> torch.Tensor:
    """The 14 losses of clg/const.py:221-236 (`relation`), each summed per graph -> (n_graph, 14)."""
    src, dst = edge_index[0], edge_index[1]
    from_canvas = y[src] == 0
    relu = torch.relu
    eps = 1e-8

    def less_equal(a, b):   # const.py:48-49
        return relu(a - b)

    def less(a, b):         # const.py:52-53
        return relu(a - b + eps)

    def per_graph(cost, cond):
        cost = cost.masked_fill(~cond, 0)
        return torch.zeros(n_graph, dtype=cost.dtype).index_add(0, batch[src], cost)   # to_dense_adj(...).sum((1,2))

    def bit(v):
        return (edge_attr & (1 << v)) != 0

    out = []
    area = bbox[:, 2] * bbox[:, 3]
    a1, a2 = area[src], area[dst]
    sm, lg = (1 - REL_SIZE_ALPHA) * a1, (1 + REL_SIZE_ALPHA) * a1
    for rel, cost in ((REL_SIZE["smaller"], less_equal(a2, sm)), (REL_SIZE["equal"], less(sm, a2) + less(a2, lg)),
                      (REL_SIZE["larger"], less_equal(lg, a2))):          # const.py:56-106
        for canvas in (False, True):
            out.append(per_graph(cost, (from_canvas == canvas) & bit(rel)))
    yc = bbox[:, 1][dst]                                                     # const.py:109-157
    y_sm, y_lg = 1.0 / 3, 2.0 / 3
    out.append(per_graph(less_equal(yc, y_sm), from_canvas & bit(REL_LOC["top"])))
    out.append(per_graph(less(y_sm, yc) + less(yc, y_lg), from_canvas & bit(REL_LOC["center"])))
    out.append(per_graph(less_equal(y_lg, yc), from_canvas & bit(REL_LOC["bottom"])))
    xc, ycc, w, h = bbox.t()                                                 # convert_xywh_to_ltrb, helpers/util.py:16-22
    l, t, r, b = xc - w / 2, ycc - h / 2, xc + w / 2, ycc + h / 2
    l1, t1, r1, b1, l2, t2, r2, b2 = l[src], t[src], r[src], b[src], l[dst], t[dst], r[dst], b[dst]
    overlap_y = less(t1, b2) + less(t2, b1)                                  # const.py:176-178
    not_canvas = ~from_canvas
    out.append(per_graph(less_equal(b2, t1), not_canvas & bit(REL_LOC["top"])))                       # relation_loc_t
    out.append(per_graph(less_equal(b1, t2), not_canvas & bit(REL_LOC["bottom"])))                    # relation_loc_b
    out.append(per_graph(less_equal(r2, l1) + overlap_y, not_canvas & bit(REL_LOC["left"])))          # relation_loc_l
    out.append(per_graph(less_equal(r1, l2) + overlap_y, not_canvas & bit(REL_LOC["right"])))         # relation_loc_r
    out.append(per_graph(less(l1, r2) + less(l2, r1) + overlap_y, not_canvas & bit(REL_LOC["center"])))  # relation_loc_c
    order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13]
    return torch.stack([out[i] for i in order], dim=-1)


def relation_update(spec: ModelSpec, logp: torch.Tensor, cond_seq, graph: dict, centres, canvas_bins, lr: float,
                    num_update: int, t: int) -> torch.Tensor:
    """logit_adjustment.update (categorical_diffusion/logit_adjustment.py:88-126): `num_update` plain-SGD steps
    (lr = relation_lambda) on the mean of the 14 per-graph losses w.r.t. the whole (B,C,S) log-probability tensor;
    no update for t < 10.  graph: y (nodes,), edge_index (2,E) global node ids, edge_attr (E,), batch (nodes,)."""
    x = logp.detach().clone().requires_grad_(True)
    y, ei = torch.as_tensor(graph["y"]), torch.as_tensor(graph["edge_index"]).long()
    ea, bt = torch.as_tensor(graph["edge_attr"]).long(), torch.as_tensor(graph["batch"]).long()
    n_graph = logp.shape[0]
    for _ in range(0 if t < 10 else num_update):
        x.grad = None
        bbox = relation_expected_boxes(spec, x, cond_seq, centres, canvas_bins)
        if ei.numel() == 0:
            continue
        loss = relation_costs(bbox, y, ei.view(2, -1), ea, bt, n_graph).mean()
        loss.backward()
        with torch.no_grad():
            x -= lr * x.grad
    return x.detach()


# ------------------------------------------------------------------------------------------------------------------
# alignment / overlap of generated layouts (eval.py:153-155,203-205) — restated element by element
def layout_metrics(bbox, mask):
    """compute_alignment + compute_overlap (trainer/helpers/metric.py:98-203) as plain loops over the elements of each
    layout, float32 like the reference.  bbox (B,S,4) (xc, yc, w, h), mask (B,S) bool -> dict of (B,) float32 arrays with the
    reference's six keys.  Quirks kept: in the alignment only the ROW of an invalid element is masked (metric.py:113), so the
    stored boxes of padded slots take part in a valid element's minimum; a minimum of exactly 1.0 counts as 0
    (metric.py:115,138); the overlap zeroes invalid boxes first (metric.py:164) and uses nan_to_num(ai / a1)."""
    import numpy as np

    bbox = np.asarray(bbox, np.float32)
    mask = np.asarray(mask, bool)
    B, S = mask.shape
    f = np.float32
    out = {k: np.zeros(B, np.float32) for k in ("alignment-ACLayoutGAN", "alignment-LayoutGAN++", "alignment-NDN",
                                               "overlap-ACLayoutGAN", "overlap-LayoutGAN++", "overlap-LayoutGAN")}
    for b in range(B):
        xc, yc, w, h = (bbox[b, :, k] for k in range(4))
        hw, hh = w / f(2), h / f(2)
        co = np.stack([xc - hw, xc, xc + hw, yc - hh, yc, yc + hh])          # (6,S): xl xc xr yt yc yb (util.py:16-22)
        sa = sy = so = su = f(0)
        for i in range(S):
            if not mask[b, i]:
                continue
            m = my = f(1)
            l1, r1, t1, b1 = co[0, i], co[2, i], co[3, i], co[5, i]
            a1 = (r1 - l1) * (b1 - t1)
            ar = au = f(0)
            for j in range(S):
                if j == i:
                    continue
                m = min(m, np.abs(co[:, i] - co[:, j]).min())
                if mask[b, j]:
                    my = min(my, np.abs(co[:3, j] - co[:3, i]).min())
                    l_max, r_min = max(l1, co[0, j]), min(r1, co[2, j])
                    t_max, b_min = max(t1, co[3, j]), min(b1, co[5, j])
                    ai = (r_min - l_max) * (b_min - t_max) if (l_max < r_min and t_max < b_min) else f(0)
                    with np.errstate(divide="ignore", invalid="ignore"):
                        ar = f(ar + np.nan_to_num(f(ai) / a1))
                    if j > i:
                        au = f(au + ai)
            m = f(0) if m == 1.0 else m
            sa = f(sa - np.log(f(1) - f(m)))
            sy = f(sy + (f(0) if my == 1.0 else my))
            so, su = f(so + ar), f(su + au)
        nv = f(mask[b].sum())
        with np.errstate(divide="ignore", invalid="ignore"):
            na, no = sa / nv, so / nv
        out["alignment-ACLayoutGAN"][b], out["alignment-LayoutGAN++"][b] = sa, (0.0 if np.isnan(na) else na)
        out["alignment-NDN"][b] = sy
        out["overlap-ACLayoutGAN"][b], out["overlap-LayoutGAN++"][b] = so, (0.0 if np.isnan(no) else no)
        out["overlap-LayoutGAN"][b] = su
    return out
