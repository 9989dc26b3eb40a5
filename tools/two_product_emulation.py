"""Dev tool (CPU): would a TWO-product format (hi + lo fp16 activations x fp16-only weights, or the other way round) stay inside the 1e-3 logits
tolerance on the FITTED checkpoint (oracle/_fit/rico25_fitted.npz) and on the synthetic mid / wide points?  Logits error against the float64
restatement, per operand format, on the fixture's own states.  (tools/one_launch_x3_emulation.py asked the same of the one-launch kernel's formats.)
A per-step engine with two MFMAs per weight product would cost 0.71 of the split mode's MFMA work."""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import restatement as R, spec as SP, synth  # noqa: E402

torch.set_num_threads(8)
spec = SP.RICO25


def h(x):
    return x.half().float()


def h2(x):
    hi = x.half().float()
    return hi + (x - hi).half().float()


def fwd(W, tokens, t, fa, fw, fa_attn=None):
    """fa: rounding of the activation operand of the four WEIGHT GEMMs and the head; fa_attn: of q, k, v, P (activation x activation products);
    fw: rounding of the weights."""
    fa_attn = fa_attn or fa
    D, H, dh = spec.d_model, spec.n_head, spec.d_head
    B, S = tokens.shape
    g = lambda k: W[k]  # noqa: E731
    tr = "transformer."
    s_idx = torch.arange(S)
    pos = g(tr + "pos_emb.elem_emb")[s_idx // spec.n_attr] + g(tr + "pos_emb.attr_emb")[s_idx % spec.n_attr]
    x = g(tr + "cat_emb.weight")[tokens] + pos
    for i in range(spec.n_layer):
        b = f"{tr}backbone.layers.{i}."
        e = g(b + "norm1.emb.weight")[t]
        e = e * torch.sigmoid(e)
        ss = g(b + "norm1.linear.weight") @ e + g(b + "norm1.linear.bias")
        x = R._ln(x) * (1 + ss[:D]) + ss[D:]
        qkv = fa(x) @ fw(g(b + "self_attn.in_proj_weight")).T + g(b + "self_attn.in_proj_bias")
        q, k, v = (fa_attn(z).view(B, S, H, dh).transpose(1, 2) for z in (qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]))
        att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(dh), dim=-1)
        a = (fa_attn(att) @ v).transpose(1, 2).reshape(B, S, D)
        x = x + fa(a) @ fw(g(b + "self_attn.out_proj.weight")).T + g(b + "self_attn.out_proj.bias")
        hh = R._ln(x) * g(b + "norm2.weight") + g(b + "norm2.bias")
        hh = torch.relu(fa(hh) @ fw(g(b + "linear1.weight")).T + g(b + "linear1.bias"))
        x = x + fa(hh) @ fw(g(b + "linear2.weight")).T + g(b + "linear2.bias")
    y = R._ln(x) * g(tr + "head.0.weight") + g(tr + "head.0.bias")
    return fa(y) @ fw(g(tr + "head.1.weight")).T


def run(name, sd, states):
    W, W64 = R.as_torch_weights(sd), R.as_torch_weights(sd, torch.float64)
    res = {}
    for tokens, t in states:
        ref = R.denoiser_logits(W64, spec, tokens, t, dtype=torch.float64)
        mx = ref.abs().max().item()
        for label, fa, fw, faa in (("fp16 x fp16 (fast)", h, h, h), ("hi+lo act x fp16 w, attention hi+lo", h2, h, h2),
                                   ("fp16 act x hi+lo w, attention hi+lo", h, h2, h2), ("hi+lo x hi+lo (split)", h2, h2, h2)):
            err = ((fwd(W, tokens, t, fa, fw, faa).double() - ref).abs().max() / mx).item()
            res.setdefault(label, []).append(err)
    print(f"[{name}]")
    for label, v in res.items():
        print(f"   {label:40s} " + "  ".join(f"{e:.2e}" for e in v) + f"   max {max(v):.2e}")


if __name__ == "__main__":
    g = torch.Generator().manual_seed(0)
    fit = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_fit", "rico25_fitted.npz")
    if os.path.exists(fit):
        w = np.load(fit)
        gold = np.load(os.path.join(os.path.dirname(fit), "..", "..", "tests", "golden", "rico25_fitted.npz"))
        states = [(torch.from_numpy(gold[f"tokens_{int(t)}"].astype(np.int64)), int(t)) for t in gold["ts"]]
        states += [(torch.from_numpy(gold["states_before"][i].astype(np.int64)), int(gold["steps"][i])) for i in (20, 60, 95, 99)]
        run("fitted checkpoint, states of its fixture (t = 90, 50, 5; trajectory states 20, 60, 95, 99)", {k: w[k] for k in w.files}, states)
    for point in ("mid", "wide"):
        sd = synth.trained_like_state_dict(spec, point, seed=3)
        states = []
        for t in (50, 90, 5):
            tokens = torch.empty(4, spec.seq_len, dtype=torch.long)
            for a in range(spec.n_attr):
                ids = torch.as_tensor(spec.full_ids(a))
                tokens[:, a::spec.n_attr] = ids[torch.randint(0, len(ids) - 1, (4, spec.max_elem), generator=g)]
            tokens[torch.rand(4, spec.seq_len, generator=g) < t / 99] = spec.mask_id
            states.append((tokens, t))
        run(f"trained-like '{point}'", sd, states)
