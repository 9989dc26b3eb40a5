"""Dev tool (CPU): which operands of the denoiser need their fp16 lo half?  Logits error against the float64 restatement, per operand format, on
the FITTED checkpoint (oracle/_fit/rico25_fitted.npz: the states of its fixture) and on the synthetic init / mid / wide points.  Each of the 14
operand sites of a block (+ head) is rounded either to fp16 (h) or to hi + lo fp16 (h2 ~ fp32); products accumulate in fp32 like the MFMA.

    python tools/two_product_emulation.py            the engines' formats        python tools/two_product_emulation.py sites      one site at a time

What it predicted and the GPU then measured to three digits (profiles/r06_mixed_mode.txt, profiles/r06_hybrid_mode.txt):
  mixed  = every WEIGHT fp16, every activation hi + lo            (LDM_PREC_MIXED_F16:  2 passes per weight product)
  hybrid = mixed + the FFN and the head in plain fp16             (LDM_PREC_HYBRID_F16: LayerNorm-2 / head-LayerNorm output and the hidden activations
           rounded once; the attention path — AdaLN output, q, k, v, P, attention output — keeps hi + lo: that is where the fp16 error lives)
(tools/one_launch_x3_emulation.py asked the same of the one-launch kernel's formats in r05.)

    python tools/two_product_emulation.py jitter     how far each format's OWN result moves under a 1e-7 relative jitter of every rounded operand

A format that rounds ACTIVATIONS to plain fp16 (fast, hybrid) is chaotic at its own error level: the jitter flips one rounding in 10^4, each flip
moves a residual row by ~1e-5, which flips 2 % of the roundings behind it — two blocks later the rounding pattern is another draw (hybrid, mid: the
emulation moves by 7e-4, its error against float64 is 9e-4).  For those formats this tool predicts the error LEVEL; only where activations keep hi + lo
(mixed: moves by 3e-5; the GPU engine matches the emulation to 1e-6) does it predict the digits."""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import restatement as R, spec as SP, synth  # noqa: E402

torch.set_num_threads(8)
spec = SP.RICO25
SITES = ["ln1", "Win", "q", "k", "v", "P", "ao", "Wo", "ln2", "W1", "hid", "W2", "hln", "Wh"]
WEIGHTS = ["Win", "Wo", "W1", "W2", "Wh"]
FORMATS = [("fp16 everywhere (fast)", SITES), ("hybrid: weights + ln2, hid, hln fp16", WEIGHTS + ["ln2", "hid", "hln"]),
           ("mixed: weights fp16", WEIGHTS), ("activations fp16, weights hi + lo", [s for s in SITES if s not in WEIGHTS]), ("hi + lo everywhere (split)", [])]


def h(x):
    return x.half().float()


def h2(x):
    hi = x.half().float()
    return hi + (x - hi).half().float()


def fwd(W, tokens, t, f):
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
        qkv = f["ln1"](x) @ f["Win"](g(b + "self_attn.in_proj_weight")).T + g(b + "self_attn.in_proj_bias")
        q, k, v = (f[n](z).view(B, S, H, dh).transpose(1, 2) for n, z in (("q", qkv[..., :D]), ("k", qkv[..., D:2 * D]), ("v", qkv[..., 2 * D:])))
        att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(dh), dim=-1)
        a = (f["P"](att) @ v).transpose(1, 2).reshape(B, S, D)
        x = x + f["ao"](a) @ f["Wo"](g(b + "self_attn.out_proj.weight")).T + g(b + "self_attn.out_proj.bias")
        hh = R._ln(x) * g(b + "norm2.weight") + g(b + "norm2.bias")
        hh = torch.relu(f["ln2"](hh) @ f["W1"](g(b + "linear1.weight")).T + g(b + "linear1.bias"))
        x = x + f["hid"](hh) @ f["W2"](g(b + "linear2.weight")).T + g(b + "linear2.bias")
    y = R._ln(x) * g(tr + "head.0.weight") + g(tr + "head.0.bias")
    return f["hln"](y) @ f["Wh"](g(tr + "head.1.weight")).T


def run(name, sd, states, formats):
    W, W64 = R.as_torch_weights(sd), R.as_torch_weights(sd, torch.float64)
    refs = [R.denoiser_logits(W64, spec, tk, t, dtype=torch.float64) for tk, t in states]
    print(f"[{name}]")
    for label, fp16_sites in formats:
        f = {s: (h if s in fp16_sites else h2) for s in SITES}
        e = [((fwd(W, tk, t, f).double() - ref).abs().max() / ref.abs().max()).item() for (tk, t), ref in zip(states, refs)]
        print(f"   {label:44s} " + " ".join(f"{x:.2e}" for x in e) + f"   max {max(e):.2e}")


def synthetic_states(g, n=4):
    out = []
    for t in (50, 90, 5):
        tokens = torch.empty(n, spec.seq_len, dtype=torch.long)
        for a in range(spec.n_attr):
            ids = torch.as_tensor(spec.full_ids(a))
            tokens[:, a::spec.n_attr] = ids[torch.randint(0, len(ids) - 1, (n, spec.max_elem), generator=g)]
        tokens[torch.rand(n, spec.seq_len, generator=g) < t / 99] = spec.mask_id
        out.append((tokens, t))
    return out


if __name__ == "__main__":
    formats = FORMATS
    if len(sys.argv) > 1 and sys.argv[1] == "jitter":
        sd = synth.trained_like_state_dict(spec, "mid", seed=3)
        W, tokens, t = R.as_torch_weights(sd), synthetic_states(torch.Generator().manual_seed(0))[0][0], 40
        for label, fp16_sites in FORMATS:
            gj = torch.Generator().manual_seed(1)
            f0 = {s: (h if s in fp16_sites else h2) for s in SITES}
            f1 = {s: (lambda fn: (lambda x: fn(x * (1 + 1e-7 * torch.randn(x.shape, generator=gj)))))(f0[s]) for s in SITES}
            a, b = fwd(W, tokens, t, f0).double(), fwd(W, tokens, t, f1).double()
            print(f"   {label:44s} moves by {((a - b).abs().max() / a.abs().max()).item():.2e} under a 1e-7 relative jitter of every rounded operand ('mid', t = 40)")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "sites":
        formats = [("fp16 everywhere (fast)", SITES)] + [(f"hi + lo everywhere except {s}", [s]) for s in SITES]
    g = torch.Generator().manual_seed(0)
    fit = os.path.join(ROOT, "oracle", "_fit", "rico25_fitted.npz")
    if os.path.exists(fit):
        w = np.load(fit)
        gold = np.load(os.path.join(ROOT, "tests", "golden", "rico25_fitted.npz"))
        states = [(torch.from_numpy(gold[f"tokens_{int(t)}"].astype(np.int64)), int(t)) for t in gold["ts"]]
        states += [(torch.from_numpy(gold["states_before"][i].astype(np.int64)), int(gold["steps"][i])) for i in (20, 60, 95, 99)]
        run("fitted checkpoint, states of its fixture (t = 90, 50, 5; trajectory states 20, 60, 95, 99)", {k: w[k] for k in w.files}, states, formats)
    run("init (the reference's own initialisation)", synth.synth_state_dict(spec, seed=0), synthetic_states(g), formats)
    for point in ("mid", "wide"):
        run(f"trained-like '{point}'", synth.trained_like_state_dict(spec, point, seed=3), synthetic_states(g), formats)
