"""Dev tool (CPU): which fp16 rounding site carries the fast mode's logits error, per trained-like weight point (oracle/synth.py
TRAINED_LIKE)?  An emulation of the fast numerics mode in torch — operands of each GEMM / attention product rounded to fp16,
fp32 accumulate — against the float64 restatement, all sites together and one site at a time.
    python tools/fp16_rounding_sites.py > profiles/r04_fp16_rounding_sites_cpu_emulation.txt"""
import sys, math, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from oracle import restatement as R, spec as SP, synth
torch.set_num_threads(8)
spec = SP.RICO25

def trained_like(spec, seed, sigma):
    point = {0.02: "init", 0.06: "mid", 0.15: "wide"}[sigma]
    return synth.trained_like_state_dict(spec, point, seed=seed)

def h(x): return x.half().float()
def fwd16(W, spec, tokens, t):
    D,H,dh = spec.d_model, spec.n_head, spec.d_head
    B,S = tokens.shape
    g = lambda k: W[k]
    tr = "transformer."
    s_idx = torch.arange(S)
    pos = g(tr+"pos_emb.elem_emb")[s_idx//spec.n_attr] + g(tr+"pos_emb.attr_emb")[s_idx%spec.n_attr]
    x = g(tr+"cat_emb.weight")[tokens] + pos
    for i in range(spec.n_layer):
        b = f"{tr}backbone.layers.{i}."
        e = g(b+"norm1.emb.weight")[t]; e = e*torch.sigmoid(e)
        ss = g(b+"norm1.linear.weight") @ e + g(b+"norm1.linear.bias")
        scale, shift = ss[:D], ss[D:]
        x = R._ln(x)*(1+scale)+shift
        qkv = h(x) @ h(g(b+"self_attn.in_proj_weight")).T + g(b+"self_attn.in_proj_bias")
        q,k,v = qkv[..., :D], qkv[..., D:2*D], qkv[..., 2*D:]
        q = h(q).view(B,S,H,dh).transpose(1,2); k = h(k).view(B,S,H,dh).transpose(1,2); v = h(v).view(B,S,H,dh).transpose(1,2)
        att = torch.softmax((q @ k.transpose(-1,-2))/math.sqrt(dh), dim=-1)
        a = (h(att) @ v).transpose(1,2).reshape(B,S,D)
        x = x + h(a) @ h(g(b+"self_attn.out_proj.weight")).T + g(b+"self_attn.out_proj.bias")
        hh = R._ln(x)*g(b+"norm2.weight")+g(b+"norm2.bias")
        hh = torch.relu(h(hh) @ h(g(b+"linear1.weight")).T + g(b+"linear1.bias"))
        x = x + h(hh) @ h(g(b+"linear2.weight")).T + g(b+"linear2.bias")
    y = R._ln(x)*g(tr+"head.0.weight")+g(tr+"head.0.bias")
    return h(y) @ h(g(tr+"head.1.weight")).T

g = torch.Generator().manual_seed(0)
for sigma in (0.02, 0.06, 0.15):
    sd = trained_like(spec, 3, sigma)
    W = R.as_torch_weights(sd)
    W64 = R.as_torch_weights(sd, torch.float64)
    for t in (90, 50, 5):
        tokens = torch.empty(4, spec.seq_len, dtype=torch.long)
        for a in range(spec.n_attr):
            ids = torch.as_tensor(spec.full_ids(a))
            tokens[:, a::spec.n_attr] = ids[torch.randint(0, len(ids)-1, (4, spec.max_elem), generator=g)]
        tokens[torch.rand(4, spec.seq_len, generator=g) < t/99] = spec.mask_id
        ref = R.denoiser_logits(W64, spec, tokens, t, dtype=torch.float64)
        f32 = R.denoiser_logits(W, spec, tokens, t)
        f16 = fwd16(W, spec, tokens, t)
        e32 = (f32-ref).abs().max().item(); e16=(f16-ref).abs().max().item(); mx=ref.abs().max().item()
        rowmax = ref.abs().amax(-1)
        rowerr = (f16-ref).abs().amax(-1)
        print(f"sigma={sigma} t={t} max|logit|={mx:.2f} std={ref.std():.2f} f32 rel={e32/mx:.2e} f16 rel={e16/mx:.2e}  worst per-row rel={ (rowerr/rowmax).max().item():.2e} min rowmax={rowmax.min().item():.2f}")

print("--- ablation: which rounding site dominates (sigma=0.15 / 0.06, t=50)")
import itertools
def fwd_sites(W, spec, tokens, t, sites):
    hs = lambda name, x: h(x) if name in sites else x
    D,H,dh = spec.d_model, spec.n_head, spec.d_head
    B,S = tokens.shape
    g = lambda k: W[k]
    tr = "transformer."
    s_idx = torch.arange(S)
    pos = g(tr+"pos_emb.elem_emb")[s_idx//spec.n_attr] + g(tr+"pos_emb.attr_emb")[s_idx%spec.n_attr]
    x = g(tr+"cat_emb.weight")[tokens] + pos
    smax = 0
    for i in range(spec.n_layer):
        b = f"{tr}backbone.layers.{i}."
        e = g(b+"norm1.emb.weight")[t]; e = e*torch.sigmoid(e)
        ss = g(b+"norm1.linear.weight") @ e + g(b+"norm1.linear.bias")
        scale, shift = ss[:D], ss[D:]
        x = R._ln(x)*(1+scale)+shift
        Wi = g(b+"self_attn.in_proj_weight"); bi = g(b+"self_attn.in_proj_bias")
        qk = hs("inproj_qk", x) @ hs("inproj_qk", Wi[:2*D]).T + bi[:2*D]
        v = hs("inproj_v", x) @ hs("inproj_v", Wi[2*D:]).T + bi[2*D:]
        q,k = qk[..., :D], qk[..., D:]
        q = hs("qk", q).view(B,S,H,dh).transpose(1,2); k = hs("qk", k).view(B,S,H,dh).transpose(1,2); v = hs("pv", v).view(B,S,H,dh).transpose(1,2)
        sc = (q @ k.transpose(-1,-2))/math.sqrt(dh); smax = max(smax, sc.abs().max().item())
        att = torch.softmax(sc, dim=-1)
        a = (hs("pv", att) @ v).transpose(1,2).reshape(B,S,D)
        x = x + hs("out", a) @ hs("out", g(b+"self_attn.out_proj.weight")).T + g(b+"self_attn.out_proj.bias")
        hh = R._ln(x)*g(b+"norm2.weight")+g(b+"norm2.bias")
        hh = torch.relu(hs("ffn1", hh) @ hs("ffn1", g(b+"linear1.weight")).T + g(b+"linear1.bias"))
        x = x + hs("ffn2", hh) @ hs("ffn2", g(b+"linear2.weight")).T + g(b+"linear2.bias")
    y = R._ln(x)*g(tr+"head.0.weight")+g(tr+"head.0.bias")
    return hs("head", y) @ hs("head", g(tr+"head.1.weight")).T, smax
ALL = ["inproj_qk","inproj_v","qk","pv","out","ffn1","ffn2","head"]
for sigma in (0.06, 0.15):
    sd = trained_like(spec, 3, sigma); W = R.as_torch_weights(sd); W64 = R.as_torch_weights(sd, torch.float64)
    t=50
    tokens = torch.empty(4, spec.seq_len, dtype=torch.long)
    for a in range(spec.n_attr):
        ids = torch.as_tensor(spec.full_ids(a))
        tokens[:, a::spec.n_attr] = ids[torch.randint(0, len(ids)-1, (4, spec.max_elem), generator=g)]
    tokens[torch.rand(4, spec.seq_len, generator=g) < t/99] = spec.mask_id
    ref = R.denoiser_logits(W64, spec, tokens, t, dtype=torch.float64); mx = ref.abs().max().item()
    for s in [[x] for x in ALL] + [ALL, [x for x in ALL if x not in ("inproj_qk","qk")]]:
        out, smax = fwd_sites(W, spec, tokens, t, s)
        print(sigma, s if len(s)<3 else ("ALL" if len(s)==8 else "ALL-but-qk-path"), f"rel={(out-ref).abs().max().item()/mx:.2e} max|score|={smax:.1f}")
