"""Dev tool (CPU): logits error of the split numerics against the float64 restatement, per trained-like weight point — the r03\nconvention (lo scaled by 2^11, two accumulators) next to the r04 one (lo unscaled + power-of-two weight pre-scale: one accumulator)\nand to the unscaled form WITHOUT the pre-scale (what the pre-scale buys).\n    python tools/split_numerics_emulation.py > profiles/r04_split_numerics_cpu_emulation.txt"""
import sys, math, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from oracle import restatement as R, spec as SP, synth
torch.set_num_threads(8)
spec = SP.RICO25
def h(x): return x.half().float()
def split(x, scaled):
    hi = h(x)
    lo = ((x - hi) * 2048.0).half().float() / 2048.0 if scaled else (x - hi).half().float()
    return hi, lo
def mm(a, w, scaled, wscale=True):
    # a @ w.T in split arithmetic; weights optionally pre-scaled by a power of two so that max|w| ~ 1..2
    k = 0
    if wscale and not scaled:
        k = -int(math.floor(math.log2(w.abs().max().item())))
        w = w * (2.0 ** k)
    ah, al = split(a, scaled); wh, wl = split(w, scaled)
    out = ah @ wh.T + (al @ wh.T + ah @ wl.T)
    return out * (2.0 ** -k)
def bmm3(a, b):
    # a @ b in split arithmetic (both operands unscaled hi + lo, three products, fp32 accumulation)
    ah, al = split(a, False); bh, bl = split(b, False)
    return ah @ bh + (al @ bh + ah @ bl)
def fwd(W, tokens, t, scaled, wscale=True, attn=None):
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
        qkv = mm(x, g(b+"self_attn.in_proj_weight"), scaled, wscale) + g(b+"self_attn.in_proj_bias")
        q,k,v = qkv[..., :D], qkv[..., D:2*D], qkv[..., 2*D:]
        q = q.view(B,S,H,dh).transpose(1,2); k = k.view(B,S,H,dh).transpose(1,2); v = v.view(B,S,H,dh).transpose(1,2)
        if attn is None:
            att = torch.softmax((q @ k.transpose(-1,-2))/math.sqrt(dh), dim=-1)
            a = (att @ v).transpose(1,2).reshape(B,S,D)
        else:  # attn16x3_k: raw scores from the split operands, scale inside the exponent, P x attn (2^10 or 1) before its split
            sc = bmm3(q, k.transpose(-1,-2))
            p = torch.exp((sc - sc.max(-1, keepdim=True).values) * (1.0/math.sqrt(dh))) * attn
            a = (bmm3(p, v) / p.sum(-1, keepdim=True)).transpose(1,2).reshape(B,S,D)
        x = x + mm(a, g(b+"self_attn.out_proj.weight"), scaled, wscale) + g(b+"self_attn.out_proj.bias")
        hh = R._ln(x)*g(b+"norm2.weight")+g(b+"norm2.bias")
        hh = torch.relu(mm(hh, g(b+"linear1.weight"), scaled, wscale) + g(b+"linear1.bias"))
        x = x + mm(hh, g(b+"linear2.weight"), scaled, wscale) + g(b+"linear2.bias")
    y = R._ln(x)*g(tr+"head.0.weight")+g(tr+"head.0.bias")
    return mm(y, g(tr+"head.1.weight"), scaled, wscale)
gen = torch.Generator().manual_seed(0)
for point in ("init", "mid", "wide"):
    sd = synth.trained_like_state_dict(spec, point, seed=2)
    W = R.as_torch_weights(sd); W64 = R.as_torch_weights(sd, torch.float64)
    for t in (90, 5):
        tokens = torch.empty(4, spec.seq_len, dtype=torch.long)
        for a in range(spec.n_attr):
            ids = torch.as_tensor(spec.full_ids(a))
            tokens[:, a::spec.n_attr] = ids[torch.randint(0, len(ids)-1, (4, spec.max_elem), generator=gen)]
        tokens[torch.rand(4, spec.seq_len, generator=gen) < t/99] = spec.mask_id
        ref = R.denoiser_logits(W64, spec, tokens, t, dtype=torch.float64); mx = ref.abs().max().item()
        f32 = R.denoiser_logits(W, spec, tokens, t)
        e = lambda o: (o.double()-ref).abs().max().item()/mx
        if len(sys.argv) > 1 and sys.argv[1] == "attention":
            print(f"{point} t={t}: f32 {e(f32):.2e} | r04 split GEMMs + fp32 attention {e(fwd(W,tokens,t,False,True)):.2e} | + split attention, P x 2^10 (attn16x3_k) {e(fwd(W,tokens,t,False,True,1024.0)):.2e} | + split attention, P unscaled {e(fwd(W,tokens,t,False,True,1.0)):.2e}", flush=True)
            continue
        print(f"{point} t={t}: f32 {e(f32):.2e} | split, lo x 2^11, two accumulators (r03) {e(fwd(W,tokens,t,True)):.2e} | lo unscaled + weight pre-scale, one accumulator (r04) {e(fwd(W,tokens,t,False,True)):.2e} | lo unscaled, NO pre-scale {e(fwd(W,tokens,t,False,False)):.2e}")
