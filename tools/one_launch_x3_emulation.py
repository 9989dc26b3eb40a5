"""Dev tool (CPU): what a ONE-LAUNCH fp16 x 3 kernel could drop (routes (a) / (b)) — logits error against the
float64 restatement on the trained-like weight points, per rounding site, with the operand formats a layout-resident kernel could
afford.  Part 1 (Rico25 mid / wide, t = 50 / 90 / 5): fp32; fp16 activations and weights (the fast mode); fp16 on ONE side only;
hi + lo fp16 on both sides (the split mode's operand format); hi fp16 + lo fp8-e4m3 (per-tensor scaled) on both sides; fp16
activations with hi + lo weights and the q / k path in full precision (route (b)).  Part 2 (both vocabularies, init / mid / wide):
the fast mode with every LayerNorm's affine shift folded out of the fp16 GEMM operand (W (n g + s) = W (n g) + W s, W s exact).
Result (profiles/r05_one_launch_x3_emulation.txt): activation side and weight side contribute equally (either one alone in fp16
already exceeds 1e-3 at 'wide'), hi + fp8-lo fails at 'wide' (1.7e-3), route (b) does not reach 1e-3, folding the shifts changes
nothing — only hi + lo fp16 on BOTH operands of every GEMM meets the tolerance, i.e. the full split format.
The oracle's float64 denoiser is the reference here; nothing of this runs in the product."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import restatement as R, spec as SP, synth

torch.set_num_threads(8)


def part1():
    spec = SP.RICO25
    def h(x): return x.half().float()
    def h2(x):  # hi + lo fp16
        hi = x.half().float(); return hi + (x-hi).half().float()
    def f8(x):  # e4m3 emulate
        return x.to(torch.float8_e4m3fn).float()
    def hl8(x):
        hi = x.half().float(); lo = x-hi
        # scale lo per-tensor into fp8 range
        s = lo.abs().max().clamp_min(1e-30)/200.0
        return hi + f8(lo/s)*s
    def fwd(W, spec, tokens, t, fa, fw, sites=None):
        # fa: activation rounding fn, fw: weight rounding fn
        D,H,dh = spec.d_model, spec.n_head, spec.d_head
        B,S = tokens.shape
        g = lambda k: W[k]
        tr = "transformer."
        s_idx = torch.arange(S)
        pos = g(tr+"pos_emb.elem_emb")[s_idx//spec.n_attr] + g(tr+"pos_emb.attr_emb")[s_idx%spec.n_attr]
        x = g(tr+"cat_emb.weight")[tokens] + pos
        def A(name, v): 
            f = fa.get(name, fa.get('*')) if isinstance(fa, dict) else fa
            return f(v)
        def Wt(name, v):
            f = fw.get(name, fw.get('*')) if isinstance(fw, dict) else fw
            return f(v)
        for i in range(spec.n_layer):
            b = f"{tr}backbone.layers.{i}."
            e = g(b+"norm1.emb.weight")[t]; e = e*torch.sigmoid(e)
            ss = g(b+"norm1.linear.weight") @ e + g(b+"norm1.linear.bias")
            scale, shift = ss[:D], ss[D:]
            x = R._ln(x)*(1+scale)+shift
            Wi = g(b+"self_attn.in_proj_weight"); bi = g(b+"self_attn.in_proj_bias")
            qk = A("inproj_qk", x) @ Wt("inproj_qk", Wi[:2*D]).T + bi[:2*D]
            v = A("inproj_v", x) @ Wt("inproj_v", Wi[2*D:]).T + bi[2*D:]
            q,k = qk[..., :D], qk[..., D:]
            q = A("qk", q).view(B,S,H,dh).transpose(1,2); k = A("qk", k).view(B,S,H,dh).transpose(1,2); v = A("pv", v).view(B,S,H,dh).transpose(1,2)
            sc = (q @ k.transpose(-1,-2))/math.sqrt(dh)
            att = torch.softmax(sc, dim=-1)
            a = (A("pv", att) @ v).transpose(1,2).reshape(B,S,D)
            x = x + A("out", a) @ Wt("out", g(b+"self_attn.out_proj.weight")).T + g(b+"self_attn.out_proj.bias")
            hh = R._ln(x)*g(b+"norm2.weight")+g(b+"norm2.bias")
            hh = torch.relu(A("ffn1", hh) @ Wt("ffn1", g(b+"linear1.weight")).T + g(b+"linear1.bias"))
            x = x + A("ffn2", hh) @ Wt("ffn2", g(b+"linear2.weight")).T + g(b+"linear2.bias")
        y = R._ln(x)*g(tr+"head.0.weight")+g(tr+"head.0.bias")
        return A("head", y) @ Wt("head", g(tr+"head.1.weight")).T
    ident = lambda x: x
    g = torch.Generator().manual_seed(0)
    for point, sigma in (("mid",0.06),("wide",0.15)):
        sd = synth.trained_like_state_dict(spec, point, seed=3); W = R.as_torch_weights(sd); W64 = R.as_torch_weights(sd, torch.float64)
        for t in (50, 90, 5):
            tokens = torch.empty(4, spec.seq_len, dtype=torch.long)
            for a in range(spec.n_attr):
                ids = torch.as_tensor(spec.full_ids(a))
                tokens[:, a::spec.n_attr] = ids[torch.randint(0, len(ids)-1, (4, spec.max_elem), generator=g)]
            tokens[torch.rand(4, spec.seq_len, generator=g) < t/99] = spec.mask_id
            ref = R.denoiser_logits(W64, spec, tokens, t, dtype=torch.float64); mx = ref.abs().max().item()
            def rel(o): return (o-ref).abs().max().item()/mx
            print(point, t, "f32", f"{rel(fwd(W,spec,tokens,t,ident,ident)):.2e}",
                  "A16,W16", f"{rel(fwd(W,spec,tokens,t,h,h)):.2e}",
                  "A16 only", f"{rel(fwd(W,spec,tokens,t,h,ident)):.2e}",
                  "W16 only", f"{rel(fwd(W,spec,tokens,t,ident,h)):.2e}",
                  "A hi+lo, W hi+lo", f"{rel(fwd(W,spec,tokens,t,h2,h2)):.2e}",
                  "A hi+lo8, W hi+lo8", f"{rel(fwd(W,spec,tokens,t,hl8,hl8)):.2e}",
                  "A16, W hi+lo, qkpath full", f"{rel(fwd(W,spec,tokens,t,{'*':h,'inproj_qk':ident,'qk':ident},h2)):.2e}",
                  flush=True)


def part2():
    def h(x): return x.half().float()
    def fwd_fold(W, spec, tokens, t, fold):
        """fast-mode emulation; fold=True: the affine shift of each LayerNorm is kept OUT of the fp16 GEMM operand: W (n g + s) = W (n g) + W s,
        W s exact (host, per (t, layer))"""
        D,H,dh = spec.d_model, spec.n_head, spec.d_head
        B,S = tokens.shape
        g = lambda k: W[k]
        tr = "transformer."
        s_idx = torch.arange(S)
        pos = g(tr+"pos_emb.elem_emb")[s_idx//spec.n_attr] + g(tr+"pos_emb.attr_emb")[s_idx%spec.n_attr]
        x = g(tr+"cat_emb.weight")[tokens] + pos
        def lin(n, gain, shift, Wm, b):
            if fold:
                return h(n*gain) @ h(Wm).T + (shift.double() @ Wm.double().T).float() + b
            return h(n*gain + shift) @ h(Wm).T + b
        stats = []
        for i in range(spec.n_layer):
            b = f"{tr}backbone.layers.{i}."
            e = g(b+"norm1.emb.weight")[t]; e = e*torch.sigmoid(e)
            ss = g(b+"norm1.linear.weight") @ e + g(b+"norm1.linear.bias")
            scale, shift = ss[:D], ss[D:]
            n = R._ln(x)
            xa = n*(1+scale)+shift
            stats.append((float((n*(1+scale)).abs().mean()), float(shift.abs().mean())))
            qkv = lin(n, 1+scale, shift, g(b+"self_attn.in_proj_weight"), g(b+"self_attn.in_proj_bias"))
            q,k,v = qkv[..., :D], qkv[..., D:2*D], qkv[..., 2*D:]
            q = h(q).view(B,S,H,dh).transpose(1,2); k = h(k).view(B,S,H,dh).transpose(1,2); v = h(v).view(B,S,H,dh).transpose(1,2)
            att = torch.softmax((q @ k.transpose(-1,-2))/math.sqrt(dh), dim=-1)
            a = (h(att) @ v).transpose(1,2).reshape(B,S,D)
            x = xa + h(a) @ h(g(b+"self_attn.out_proj.weight")).T + g(b+"self_attn.out_proj.bias")
            n2 = R._ln(x)
            hh = torch.relu(lin(n2, g(b+"norm2.weight"), g(b+"norm2.bias"), g(b+"linear1.weight"), g(b+"linear1.bias")))
            x = x + h(hh) @ h(g(b+"linear2.weight")).T + g(b+"linear2.bias")
        n3 = R._ln(x)
        return lin(n3, g(tr+"head.0.weight"), g(tr+"head.0.bias"), g(tr+"head.1.weight"), 0.0), stats
    gen = torch.Generator().manual_seed(0)
    for ds in ("rico25","publaynet"):
      spec = SP.SPECS[ds]
      for point in ("init","mid","wide"):
        sd = synth.trained_like_state_dict(spec, point, seed=2); W = R.as_torch_weights(sd); W64 = R.as_torch_weights(sd, torch.float64)
        for t in (50, 90, 5):
            tokens = torch.empty(4, spec.seq_len, dtype=torch.long)
            for a in range(spec.n_attr):
                ids = torch.as_tensor(spec.full_ids(a))
                tokens[:, a::spec.n_attr] = ids[torch.randint(0, len(ids)-1, (4, spec.max_elem), generator=gen)]
            tokens[torch.rand(4, spec.seq_len, generator=gen) < t/99] = spec.mask_id
            ref = R.denoiser_logits(W64, spec, tokens, t, dtype=torch.float64); mx = ref.abs().max().item()
            a,st = fwd_fold(W,spec,tokens,t,False); b,_ = fwd_fold(W,spec,tokens,t,True)
            print(ds, point, t, f"fast {((a-ref).abs().max()/mx).item():.2e}  shift-folded {((b-ref).abs().max()/mx).item():.2e}", "|n g|, |shift| per layer:", [(round(x,2),round(y,2)) for x,y in st], flush=True)


if __name__ == "__main__":
    part1()
    part2()
