"""TEST INFRASTRUCTURE ONLY — a checkpoint that was actually TRAINED, by the reference's own code.

BASELINE config 5 names pretrained weights; none exist offline, and oracle/synth.py TRAINED_LIKE only scales random
tensors.  This script fits the REAL reference model (imported from /root/reference through oracle/ref_harness.py) with the
reference's OWN training step:

  loss        ConstrainedMaskAndReplaceDiffusion.forward           trainer/models/categorical_diffusion/constrained.py:232-333
              (vb_stochastic + auxiliary loss, importance-sampled timesteps; LayoutDM.forward sums / means it: layoutdm.py:69-75)
  step        zero_grad -> sum(losses) -> backward -> clip_grad_norm_(1.0) -> optimizer.step     trainer/main.py:208-268
  optimizer   AdamW(lr 5e-4 [config/experiment/layoutdm.yaml], betas (0.9, 0.98) [config/optimizer/adamw.yaml]),
              weight_decay 0.1 [hydra_configs.py:57] on Linear / MHA weights only: BaseModel.optim_groups, base_model.py:54-103,
              with the positional embedding in the no-decay group as LayoutDM.optim_groups does (layoutdm.py:115-126)
  batch       64 [hydra_configs.py:66], tokenised by the reference's LayoutSequenceTokenizer.encode

on STRUCTURED synthetic layouts (there is no dataset offline): rows x columns grids with per-row heights, column-aligned
x / w, categories tied to the row band and the size class — so that attention has something to learn (alignment between
elements of a row / column, category <-> geometry).  Bounded: <= 2 000 steps, <= 1 h of CPU.

Outputs
  oracle/_fit/rico25_fitted.npz      the fitted state dict, fp32 (50 MB: git-ignored build output like oracle/_ref/, it
                                     travels to the GPU box with the snapshot); keys as the reference's checkpoint
  tests/golden/rico25_fitted.npz     small, committed: sha256 of the weight file, the loss curve, per-tensor statistics
                                     (sigma, max |w|), max |logit| / max |attention score| / the reference's own f32 noise floor,
                                     teacher-forced logits + posterior at three timesteps and a 100-state stochastic
                                     trajectory with the reference's greedy answers and margins (make_golden.trained_like's form)

Run:  python -m oracle.make_trained_fixture [--steps 1500] [--threads 6]
"""
from __future__ import annotations

import argparse
import hashlib
import os
import time

import numpy as np
import torch

from . import ref_harness as rh
from . import spec as SP

HERE = os.path.dirname(os.path.abspath(__file__))
FIT_DIR = os.path.join(HERE, "_fit")
WEIGHTS = os.path.join(FIT_DIR, "rico25_fitted.npz")
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden", "rico25_fitted.npz")


def structured_layouts(n: int, n_category: int, seed: int, max_elem: int = 25):
    """n layouts as dense (label, bbox, mask) tensors: grids of 1-4 columns x 1-6 rows inside a margin, a header band on top
    of most of them; boxes are (xc, yc, w, h) in [0, 1].  Category = f(row band, size class) + a little noise."""
    g = np.random.default_rng(seed)
    label = np.zeros((n, max_elem), np.int64)
    bbox = np.zeros((n, max_elem, 4), np.float32)
    mask = np.zeros((n, max_elem), bool)
    for i in range(n):
        cols = int(g.integers(1, 5))
        rows = int(g.integers(1, 7))
        margin = float(g.choice([0.04, 0.06, 0.08]))
        gap = float(g.choice([0.01, 0.02, 0.03]))
        top = margin
        elems = []
        if g.random() < 0.7:  # header: one wide element
            hh = float(g.choice([0.06, 0.08, 0.1]))
            elems.append((0, 0.5, top + hh / 2, 1 - 2 * margin, hh))
            top += hh + gap
        heights = g.choice([0.06, 0.1, 0.14, 0.2], size=rows)
        heights = heights * min(1.0, (1 - margin - top - gap * rows) / heights.sum())
        cw = (1 - 2 * margin - gap * (cols - 1)) / cols
        y = top
        for r in range(rows):
            h = float(heights[r])
            band = min(3, int(4 * (y + h / 2)))
            span_row = g.random() < 0.15
            for c in range(1 if span_row else cols):
                if len(elems) >= max_elem or g.random() < 0.1:
                    continue
                w = (1 - 2 * margin) if span_row else cw
                xc = 0.5 if span_row else margin + c * (cw + gap) + cw / 2
                size_cls = 0 if w * h < 0.02 else (1 if w * h < 0.06 else 2)
                cat = 1 + (band * 3 + size_cls) * 2 + int(g.random() < 0.2)
                if g.random() < 0.05:
                    cat = int(g.integers(0, n_category))
                elems.append((cat % n_category, xc, y + h / 2, w, h))
            y += h + gap
        if not elems:
            elems.append((0, 0.5, 0.5, 0.5, 0.5))
        order = g.permutation(len(elems)) if g.random() < 0.5 else np.arange(len(elems))
        for k, j in enumerate(order[:max_elem]):
            cat, xc, yc, w, h = elems[j]
            label[i, k] = cat
            bbox[i, k] = (xc, yc, w, h)
            mask[i, k] = True
    return torch.from_numpy(label), torch.from_numpy(bbox).clamp(0.0, 1.0), torch.from_numpy(mask)


def fit(steps: int, batch: int, n_layout: int, seed: int, log_every: int = 25, budget_s: float = 3300.0):
    from trainer.models.base_model import BaseModel

    m, tok = rh.build_reference_model("rico25", seed=seed)   # the reference's own init (base_model.py:108-116)
    spec = SP.SPECS["rico25"]
    label, bbox, mask = structured_layouts(n_layout, spec.n_category, seed + 1)
    seq = tok.encode({"label": label, "bbox": bbox, "mask": mask})["seq"]
    assert seq.shape == (n_layout, spec.seq_len) and int(seq.max()) < spec.n_class
    no_decay = [f"transformer.pos_emb.{n}" for n in m.transformer.pos_emb.no_decay_param_names]
    groups = BaseModel.optim_groups(m, weight_decay=0.1, additional_no_decay=no_decay)
    opt = torch.optim.AdamW(groups, lr=5e-4, betas=(0.9, 0.98))
    m.train()
    g = torch.Generator().manual_seed(seed + 2)
    torch.manual_seed(seed + 3)   # (sample_time / q_sample draw from the global generator)
    curve, t0 = [], time.time()
    done = 0
    for it in range(steps):
        idx = torch.randint(0, n_layout, (batch,), generator=g)
        opt.zero_grad()
        _, losses = m(seq[idx])
        loss = sum(v.mean() for v in losses.values())     # LayoutDM.forward: mean per loss, main.py: sum
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        done = it + 1
        if it % log_every == 0 or it == steps - 1:
            curve.append((it, float(loss.item()), float(losses["kl_loss"].mean().item())))
            print(f"step {it:5d}  loss {loss.item():9.4f}  kl {losses['kl_loss'].mean().item():9.4f}  {time.time() - t0:6.0f} s", flush=True)
        if time.time() - t0 > budget_s:
            print(f"time budget reached after {done} steps", flush=True)
            break
    m.eval()
    return m, tok, spec, np.array(curve, np.float64), done, seq


def attention_score_max(m, tokens, tt):
    """max |q . k / sqrt(dh)| over layers / heads / positions of one forward (hook on the MHA modules' inputs)."""
    import math

    best = [0.0]
    hooks = []

    def hook(mod, args, kwargs):
        x = args[0]
        E = mod.embed_dim
        qkv = torch.nn.functional.linear(x, mod.in_proj_weight, mod.in_proj_bias)
        q, k = qkv[..., :E], qkv[..., E:2 * E]
        B, S, _ = q.shape
        H = mod.num_heads
        q = q.view(B, S, H, E // H).transpose(1, 2)
        k = k.view(B, S, H, E // H).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) / math.sqrt(E // H)
        best[0] = max(best[0], float(s.abs().max()))

    for mod in m.modules():
        if isinstance(mod, torch.nn.MultiheadAttention):
            hooks.append(mod.register_forward_pre_hook(hook, with_kwargs=True))
    with torch.no_grad():
        m.transformer(tokens, timestep=tt)
    for h in hooks:
        h.remove()
    return best[0]


def goldens(m, spec, seq, B=2):
    """make_golden.trained_like_cases' content for ONE checkpoint; the teacher-forced states are real (noised) training
    layouts, not uniform random tokens: q_sample of the reference at the stated timestep."""
    from trainer.models.categorical_diffusion.util import index_to_log_onehot

    from . import make_golden as MG

    out = {}
    ts = [90, 50, 5]
    out["ts"] = np.array(ts, np.int32)
    g = torch.Generator().manual_seed(77)
    floor, smax, lmax = 0.0, 0.0, 0.0
    for t in ts:
        tokens = MG.random_valid_tokens(spec, B, t / (spec.n_step - 1), g)
        # half of the states: a training layout with the step's share of [MASK]s (what the sampler actually visits)
        x0 = seq[torch.randint(0, seq.shape[0], (1,), generator=g)][0].clone()
        x0[torch.rand(spec.seq_len, generator=g) < t / (spec.n_step - 1)] = spec.mask_id
        tokens[0] = x0
        tt = torch.full((B,), t, dtype=torch.long)
        with torch.no_grad():
            logits = m.transformer(tokens, timestep=tt)["logits"]
            lz = index_to_log_onehot(tokens, spec.n_class)
            post = m.q_posterior(m.predict_start(lz, tt), lz, tt)
            m.double()
            logits64 = m.transformer(tokens, timestep=tt)["logits"]
            m.float()
        floor = max(floor, ((logits.double() - logits64).abs().max() / logits64.abs().max()).item())
        lmax = max(lmax, float(logits.abs().max()))
        smax = max(smax, attention_score_max(m, tokens, tt))
        out[f"tokens_{t}"] = tokens.numpy().astype(np.int16)
        out[f"logits_{t}"] = logits.numpy()
        out[f"post_{t}"] = post.numpy()
    out["f32_noise_floor"] = np.float64(floor)
    out["max_abs_logit"] = np.float64(lmax)
    out["max_abs_attention_score"] = np.float64(smax)
    tr = MG.trajectory(m, spec, B, rh.sampling_cfg("random"), None, seed=11)
    for k, v in tr.items():
        out[k] = v
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--layouts", type=int, default=8192)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--threads", type=int, default=6)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    rh.install_stubs()
    m, tok, spec, curve, done, seq = fit(a.steps, a.batch, a.layouts, a.seed)
    sd = {k: v.detach().cpu().numpy().astype(np.float32) for k, v in rh.state_dict_layoutdm_keys(m).items()}
    os.makedirs(FIT_DIR, exist_ok=True)
    np.savez(WEIGHTS, **sd)
    with open(WEIGHTS, "rb") as fh:
        digest = hashlib.sha256(fh.read()).hexdigest()
    out = goldens(m, spec, seq)
    out["weights_sha256"] = np.array(digest)
    out["loss_curve"] = curve                      # (step, total loss, kl loss)
    out["steps_done"] = np.int64(done)
    out["train_args"] = np.array([a.steps, a.batch, a.layouts, a.seed], np.int64)
    names, stats = [], []
    for k, v in sorted(sd.items()):
        if v.ndim >= 1 and v.size > 1 and "log_" not in k and "Lt_" not in k:
            names.append(k)
            stats.append((float(v.std()), float(np.abs(v).max()), float(v.mean())))
    out["tensor_names"] = np.array(names)
    out["tensor_stats"] = np.array(stats, np.float64)   # sigma, max |w|, mean
    np.savez_compressed(GOLDEN, **out)
    print(f"wrote {WEIGHTS} ({os.path.getsize(WEIGHTS) / 1e6:.1f} MB, sha256 {digest[:16]}...) and {GOLDEN} "
          f"({os.path.getsize(GOLDEN) / 1e3:.0f} KB); max |logit| {out['max_abs_logit']:.2f}, max |attention score| "
          f"{out['max_abs_attention_score']:.2f}, f32 noise floor {out['f32_noise_floor']:.2e}")


if __name__ == "__main__":
    main()
