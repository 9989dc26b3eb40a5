"""Dev tool (CPU, build container: needs /root/reference): is the reference's OWN cond=relation decision a function of its
float32 summation order?  (VERDICT r4 next #8.)

On every state of the reference's T = 200 cond=relation trajectory (tests/golden/rico25_config5_T200.npz) the reference's
float32 posterior log-probabilities (strong mask applied, base.py:243-251) go through its own logit adjustment
(logit_adjustment.update: 3 autograd SGD steps, lr 3e6) twice: in float32, as it runs, and with the SAME numbers cast to
float64.  Then [PAD] disable + argmax as base.py:272-291.  Tokens on which the two differ are tokens whose float32 value is
decided by rounding inside the update (a hinge of clg/const.py switching on the last bits of an expected box): no
implementation with another summation order (another BLAS / vector width / GPU) can be expected to reproduce them.

    python tools/relation_order_dependence.py > profiles/r05_relation_order_dependence.txt
"""
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import make_golden as MG, ref_harness as rh, spec as SP, synth  # noqa: E402
import dataclasses  # noqa: E402

torch.set_num_threads(8)
spec = dataclasses.replace(SP.SPECS["rico25"], name="rico25_t200", n_step=200)
m, tok = rh.build_reference_model("rico25", seed=0, n_step=200)
ssd = synth.synth_state_dict(spec, seed=MG.WEIGHT_SEED, perturb=True, prefix="")
m.load_state_dict({k: torch.from_numpy(v) for k, v in ssd.items()})
from einops import rearrange, repeat  # noqa: E402
from trainer.models.categorical_diffusion.logit_adjustment import update  # noqa: E402
from trainer.models.categorical_diffusion.util import LOG_EPS, index_to_log_onehot  # noqa: E402

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "rico25_config5_T200.npz"))
sub = {k[4:]: g[k] for k in g.files if k.startswith("rel_")}
graph = rh.GraphBatch(*(torch.from_numpy(sub[k]) for k in ("y", "edge_index", "edge_attr", "batch")))
seq, mask = torch.from_numpy(sub["cond_seq"].astype(np.int64)), torch.from_numpy(sub["cond_mask"])
cond = {"seq": seq, "mask": mask, "type": "relation", "batch_w_canvas": graph}
cfg = rh.sampling_cfg("deterministic", num_timesteps=200, relation_lambda=3e6, relation_mode="average", relation_tau=1.0,
                      relation_num_update=3)
B, S = seq.shape
C = spec.n_class
pad_mask = (repeat(torch.arange(S), "s -> b s", b=B) % 5 != 0) & (seq != spec.pad_id)
pad_mask = repeat(pad_mask, "b s -> b c s", c=C) & (rearrange(torch.arange(C), "c -> 1 c 1") == spec.pad_id)


def finish(lp):
    lp = lp.clone()
    lp[pad_mask] = LOG_EPS
    top2 = lp.topk(2, dim=1).values
    return lp.argmax(1), (top2[:, 0] - top2[:, 1])


n_diff = n_tok = n_ref_diff = 0
print("# state i, t | tokens where the reference's float32 update and the same update in float64 decide differently")
for i, t in enumerate(sub["steps"]):
    t = int(t)
    before = torch.from_numpy(sub["states_before"][i].astype(np.int64))
    with torch.no_grad():
        lz = index_to_log_onehot(before, C)
        tt = torch.full((B,), t, dtype=torch.long)
        post = m.q_posterior(log_x_start=m.predict_start(lz, tt), log_x_t=lz, t=tt)
        post = torch.where(rearrange(mask, "b s -> b 1 s"), index_to_log_onehot(seq, C), post)
    lp32 = update(t=t, cond=copy.copy(cond), model_log_prob=post.clone(), tokenizer=tok, sampling_cfg=cfg)
    lp64 = update(t=t, cond=copy.copy(cond), model_log_prob=post.double(), tokenizer=tok, sampling_cfg=cfg)
    tok32, mar32 = finish(lp32)
    tok64, mar64 = finish(lp64)
    ref = torch.from_numpy(sub["greedy_next"][i].astype(np.int64))
    n_ref_diff += int((tok32 != ref).sum())          # (sanity: this script's float32 path IS the fixture's)
    d = tok32 != tok64
    n_diff += int(d.sum())
    n_tok += tok32.numel()
    for b, s in d.nonzero().tolist():
        print(f"state {i:3d} t={t:3d} layout {b} position {s:3d} (element {s // 5}, attribute {s % 5}): float32 -> {int(tok32[b, s])} "
              f"(margin {float(mar32[b, s]):.3e}), float64 -> {int(tok64[b, s])} (margin {float(mar64[b, s]):.3e}); "
              f"max |logp32 - logp64| of the row {float((lp32[b, :, s].double() - lp64[b, :, s]).abs().max()):.3e}")
print(f"# {n_diff} of {n_tok} greedy tokens of the reference's own trajectory depend on the precision / order of its update "
      f"(float32 vs float64 on identical inputs); float32 path differs from the committed fixture on {n_ref_diff} tokens")
