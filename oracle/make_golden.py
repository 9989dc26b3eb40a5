"""TEST INFRASTRUCTURE ONLY — generates tests/golden/*.npz by running the REAL reference
(/root/reference, imported through oracle/ref_harness.py) on deterministic synthetic
checkpoints (oracle/synth.py).  Run in the build container only:

    python -m oracle.make_golden

The reference ships no golden vectors / tests of its own, so these
reference-produced fixtures are what pins oracle/restatement.py and, through it, the
HIP path.  Fixtures are small (B<=4) so they can live in git.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from . import ref_harness as rh
from . import spec as SP
from . import synth

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
WEIGHT_SEED = 1


def load_synth(m, spec, seed=WEIGHT_SEED, perturb=True):
    ssd = synth.synth_state_dict(spec, seed=seed, perturb=perturb, prefix="")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in ssd.items()})
    return ssd


def random_valid_tokens(spec, B, mask_frac, g):
    tokens = torch.empty(B, spec.seq_len, dtype=torch.long)
    for a in range(spec.n_attr):
        ids = torch.as_tensor(spec.full_ids(a))
        # any live class except MASK (pad allowed)
        tokens[:, a::spec.n_attr] = ids[torch.randint(0, len(ids) - 1, (B, spec.max_elem), generator=g)]
    mk = torch.rand(B, spec.seq_len, generator=g) < mask_frac
    tokens[mk] = spec.mask_id
    return tokens


def step_cases(m, spec, ts, B=2):
    """Teacher-forced single-step pieces from the reference: transformer logits
    (nn_lib.py:191-237), predict_start (base.py:127-146), q_posterior
    (constrained.py:135-206)."""
    from trainer.models.categorical_diffusion.util import index_to_log_onehot

    g = torch.Generator().manual_seed(123)
    out = {"ts": np.array(ts, np.int32)}
    for t in ts:
        tokens = random_valid_tokens(spec, B, t / (spec.n_step - 1), g)
        tt = torch.full((B,), t, dtype=torch.long)
        with torch.no_grad():
            logits = m.transformer(tokens, timestep=tt)["logits"]
            lz = index_to_log_onehot(tokens, spec.n_class)
            x0 = m.predict_start(lz, tt)
            post = m.q_posterior(x0, lz, tt)
        out[f"tokens_{t}"] = tokens.numpy().astype(np.int16)
        out[f"logits_{t}"] = logits.numpy()
        out[f"post_{t}"] = post.numpy()
        if t == ts[0]:
            out[f"x0_{t}"] = x0.numpy()
    return out


def trajectory(m, spec, B, cfg, cond=None, seed=0):
    """Full reference sample() with intermediate results (base.py:293-371), plus for
    every visited state the reference's own greedy (argmax) next tokens — i.e. a
    teacher-forced per-step known-answer table covering every t."""
    from trainer.models.categorical_diffusion.util import index_to_log_onehot
    import copy

    torch.manual_seed(seed)
    c = copy.deepcopy(cond)
    inter = m.sample(batch_size=B, cond=c, sampling_cfg=cfg, get_intermediate_results=True)
    states = torch.stack(inter)  # (T, B, S): state AFTER each step
    T = states.shape[0]
    if cond is None:
        start = torch.full((B, spec.seq_len), spec.mask_id, dtype=torch.long)
    else:
        start = cond["seq"].clone()
    prev_states = torch.cat([start[None], states[:-1]])  # state BEFORE each step
    det = rh.sampling_cfg("deterministic")
    for k in cfg:
        if k.startswith("refine") or k.startswith("relation") or k == "time_difference":
            det[k] = cfg[k]
    greedy, margin = [], []
    steps = [int(i * spec.n_step / T) for i in range(T - 1, -1, -1)]
    # the log-probabilities the reference hands to sample() (base.py:286-287), captured at the call site: their
    # top-2 gap says how far a reduced-precision implementation may be off before ITS argmax may legitimately differ
    import trainer.models.categorical_diffusion.base as ref_base

    orig_sample = ref_base.sample
    seen = {}

    def spy(logits, sampling_cfg):
        seen["logp"] = logits.detach().clone()
        return orig_sample(logits, sampling_cfg)

    ref_base.sample = spy
    try:
        for i, t in enumerate(steps):
            lz = index_to_log_onehot(prev_states[i], spec.n_class)
            tt = torch.full((B,), t, dtype=torch.long)
            cc = copy.deepcopy(c)  # already holds weak_* keys for refinement
            nz = m._sample_single_step(lz, tt, skip_step=0, sampling_cfg=det, cond=cc)
            greedy.append(nz.argmax(1))
            top2 = seen["logp"].topk(2, dim=1).values
            margin.append(top2[:, 0] - top2[:, 1])
    finally:
        ref_base.sample = orig_sample
    return {
        "steps": np.array(steps, np.int32),
        "states_before": prev_states.numpy().astype(np.int16),
        "states_after": states.numpy().astype(np.int16),
        "greedy_next": torch.stack(greedy).numpy().astype(np.int16),
        "greedy_margin": torch.stack(margin).numpy().astype(np.float32),
    }


def capture_probs(m, spec, tokens, t, cfg, cond=None):
    """Probabilities the reference feeds torch.multinomial (helpers/sampling.py:119-127)."""
    from trainer.models.categorical_diffusion.util import index_to_log_onehot
    import copy

    captured = {}
    orig = torch.multinomial

    def fake(probs, num_samples, *a, **k):
        captured["p"] = probs.detach().clone()
        return probs.argmax(dim=-1, keepdim=True)

    torch.multinomial = fake
    try:
        lz = index_to_log_onehot(tokens, spec.n_class)
        tt = torch.full((tokens.shape[0],), t, dtype=torch.long)
        m._sample_single_step(lz, tt, skip_step=0, sampling_cfg=cfg, cond=copy.deepcopy(cond))
    finally:
        torch.multinomial = orig
    B, S = tokens.shape
    return captured["p"].view(B, S, spec.n_class).permute(0, 2, 1).contiguous()  # (B,C,S)


def cond_variant_cases(m, spec, B=3):
    """The cond types and sampler options that r01 only checked against the oracle (VERDICT a12 / a13 / f4), now from
    the reference itself: cond=cwh and cond=partial built as helpers/task.py:61-110 builds them (on synthetic layouts),
    a time_difference run (base.py:218-226), and the top-k / temperature probabilities at torch.multinomial."""
    out = {}
    g = torch.Generator().manual_seed(21)
    c = synth.synth_cond_c(spec, B, seed=13)
    A = spec.n_attr
    full = torch.from_numpy(c["seq"]).clone()          # a complete ("gt") sequence: every attribute of valid elements
    valid = full != spec.pad_id
    for a in range(1, A):
        ids = torch.as_tensor(spec.full_ids(a))
        rnd = ids[torch.randint(0, spec.n_bin, (B, spec.max_elem), generator=g)]
        full[:, a::A] = torch.where(valid[:, a::A], rnd, full[:, a::A])
    elem_valid = valid[:, 0::A]
    attr = torch.arange(spec.seq_len).view(1, -1) % A
    # cwh (task.py:91-107): keep c, w, h of valid elements, everything else [MASK]; padded slots [PAD] and fixed
    keep = (attr == 0) | (attr == 3) | (attr == 4)
    seq = torch.where(keep, full, torch.full_like(full, spec.mask_id))
    seq = torch.where(valid, seq, torch.full_like(full, spec.pad_id))
    mask = (valid & keep) | ~valid
    cond = {"seq": seq.clone(), "mask": mask.clone(), "type": "cwh"}
    tr = trajectory(m, spec, B, rh.sampling_cfg("random"), cond, seed=4)
    for k, v in tr.items():
        out["cwh_" + k] = v
    out["cwh_cond_seq"] = seq.numpy().astype(np.int16)
    out["cwh_cond_mask"] = mask.numpy()
    # partial (task.py:61-88, no bos): a random subset of the valid elements is kept whole, the rest is [MASK]
    # (also the padded slots: the number of elements is NOT given in this setting)
    keep_e = (torch.rand(B, spec.max_elem, generator=g) < 0.4) & elem_valid
    keep_e[:, 0] = True  # at least one element (task.py:68-72)
    keep = keep_e.repeat_interleave(A, dim=1)
    seq = torch.where(keep, full, torch.full_like(full, spec.mask_id))
    cond = {"seq": seq.clone(), "mask": keep.clone(), "type": "partial"}
    tr = trajectory(m, spec, B, rh.sampling_cfg("random"), cond, seed=5)
    for k, v in tr.items():
        out["partial_" + k] = v
    out["partial_cond_seq"] = seq.numpy().astype(np.int16)
    out["partial_cond_mask"] = keep.numpy()
    # time_difference (base.py:218-226): the posterior is evaluated at t - int(T * time_difference)
    tr = trajectory(m, spec, B, rh.sampling_cfg("random", time_difference=0.15), None, seed=6)
    for k, v in tr.items():
        out["td_" + k] = v
    # sampler options at torch.multinomial (sampling.py:81-127) on three states of the cwh trajectory
    cw = {"seq": torch.from_numpy(out["cwh_cond_seq"].astype(np.int64)), "mask": torch.from_numpy(out["cwh_cond_mask"]),
          "type": "cwh"}
    for name, cfg in (("top_k", rh.sampling_cfg("top_k", top_k=5, temperature=0.7)),
                      ("temp", rh.sampling_cfg("random", temperature=0.6)),
                      ("top_p_temp", rh.sampling_cfg("top_p", top_p=0.8, temperature=1.3))):
        for i in (0, 60, 99):
            toks = torch.from_numpy(out["cwh_states_before"][i].astype(np.int64))
            out[f"probs_{name}_{i}"] = capture_probs(m, spec, toks, int(out["cwh_steps"][i]), cfg, cw).numpy()
    return out


def decode_cases(tok, spec, B=16, seed=11):
    """ids -> {bbox,label,mask} through the reference's own LayoutSequenceTokenizer.decode: valid layouts plus
    every corruption the range filter sees (pad / mask tokens, labels in bbox slots, bbox ids in the label slot,
    bbox ids of ANOTHER attribute's sub-vocabulary, which pass the filter and are clamped by BboxTokenizer.decode).
    Both bbox_quantization=linear and the cluster-centre path (kmeans/percentile code path driven with synthetic
    float64 centres through the reference's DummyClusteringModel, bbox_tokenizer.py:23-25)."""
    from trainer.helpers.bbox_tokenizer import DummyClusteringModel

    g = torch.Generator().manual_seed(seed)
    tokens = random_valid_tokens(spec, B, 0.0, g)
    noise = torch.randint(0, spec.n_class, tokens.shape, generator=g)
    corrupt = torch.rand(tokens.shape, generator=g) < 0.15
    corrupt[0] = False                     # one fully valid layout
    tokens = torch.where(corrupt, noise, tokens)
    tokens[1] = spec.mask_id               # all [MASK]
    tokens[2] = spec.pad_id                # all [PAD]
    out = {"tokens": tokens.numpy()}
    dec = tok.decode(tokens.clone())
    out["linear_bbox"], out["linear_label"], out["linear_mask"] = (dec["bbox"].numpy(), dec["label"].numpy(),
                                                                   dec["mask"].numpy())
    bbt = tok.bbox_tokenizer
    saved = (bbt._bbox_quantization, bbt._clustering_models)
    rng = np.random.default_rng(seed)
    centres = np.sort(rng.uniform(-0.05, 1.05, size=(4, spec.n_bin)), axis=1)   # some outside [0,1]: clamp path
    bbt._bbox_quantization = "kmeans"
    bbt._clustering_models = {f"{k}-{spec.n_bin}": DummyClusteringModel(centres[i].reshape(-1, 1))
                              for i, k in enumerate(["x", "y", "w", "h"])}
    try:
        dec = tok.decode(tokens.clone())
    finally:
        bbt._bbox_quantization, bbt._clustering_models = saved
    out["centres"] = centres
    out["kmeans_bbox"], out["kmeans_label"], out["kmeans_mask"] = (dec["bbox"].numpy(), dec["label"].numpy(),
                                                                   dec["mask"].numpy())
    return out


def vanilla_cases(spec, B=2):
    """q_type=vanilla (models/layoutdm.py:20-23 -> categorical_diffusion/vanilla.py): teacher-forced step pieces on
    states whose tokens are NOT restricted to their attribute's sub-vocabulary (the single-vocabulary posterior
    accepts any class anywhere) and a stochastic reference trajectory with per-state greedy answers."""
    m, _ = rh.build_reference_model("rico25", seed=0, q_type="vanilla")  # (installs the import stubs)
    from trainer.models.categorical_diffusion.util import index_to_log_onehot

    ssd = synth.synth_state_dict(spec, seed=WEIGHT_SEED, perturb=True, prefix="", q_type="vanilla")
    sched_ref = {k: v.clone() for k, v in m.state_dict().items() if k.startswith("log_")}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in ssd.items()})
    out = {f"sched_{k}": v.numpy() for k, v in sched_ref.items()}  # the reference's own buffers (pin spec.py)
    g = torch.Generator().manual_seed(321)
    ts = [99, 60, 20, 1, 0]
    out["ts"] = np.array(ts, np.int32)
    for t in ts:
        tokens = torch.randint(0, spec.n_class - 1, (B, spec.seq_len), generator=g)
        tokens[torch.rand(B, spec.seq_len, generator=g) < t / (spec.n_step - 1)] = spec.mask_id
        tt = torch.full((B,), t, dtype=torch.long)
        with torch.no_grad():
            lz = index_to_log_onehot(tokens, spec.n_class)
            x0 = m.predict_start(lz, tt)
            post = m.q_posterior(x0, lz, tt)
        out[f"tokens_{t}"] = tokens.numpy().astype(np.int16)
        out[f"post_{t}"] = post.numpy()
    tr = trajectory(m, spec, B, rh.sampling_cfg("random"), None, seed=3)
    out.update({f"traj_{k}": v for k, v in tr.items()})
    return out


def relation_cases(spec, B=2, seed=7):
    """cond=relation: the reference's own logit_adjustment.update (SGD on the 14 relational-constraint losses of
    clg/const.py) applied to posterior-shaped log-probabilities, on graphs produced by the reference's own
    AddCanvasElement + AddRelationConstraints transforms (data/util.py:111-177) from random layouts."""
    _, tok = rh.build_reference_model("rico25", seed=0)          # (installs the import stubs)
    from trainer.data.util import AddCanvasElement, AddRelationConstraints
    from trainer.models.categorical_diffusion.logit_adjustment import update

    g = torch.Generator().manual_seed(seed)
    rel = AddRelationConstraints(seed=seed, edge_ratio=0.5)
    ys, eis, eas, bts, seqs = [], [], [], [], []
    off = 0
    for b in range(B):
        n = int(torch.randint(3, 9, (1,), generator=g))
        box = torch.rand(n, 4, generator=g) * torch.tensor([0.8, 0.8, 0.5, 0.5]) + torch.tensor([0.1, 0.1, 0.05, 0.05])
        lab = torch.randint(0, spec.n_category, (n,), generator=g)
        data = type("Data", (), {})()
        data.x, data.y, data.attr = box, lab, {"has_canvas_element": torch.tensor([False])}
        data = rel(AddCanvasElement()(data))
        ys.append(data.y)
        eis.append(data.edge_index.view(2, -1) + off)
        eas.append(data.edge_attr)
        bts.append(torch.full((n + 1,), b, dtype=torch.long))
        off += n + 1
        seq = torch.full((spec.seq_len,), spec.pad_id, dtype=torch.long)
        seq[: n * spec.n_attr] = spec.mask_id
        seq[0: n * spec.n_attr: spec.n_attr] = lab                      # categories given, the rest [MASK]
        seqs.append(seq)
    graph = rh.GraphBatch(torch.cat(ys), torch.cat(eis, dim=1), torch.cat(eas), torch.cat(bts))
    cond = {"seq": torch.stack(seqs), "batch_w_canvas": graph, "type": "relation"}
    logp = torch.log_softmax(2.0 * torch.randn(B, spec.n_class, spec.seq_len, generator=g), dim=1).clamp(-70, 0)
    cfg = rh.sampling_cfg("random", relation_lambda=3e6, relation_mode="average", relation_tau=1.0,
                          relation_num_update=3)
    bt = tok.bbox_tokenizer
    centres = np.stack([np.asarray(bt.clustering_models[f"{k}-{spec.n_bin}"].cluster_centers_, np.float64).reshape(-1)
                        for k in ("x", "y", "w", "h")])
    canvas_ids = bt.encode(torch.tensor([[[0.5, 0.5, 1.0, 1.0]]])).long().view(-1)    # logit_adjustment.py:38
    canvas_bins = (canvas_ids - torch.arange(4) * spec.n_bin).numpy()
    out = {"logp_in": logp.numpy(), "cond_seq": cond["seq"].numpy().astype(np.int16), "y": graph.y.numpy(),
           "edge_index": graph.edge_index.numpy(), "edge_attr": graph.edge_attr.numpy(), "batch": graph.batch.numpy(),
           "centres": centres, "canvas_bins": canvas_bins.astype(np.int32), "lr": np.float64(3e6),
           "num_update": np.int32(3)}
    for t in (50, 5):
        new = update(t=t, cond=cond, model_log_prob=logp.clone(), tokenizer=tok, sampling_cfg=cfg)
        out[f"logp_out_t{t}"] = new.numpy()
    return out


def _relation_cond(spec, B, seed, edge_ratio=0.5):
    """cond=relation inputs as the reference builds them: random layouts -> AddCanvasElement + AddRelationConstraints
    (data/util.py:111-177) -> graph batch; seq / mask as helpers/task.py:94-114 (categories given, [PAD] beyond)."""
    from trainer.data.util import AddCanvasElement, AddRelationConstraints

    g = torch.Generator().manual_seed(seed)
    rel = AddRelationConstraints(seed=seed, edge_ratio=edge_ratio)
    ys, eis, eas, bts, seqs, off = [], [], [], [], [], 0
    for b in range(B):
        n = int(torch.randint(3, 9, (1,), generator=g))
        box = torch.rand(n, 4, generator=g) * torch.tensor([0.8, 0.8, 0.5, 0.5]) + torch.tensor([0.1, 0.1, 0.05, 0.05])
        lab = torch.randint(0, spec.n_category, (n,), generator=g)
        data = type("Data", (), {})()
        data.x, data.y, data.attr = box, lab, {"has_canvas_element": torch.tensor([False])}
        data = rel(AddCanvasElement()(data))
        ys.append(data.y)
        eis.append(data.edge_index.view(2, -1) + off)
        eas.append(data.edge_attr)
        bts.append(torch.full((n + 1,), b, dtype=torch.long))
        off += n + 1
        seq = torch.full((spec.seq_len,), spec.pad_id, dtype=torch.long)
        seq[: n * spec.n_attr] = spec.mask_id
        seq[0: n * spec.n_attr: spec.n_attr] = lab
        seqs.append(seq)
    graph = rh.GraphBatch(torch.cat(ys), torch.cat(eis, dim=1), torch.cat(eas), torch.cat(bts))
    seq = torch.stack(seqs)
    return graph, seq, seq != spec.mask_id


GETCOND_TYPES = ("c", "cwh", "partial", "refinement", "relation")


def getcond_sampling_cfg(ctype, name="random"):
    """sampling_cfg as test.py:107-118 + aggregate_sampling_settings (base_model.py:124-150) leave it for `ctype`, with the
    reference's TestConfig defaults (hydra_configs.py:34-47)."""
    kw = {}
    if ctype == "refinement":
        kw = dict(refine_mode="uniform", refine_offset_ratio=0.1, refine_lambda=3.0)
    if ctype == "relation":
        kw = dict(relation_lambda=3e6, relation_mode="average", relation_tau=1.0, relation_num_update=3)
    return rh.sampling_cfg(name, **kw)


def getcond_cases(spec, B=2):
    """the cond dicts are PRODUCED BY THE REFERENCE'S OWN get_cond (helpers/task.py:27-151) from a
    collated batch (ref_harness.synth_layout_batch: for cond=relation through its own AddCanvasElement +
    AddRelationConstraints transforms), for every cond type of the `test` entry point; each is then sampled by the
    reference — the greedy sample() of test.py:195-200 and a stochastic trajectory with the greedy answer and top-2 margin
    at every visited state.  The fixture stores the dicts field by field; the GPU test hands them, unchanged, to
    layout_dm_amd.layoutdm.LayoutDM.sample."""
    import copy
    import random as pyrandom

    m, tok = rh.build_reference_model(spec.name, seed=0)
    load_synth(m, spec)
    from trainer.helpers.task import get_cond

    out = {"types": np.array(GETCOND_TYPES)}
    for i, ctype in enumerate(GETCOND_TYPES):
        rel = ctype == "relation"
        batch = rh.synth_layout_batch(spec.n_category, B, seed=40 + i, relation=rel, transform_seed=0,
                                      n_lo=8 if rel else 3, n_hi=14 if rel else 12)
        pyrandom.seed(50 + i)           # set_seed (helpers/util.py:10-13): partial draws from `random` and torch
        torch.manual_seed(50 + i)
        cond = get_cond(batch=batch, tokenizer=tok, cond_type=ctype, model_type="LayoutDM")
        p = ctype + "_"
        out[p + "x"], out[p + "y"], out[p + "batch"] = batch.x.numpy(), batch.y.numpy(), batch.batch.numpy()
        out[p + "cond_seq"] = cond["seq"].numpy().astype(np.int16)
        out[p + "cond_mask"] = cond["mask"].numpy()
        out[p + "cond_keys"] = np.array(sorted(cond.keys()))
        if "num_element" in cond:
            out[p + "num_element"] = cond["num_element"].numpy()
        if "seq_orig" in cond:
            out[p + "seq_orig"] = cond["seq_orig"].numpy().astype(np.int16)
        if ctype == "relation":
            gb = cond["batch_w_canvas"]
            assert gb is batch
            out[p + "edge_index"], out[p + "edge_attr"] = gb.edge_index.numpy(), gb.edge_attr.numpy()
            bt = tok.bbox_tokenizer   # what logit_adjustment.py:30-41 reads from the tokenizer
            out[p + "centres"] = np.stack([np.asarray(bt.clustering_models[f"{k}-{spec.n_bin}"].cluster_centers_,
                                                      np.float64).reshape(-1) for k in ("x", "y", "w", "h")])
            canvas_ids = bt.encode(torch.tensor([[[0.5, 0.5, 1.0, 1.0]]])).long().view(-1)
            out[p + "canvas_bins"] = (canvas_ids - torch.arange(4) * spec.n_bin).numpy().astype(np.int32)
        if ctype == "refinement":
            from trainer.helpers.task import _index_to_smoothed_log_onehot

            table = _index_to_smoothed_log_onehot(torch.arange(spec.n_class)[None], tok, mode="uniform",
                                                  offset_ratio=0.1)[0].T.contiguous() * 3.0
            out[p + "weak_table"] = table.numpy()   # [token, class], lambda applied (helpers/task.py:154-224)
        torch.manual_seed(60 + i)
        greedy = m.sample(batch_size=B, cond=copy.deepcopy(cond), sampling_cfg=getcond_sampling_cfg(ctype, "deterministic"))
        out[p + "greedy_tokens"] = greedy.numpy().astype(np.int16)
        tr = trajectory(m, spec, B, getcond_sampling_cfg(ctype, "random"), cond, seed=70 + i)
        for k, v in tr.items():
            out[p + "traj_" + k] = v
        if ctype == "relation":
            out[p + "traj_order_dependent"] = relation_order_dependent_tokens(m, tok, spec, tr, cond,
                                                                              getcond_sampling_cfg(ctype, "random"))
    return out


def trained_like_cases(spec, B=2, points=None):
    """VERDICT r3 next #1a: the reference itself on weight distributions other than its init (oracle/synth.py
    TRAINED_LIKE): teacher-forced denoiser logits / posterior at three timesteps, the reference's own float32 noise floor
    at each (its float32 forward against its own float64 forward: what "bit-exact greedy tokens" and "logits <= 2e-5" can
    mean at that point), the largest attention score, and a stochastic trajectory with the reference's greedy next
    tokens and top-2 margins at every step."""
    m, _ = rh.build_reference_model(spec.name, seed=0)               # (installs the import stubs)
    from trainer.models.categorical_diffusion.util import index_to_log_onehot

    points = list(points or synth.TRAINED_LIKE)
    out = {"points": np.array(points)}
    g = torch.Generator().manual_seed(77)
    ts = [90, 50, 5]
    out["ts"] = np.array(ts, np.int32)
    for point in points:
        ssd = synth.trained_like_state_dict(spec, point, seed=2, prefix="")
        m.load_state_dict({k: torch.from_numpy(v) for k, v in ssd.items()})
        floor, smax = 0.0, 0.0
        for t in ts:
            tokens = random_valid_tokens(spec, B, t / (spec.n_step - 1), g)
            tt = torch.full((B,), t, dtype=torch.long)
            with torch.no_grad():
                logits = m.transformer(tokens, timestep=tt)["logits"]
                lz = index_to_log_onehot(tokens, spec.n_class)
                post = m.q_posterior(m.predict_start(lz, tt), lz, tt)
                m.double()
                logits64 = m.transformer(tokens, timestep=tt)["logits"]
                m.float()
            floor = max(floor, ((logits.double() - logits64).abs().max() / logits64.abs().max()).item())
            out[f"{point}_tokens_{t}"] = tokens.numpy().astype(np.int16)
            out[f"{point}_logits_{t}"] = logits.numpy()
            out[f"{point}_post_{t}"] = post.numpy()
        # m.double()/m.float() round-trips the float32 parameters exactly; reload anyway so the trajectory below starts
        # from the checkpoint as loaded
        m.load_state_dict({k: torch.from_numpy(v) for k, v in ssd.items()})
        out[f"{point}_f32_noise_floor"] = np.float64(floor)
        tr = trajectory(m, spec, B, rh.sampling_cfg("random"), None, seed=11)
        for k, v in tr.items():
            out[f"{point}_{k}"] = v
    return out


def relation_order_dependent_tokens(m, tok, spec, tr, cond, cfg):
    """Annotation of a cond=relation trajectory fixture: the greedy tokens whose value in the reference's
    float32 run is decided by ROUNDING inside its logit adjustment.  On every state the reference's float32 posterior (strong
    mask applied, base.py:243-251) goes through logit_adjustment.update twice — in float32 as it runs, and with the same
    numbers cast to float64 — then [PAD] disable + argmax (base.py:272-291).  Rows: [state, layout, position, float32 token,
    float64 token] wherever the two differ.  An implementation with another summation order can agree with either."""
    import copy

    from einops import rearrange, repeat
    from trainer.models.categorical_diffusion.logit_adjustment import update
    from trainer.models.categorical_diffusion.util import LOG_EPS, index_to_log_onehot

    seq, mask = cond["seq"], cond["mask"]
    B, S = seq.shape
    C = spec.n_class
    det = rh.sampling_cfg("deterministic")
    for k in cfg:
        if k.startswith("relation") or k == "num_timesteps":
            det[k] = cfg[k]
    pad_mask = (repeat(torch.arange(S), "s -> b s", b=B) % spec.n_attr != 0) & (seq != spec.pad_id)
    pad_mask = repeat(pad_mask, "b s -> b c s", c=C) & (rearrange(torch.arange(C), "c -> 1 c 1") == spec.pad_id)
    rows = []
    for i, t in enumerate(tr["steps"]):
        t = int(t)
        before = torch.from_numpy(tr["states_before"][i].astype(np.int64))
        with torch.no_grad():
            lz = index_to_log_onehot(before, C)
            tt = torch.full((B,), t, dtype=torch.long)
            post = m.q_posterior(log_x_start=m.predict_start(lz, tt), log_x_t=lz, t=tt)
            post = torch.where(rearrange(mask, "b s -> b 1 s"), index_to_log_onehot(seq, C), post)
        toks = []
        for lp in (post.clone(), post.double()):
            lp = update(t=t, cond=copy.copy(cond), model_log_prob=lp, tokenizer=tok, sampling_cfg=det).clone()
            lp[pad_mask] = LOG_EPS
            toks.append(lp.argmax(1))
        assert np.array_equal(toks[0].numpy(), tr["greedy_next"][i])      # the float32 path here IS the fixture's
        for b, s_ in (toks[0] != toks[1]).nonzero().tolist():
            rows.append([i, b, s_, int(toks[0][b, s_]), int(toks[1][b, s_])])
    return np.array(rows, np.int32).reshape(-1, 5)


def config5_cases(B=2):
    """BASELINE config 5's shape (VERDICT r3 next #1b): a T = 200 model (schedule buffers and AdaLN tables of 200
    timesteps; base.py:310-311 rejects num_timesteps > the model's) sampled with cond=refinement and with cond=relation —
    the reference's own sample() on reference-built inputs (helpers/task.py:154-224 prior, data/util.py:111-177 graphs),
    all 200 steps, with the reference's greedy next tokens and top-2 margins at every visited state."""
    import dataclasses

    spec = dataclasses.replace(SP.SPECS["rico25"], name="rico25_t200", n_step=200)
    m, tok = rh.build_reference_model("rico25", seed=0, n_step=200)
    ssd = synth.synth_state_dict(spec, seed=WEIGHT_SEED, perturb=True, prefix="")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in ssd.items()})
    out = {}
    # refinement (helpers/task.py:126-138, LayoutDM branch)
    c = synth.synth_cond_c(spec, B, seed=15)
    g = torch.Generator().manual_seed(19)
    seq_orig = torch.from_numpy(c["seq"]).clone()
    for a in range(1, spec.n_attr):
        ids = torch.as_tensor(spec.full_ids(a))
        valid = torch.from_numpy(c["seq"][:, a::spec.n_attr] == spec.mask_id)
        rnd = ids[torch.randint(0, spec.n_bin, (B, spec.max_elem), generator=g)]
        seq_orig[:, a::spec.n_attr] = torch.where(valid, rnd, seq_orig[:, a::spec.n_attr])
    cond = {"seq": torch.from_numpy(c["seq"]), "mask": torch.from_numpy(c["mask"]), "type": "refinement",
            "seq_orig": seq_orig}
    cfg = rh.sampling_cfg("random", num_timesteps=200, refine_mode="uniform", refine_offset_ratio=0.1,
                          refine_lambda=3.0)
    tr = trajectory(m, spec, B, cfg, cond, seed=23)
    assert len(tr["steps"]) == 200
    for k, v in tr.items():
        out["ref_" + k] = v
    out["ref_cond_seq"] = c["seq"].astype(np.int16)
    out["ref_cond_mask"] = c["mask"]
    out["ref_seq_orig"] = seq_orig.numpy().astype(np.int16)
    from trainer.helpers.task import _index_to_smoothed_log_onehot

    table = _index_to_smoothed_log_onehot(torch.arange(spec.n_class)[None], tok, mode="uniform",
                                          offset_ratio=0.1)[0].T.contiguous() * 3.0
    out["ref_weak_table"] = table.numpy()
    # relation (logit_adjustment.py:88-126 inside _sample_single_step, base.py:261-269)
    graph, seq, mask = _relation_cond(spec, B, seed=29)
    cond = {"seq": seq, "mask": mask, "type": "relation", "batch_w_canvas": graph}
    cfg = rh.sampling_cfg("random", num_timesteps=200, relation_lambda=3e6, relation_mode="average", relation_tau=1.0,
                          relation_num_update=3)
    tr = trajectory(m, spec, B, cfg, cond, seed=31)
    for k, v in tr.items():
        out["rel_" + k] = v
    out["rel_order_dependent"] = relation_order_dependent_tokens(m, tok, spec, tr, cond, cfg)
    bt = tok.bbox_tokenizer
    out["rel_cond_seq"] = seq.numpy().astype(np.int16)
    out["rel_cond_mask"] = mask.numpy()
    out["rel_y"], out["rel_edge_index"] = graph.y.numpy(), graph.edge_index.numpy()
    out["rel_edge_attr"], out["rel_batch"] = graph.edge_attr.numpy(), graph.batch.numpy()
    out["rel_centres"] = np.stack([np.asarray(bt.clustering_models[f"{k}-{spec.n_bin}"].cluster_centers_,
                                              np.float64).reshape(-1) for k in ("x", "y", "w", "h")])
    canvas_ids = bt.encode(torch.tensor([[[0.5, 0.5, 1.0, 1.0]]])).long().view(-1)
    out["rel_canvas_bins"] = (canvas_ids - torch.arange(4) * spec.n_bin).numpy().astype(np.int32)
    return out


def metrics_layouts(B=24, S=25, seed=5):
    """Layouts for the alignment / overlap fixture: random ones as decode_layouts_k writes them (grid-quantised boxes, padded
    slots zeroed), plus the edge cases the reference's code branches on: an empty layout, a single element, identical boxes
    (distance 0), zero-area boxes, touching boxes (l_max == r_min), a full layout, and one whose padded slots carry stale
    boxes (the alignment reads them: metric.py:113 masks rows only)."""
    rng = np.random.default_rng(seed)
    bbox = np.zeros((B, S, 4), np.float32)
    mask = np.zeros((B, S), bool)
    for b in range(B):
        n = int(rng.integers(2, S + 1))
        k = rng.integers(0, 32, size=(n, 4))
        bbox[b, :n, 0], bbox[b, :n, 1] = k[:, 0] / 32.0, k[:, 1] / 32.0
        bbox[b, :n, 2], bbox[b, :n, 3] = (k[:, 2] + 1) / 32.0, (k[:, 3] + 1) / 32.0
        mask[b, :n] = True
    mask[0] = False; bbox[0] = 0                                   # empty layout
    mask[1] = False; mask[1, 0] = True; bbox[1, 1:] = 0            # a single element
    bbox[2, 1] = bbox[2, 0]; mask[2, :2] = True                    # two identical boxes
    bbox[3, :3, 2:] = 0; mask[3, :3] = True                        # zero-area boxes
    bbox[4, 0] = [0.25, 0.5, 0.5, 0.5]; bbox[4, 1] = [0.75, 0.5, 0.5, 0.5]; mask[4] = False; mask[4, :2] = True; bbox[4, 2:] = 0  # touching
    n = S; k = rng.integers(0, 32, size=(n, 4)); mask[5] = True    # a full layout
    bbox[5] = np.stack([k[:, 0] / 32.0, k[:, 1] / 32.0, (k[:, 2] + 1) / 32.0, (k[:, 3] + 1) / 32.0], 1)
    bbox[6] = rng.random((S, 4)).astype(np.float32) * 0.5 + 0.1; mask[6] = False; mask[6, :5] = True  # stale boxes in padded slots
    bbox[7] = rng.random((S, 4)).astype(np.float32); mask[7] = rng.random(S) < 0.5                    # un-quantised, holes
    return bbox, mask


def metrics_cases():
    """compute_alignment / compute_overlap of the REAL reference (helpers/metric.py:98-203) on metrics_layouts()."""
    rh.install_stubs()
    from trainer.helpers.metric import compute_alignment, compute_overlap

    bbox, mask = metrics_layouts()
    out = {"bbox": bbox, "mask": mask}
    for fn in (compute_alignment, compute_overlap):
        for k, v in fn(torch.from_numpy(bbox.copy()), torch.from_numpy(mask.copy())).items():
            out[k] = v.numpy().astype(np.float32)
    return out


def fid_cases(num_label=25, B=6, N=25):
    """FIDNetV3.extract_features (trainer/fid/model.py:147-152) of the REAL reference class on the synthetic
    checkpoint of oracle/fid.py (decoder-half parameters keep their torch init: they do not enter the features)."""
    rh.install_stubs()
    from trainer.fid.model import FIDNetV3

    from . import fid as OF

    torch.manual_seed(0)
    m = FIDNetV3(num_label=num_label, max_bbox=N)
    sd = OF.synth_fid_state_dict(num_label, seed=0, max_bbox=N)
    missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not missing.unexpected_keys and all(
        k.startswith(("dec_", "pos_token", "fc_out_", "enc_transformer.token_mask")) for k in missing.missing_keys)
    m.eval()
    bbox, label, pm = OF.synth_layouts(num_label, B, N, seed=0)
    with torch.no_grad():
        feat = m.extract_features(torch.from_numpy(bbox), torch.from_numpy(label), torch.from_numpy(pm))
    return {"bbox": bbox, "label": label, "padding_mask": pm, "features": feat.numpy(),
            "num_label": np.int32(num_label)}


def main(out_dir=None, only=None):
    """Writes every fixture into out_dir (default tests/golden).  tests/test_oracle_vs_reference.py regenerates them
    into a temporary directory and compares with the committed files, so the pin is re-checked by CI wherever the
    reference tree is present."""
    global OUT
    if out_dir is not None:
        OUT = out_dir
    os.makedirs(OUT, exist_ok=True)
    if only is None and out_dir is None:
        only = sys.argv[1] if len(sys.argv) > 1 else None
    if only == "relation":
        np.savez_compressed(os.path.join(OUT, "rico25_relation.npz"), **relation_cases(SP.SPECS["rico25"]))
        return
    if only == "vanilla":
        np.savez_compressed(os.path.join(OUT, "rico25_vanilla.npz"), **vanilla_cases(SP.SPECS["rico25"]))
        return
    if only == "cond_variants":
        m, _ = rh.build_reference_model("rico25", seed=0)
        load_synth(m, SP.SPECS["rico25"])
        np.savez_compressed(os.path.join(OUT, "rico25_cond_variants.npz"), **cond_variant_cases(m, SP.SPECS["rico25"]))
        return
    if only == "metrics":
        np.savez_compressed(os.path.join(OUT, "layout_metrics.npz"), **metrics_cases())
        return
    if only == "fid":
        np.savez_compressed(os.path.join(OUT, "fid_v3.npz"), **fid_cases())
        return
    if only == "trained_like":
        np.savez_compressed(os.path.join(OUT, "rico25_trained_like.npz"), **trained_like_cases(SP.SPECS["rico25"]))
        np.savez_compressed(os.path.join(OUT, "publaynet_trained_like.npz"), **trained_like_cases(SP.SPECS["publaynet"], points=["mid"]))
        return
    if only == "getcond":
        np.savez_compressed(os.path.join(OUT, "rico25_getcond.npz"), **getcond_cases(SP.SPECS["rico25"]))
        return
    if only == "config5":
        np.savez_compressed(os.path.join(OUT, "rico25_config5_T200.npz"), **config5_cases())
        return
    if only == "decode":
        for ds in ("rico25", "publaynet"):
            _, tok = rh.build_reference_model(ds, seed=0)
            np.savez_compressed(os.path.join(OUT, f"{ds}_decode.npz"), **decode_cases(tok, SP.SPECS[ds]))
        return
    np.savez_compressed(os.path.join(OUT, "fid_v3.npz"), **fid_cases())
    np.savez_compressed(os.path.join(OUT, "layout_metrics.npz"), **metrics_cases())
    np.savez_compressed(os.path.join(OUT, "rico25_trained_like.npz"), **trained_like_cases(SP.SPECS["rico25"]))
    np.savez_compressed(os.path.join(OUT, "publaynet_trained_like.npz"), **trained_like_cases(SP.SPECS["publaynet"], points=["mid"]))
    np.savez_compressed(os.path.join(OUT, "rico25_config5_T200.npz"), **config5_cases())
    np.savez_compressed(os.path.join(OUT, "rico25_getcond.npz"), **getcond_cases(SP.SPECS["rico25"]))
    for ds in ("rico25", "publaynet"):
        spec = SP.SPECS[ds]
        m, tok = rh.build_reference_model(ds, seed=0)
        # schedule buffers + key/shape manifest straight from the reference
        sd0 = m.state_dict()
        np.savez_compressed(
            os.path.join(OUT, f"{ds}_schedule.npz"),
            **{k: v.numpy() for k, v in sd0.items() if "_log_" in k},
        )
        with open(os.path.join(OUT, f"{ds}_state_dict_manifest.txt"), "w") as f:
            for k, v in rh.state_dict_layoutdm_keys(m).items():
                f.write(f"{k} {tuple(v.shape)} {str(v.dtype).replace('torch.', '')}\n")
        np.savez_compressed(os.path.join(OUT, f"{ds}_decode.npz"), **decode_cases(tok, spec))
        load_synth(m, spec)
        np.savez_compressed(os.path.join(OUT, f"{ds}_step_cases.npz"), **step_cases(m, spec, [99, 60, 20, 1, 0]))

        if ds == "rico25":
            np.savez_compressed(os.path.join(OUT, "rico25_vanilla.npz"), **vanilla_cases(spec))
            np.savez_compressed(os.path.join(OUT, "rico25_relation.npz"), **relation_cases(spec))
            tr = trajectory(m, spec, 4, rh.sampling_cfg("random"), None, seed=0)
            np.savez_compressed(os.path.join(OUT, "rico25_uncond_trajectory.npz"), **tr)
            torch.manual_seed(0)
            full = m.sample(batch_size=4, cond=None, sampling_cfg=rh.sampling_cfg("deterministic"),
                            get_intermediate_results=True)
            np.savez_compressed(os.path.join(OUT, "rico25_uncond_greedy_loop.npz"),
                                states_after=torch.stack(full).numpy().astype(np.int16))
            # strided schedule (num_timesteps=25 -> skip_step=3): base.py:227-235
            torch.manual_seed(0)
            cfg = rh.sampling_cfg("deterministic", num_timesteps=25)
            full = m.sample(batch_size=2, cond=None, sampling_cfg=cfg, get_intermediate_results=True)
            np.savez_compressed(os.path.join(OUT, "rico25_uncond_greedy_T25.npz"),
                                states_after=torch.stack(full).numpy().astype(np.int16))
            # refinement: cond built like helpers/task.py:126-138 (LayoutDM branch)
            c = synth.synth_cond_c(spec, 3, seed=5)
            g = torch.Generator().manual_seed(9)
            seq_orig = torch.from_numpy(c["seq"]).clone()
            for a in range(1, spec.n_attr):
                ids = torch.as_tensor(spec.full_ids(a))
                valid = torch.from_numpy(c["seq"][:, a::spec.n_attr] == spec.mask_id)
                rnd = ids[torch.randint(0, spec.n_bin, (3, spec.max_elem), generator=g)]
                seq_orig[:, a::spec.n_attr] = torch.where(valid, rnd, seq_orig[:, a::spec.n_attr])
            cond = {"seq": torch.from_numpy(c["seq"]), "mask": torch.from_numpy(c["mask"]),
                    "type": "refinement", "seq_orig": seq_orig}
            cfg = rh.sampling_cfg("random", refine_mode="uniform", refine_offset_ratio=0.1, refine_lambda=3.0)
            tr = trajectory(m, spec, 3, cfg, cond, seed=3)
            tr["cond_seq"] = c["seq"].astype(np.int16)
            tr["cond_mask"] = c["mask"]
            tr["seq_orig"] = seq_orig.numpy().astype(np.int16)
            # the (C,C) prior table the reference builds (helpers/task.py:154-201), lambda applied
            from trainer.helpers.task import _index_to_smoothed_log_onehot
            table = _index_to_smoothed_log_onehot(torch.arange(spec.n_class)[None], tok, mode="uniform",
                                                  offset_ratio=0.1)[0].T.contiguous() * 3.0
            tr["weak_table"] = table.numpy()  # [token, class]
            np.savez_compressed(os.path.join(OUT, "rico25_refinement_trajectory.npz"), **tr)
            np.savez_compressed(os.path.join(OUT, "rico25_cond_variants.npz"), **cond_variant_cases(m, spec))
        else:
            c = synth.synth_cond_c(spec, 4, seed=0)
            cond = {"seq": torch.from_numpy(c["seq"]), "mask": torch.from_numpy(c["mask"]), "type": "c"}
            cfg = rh.sampling_cfg("top_p", top_p=0.9)
            tr = trajectory(m, spec, 4, cfg, cond, seed=1)
            tr["cond_seq"] = c["seq"].astype(np.int16)
            tr["cond_mask"] = c["mask"]
            # top-p probabilities at three states of that trajectory
            for i in (0, 50, 98):
                toks = torch.from_numpy(tr["states_before"][i].astype(np.int64))
                p = capture_probs(m, spec, toks[:2], int(tr["steps"][i]), cfg,
                                  {"seq": cond["seq"][:2], "mask": cond["mask"][:2], "type": "c"})
                tr[f"top_p_probs_{i}"] = p.numpy()
            np.savez_compressed(os.path.join(OUT, "publaynet_cond_c_trajectory.npz"), **tr)
        print(ds, "done", file=sys.stderr)


if __name__ == "__main__":
    main()
