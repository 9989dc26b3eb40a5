"""cond=relation through the split-step API.

The reference interleaves a gradient-descent logit adjustment between the posterior and the draw
(`update()`, trainer/models/categorical_diffusion/logit_adjustment.py:88-126, called at
base.py:261-269).  Here every stage runs in libldm_hip.so on device tensors:
ldm_denoise_logits -> ldm_posterior -> ldm_relation_update (analytic gradient of the 14 relational
losses, kernels_relation.hip) -> ldm_sample_tokens.

The HIP update covers relation_mode="average" (the reference's default; "gumbel" `did not work at all`,
logit_adjustment.py:25) with the LayoutDM tokenizer (stacked x-y-w-h vocabulary, 1-D cluster centres).
Anything else — or an explicit `update_fn(t, cond, model_log_prob, tokenizer, sampling_cfg)` — goes
through that callable instead (default: the reference's own PyTorch function when the package is installed).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch

from .diffusion import HipMaskAndReplaceDiffusion, _cfg_get, timestep_schedule

LOG_EPS = -69.07755278982137


def _reference_update():
    try:
        from trainer.models.categorical_diffusion.logit_adjustment import update
    except Exception as e:  # pragma: no cover - depends on the deployment
        raise RuntimeError(
            "cond=relation needs trainer.models.categorical_diffusion.logit_adjustment.update "
            "(the reference package) or an explicit update_fn") from e
    return update


def graph_to_csr(graph, n_graph: int):
    """cond["batch_w_canvas"] (torch_geometric DataBatch fields: batch (nodes,), edge_index (2,E) GLOBAL node ids,
    edge_attr (E,); helpers/task.py:112-114, data/util.py:128-177) -> per-layout CSR with LOCAL node ids, the form
    ldm_relation_update takes: (offsets int32 (n_graph+1,), src int32 (E,), dst int32 (E,), attr int32 (E,)), edges
    of a layout kept in their original order."""
    get = (lambda k: graph[k]) if isinstance(graph, dict) else (lambda k: getattr(graph, k))
    batch = torch.as_tensor(get("batch")).long().cpu()
    ei = torch.as_tensor(get("edge_index")).long().cpu().reshape(2, -1)
    ea = torch.as_tensor(get("edge_attr")).long().cpu().reshape(-1)
    if batch.numel() and int(batch.max()) >= n_graph:
        raise ValueError("graph batch index exceeds the number of layouts")
    if ei.shape[1] != ea.numel():
        raise ValueError("edge_index / edge_attr length mismatch")
    n_nodes = torch.bincount(batch, minlength=n_graph)
    first = torch.cat([n_nodes.new_zeros(1), n_nodes.cumsum(0)])[:-1]
    eg = batch[ei[0]] if ei.numel() else torch.zeros(0, dtype=torch.long)
    if ei.numel() and not bool((batch[ei[1]] == eg).all()):
        raise ValueError("an edge connects nodes of two different layouts")
    order = torch.argsort(eg, stable=True)
    eg, src, dst, ea = eg[order], ei[0][order], ei[1][order], ea[order]
    off = torch.cat([eg.new_zeros(1), torch.bincount(eg, minlength=n_graph).cumsum(0)])
    return off.int(), (src - first[eg]).int(), (dst - first[eg]).int(), ea.int()


def hip_relation_plan(eng, cond: Dict, sampling_cfg, tokenizer, batch_size: int):
    """(LdmRelation, keep-alives) when the HIP logit adjustment applies, else None."""
    try:
        if str(_cfg_get(sampling_cfg, "relation_mode", "average")) != "average":
            return None
        bt = tokenizer.bbox_tokenizer
        N = tokenizer.N_bbox_per_var
        if bt.shared_bbox_vocab != "x-y-w-h" or list(bt.var_names) != ["x", "y", "w", "h"] or N > 32:
            return None
        import numpy as np

        cs = [np.asarray(bt.clustering_models[f"{k}-{N}"].cluster_centers_, dtype=np.float64).reshape(-1)
              for k in ("x", "y", "w", "h")]
        if any(c.shape != (N,) for c in cs):
            return None
        canvas = bt.encode(torch.tensor([[[0.5, 0.5, 1.0, 1.0]]])).long().view(-1)  # logit_adjustment.py:38-41
        bins = [int(canvas[i]) - i * N for i in range(4)]
        return eng.make_relation(cond["batch_w_canvas"], np.stack(cs), bins,
                                 float(_cfg_get(sampling_cfg, "relation_lambda", 3e6)),
                                 int(_cfg_get(sampling_cfg, "relation_num_update", 3)), batch_size)
    except (AttributeError, KeyError):
        return None


def sample_with_relation(inner: HipMaskAndReplaceDiffusion, batch_size: int, cond: Dict, sampling_cfg, tokenizer,
                         update_fn: Optional[Callable] = None, get_intermediate_results: bool = False,
                         seed: Optional[int] = None, first_layout: int = 0, **_kw):
    """BaseMaskAndReplaceDiffusion.sample for cond["type"] == "relation" (base.py:293-371)."""
    eng = inner.engine
    plan = None
    if update_fn is None:
        plan = hip_relation_plan(eng, cond, sampling_cfg, tokenizer, int(batch_size))
        if plan is None:
            update_fn = _reference_update()
    T = inner.num_timesteps
    t_model, t_post = timestep_schedule(T, int(_cfg_get(sampling_cfg, "num_timesteps", T)),
                                        float(_cfg_get(sampling_cfg, "time_difference", 0.0) or 0.0))
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    B = int(batch_size)
    cond = dict(cond)
    for k, v in list(cond.items()):  # duplicate_cond + .to(device), base.py:321-336
        if isinstance(v, torch.Tensor):
            if v.size(0) == 1 and B > 1:
                v = v.repeat([B] + [1] * (v.dim() - 1))
            cond[k] = v.to(eng.device)
    tokens = cond["seq"].to(dtype=torch.int32).contiguous().clone()
    tokens_cond = tokens.clone()  # the graph's node set is defined by the CONDITIONED sequence (logit_adjustment.py:44)
    # strong mask + PAD-disable are applied by ldm_posterior exactly as base.py:245-251,272-284 would
    # after update(); update() itself must see the strong-masked log-probs (base.py order), so the
    # hook applies them first and they are re-imposed after the adjustment.
    hip_cond = {"seq": cond["seq"], "mask": cond.get("mask"), "type": "relation"}
    seq = cond["seq"].long()
    pos = torch.arange(eng.S, device=eng.device).view(1, -1)
    pad_mask = ((pos % tokenizer.N_var_per_element != 0) & (seq != eng.pad_id))  # (B,S)
    inter = []
    for i, (tm, tp) in enumerate(zip(t_model, t_post)):
        logits = eng.denoise_logits(tokens, tm)
        logp = eng.posterior(logits, tokens, tp, {"seq": hip_cond["seq"], "mask": hip_cond["mask"], "type": "partial"})
        if plan is not None:
            logp = eng.relation_update(logp, tokens_cond, plan, tm)
        else:
            logp = update_fn(t=tm, cond=cond, model_log_prob=logp, tokenizer=tokenizer, sampling_cfg=sampling_cfg)
        with torch.no_grad():
            logp = logp.detach().float().contiguous()
            logp[:, eng.pad_id, :] = torch.where(pad_mask, torch.full_like(logp[:, eng.pad_id, :], LOG_EPS),
                                                 logp[:, eng.pad_id, :])
            tokens = eng.sample_tokens(logp, sampling_cfg, seed=seed, first_layout=first_layout, step=i)
        if get_intermediate_results:
            inter.append(tokens.long().cpu())
    return inter if get_intermediate_results else tokens.long().cpu()
