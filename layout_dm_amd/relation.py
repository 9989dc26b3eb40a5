"""cond=relation.

The reference interleaves a gradient-descent logit adjustment between the posterior and the draw
(`update()`, trainer/models/categorical_diffusion/logit_adjustment.py:88-126, called at base.py:261-269).
Here the whole step — posterior (+ strong mask) -> ldm_relation_update (analytic gradient of the 14 relational
losses, kernels_relation.hip) -> [PAD] disable -> draw — runs inside ldm_sample_loop, i.e. inside the same
hipGraph as every other cond type (`sample_with_relation`, default path).

Covered: relation_mode="average" (the reference's default) with the LayoutDM tokenizer (stacked x-y-w-h vocabulary,
1-D cluster centres, <= 32 bins).  relation_mode="gumbel" — which the reference itself reports as not working
(logit_adjustment.py:25) — and other vocabularies raise NotImplementedError: there is no PyTorch / reference
fallback in the product path.  A caller-supplied `update_fn(t, cond, model_log_prob, tokenizer, sampling_cfg)`
(research use, tests) is driven through the split-step API instead: ldm_denoise_logits -> ldm_posterior ->
update_fn -> ldm_sample_tokens.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, Optional

import torch

from .diffusion import HipMaskAndReplaceDiffusion, _cfg_get, batch_cuts, timestep_schedule


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


def relation_geometry(tokenizer):
    """Pure host logic: ((4, n_bin) float64 cluster centres in x,y,w,h order, canvas bins per coordinate) — the
    constants `_stochastic_convert` uses (logit_adjustment.py:30-41): the centres of the bbox vocabulary and the
    tokens of the canvas box (0.5, 0.5, 1, 1).  Raises NotImplementedError for vocabularies the kernel does not
    cover."""
    import numpy as np

    bt = tokenizer.bbox_tokenizer
    N = tokenizer.N_bbox_per_var
    if bt.shared_bbox_vocab != "x-y-w-h" or list(bt.var_names) != ["x", "y", "w", "h"]:
        raise NotImplementedError("cond=relation: only the stacked x-y-w-h bbox vocabulary (LayoutDM) is supported")
    if N > 32:
        raise NotImplementedError("cond=relation: more than 32 bins per coordinate")
    cs = [np.asarray(bt.clustering_models[f"{k}-{N}"].cluster_centers_, dtype=np.float64).reshape(-1)
          for k in ("x", "y", "w", "h")]
    if any(c.shape != (N,) for c in cs):
        raise NotImplementedError("cond=relation: cluster centres must be one-dimensional")
    canvas = bt.encode(torch.tensor([[[0.5, 0.5, 1.0, 1.0]]])).long().view(-1)  # logit_adjustment.py:38-41
    bins = [int(canvas[i]) - i * N for i in range(4)]
    return np.stack(cs), bins


def hip_relation_plan(eng, cond: Dict, sampling_cfg, tokenizer, batch_size: int):
    """(LdmRelation, keep-alives) for ldm_sample_loop / ldm_relation_update."""
    mode = str(_cfg_get(sampling_cfg, "relation_mode", "average"))
    if mode != "average":
        raise NotImplementedError(
            f"relation_mode={mode}: only 'average' (the reference's default; its 'gumbel' variant is reported as not "
            "working, logit_adjustment.py:25) is implemented")
    centres, bins = relation_geometry(tokenizer)
    return eng.make_relation(cond["batch_w_canvas"], centres, bins,
                             float(_cfg_get(sampling_cfg, "relation_lambda", 3e6)),
                             int(_cfg_get(sampling_cfg, "relation_num_update", 3)), batch_size)


def _relation_window(plan, lo: int):
    """The same relation graph seen from layout `lo` on: CSR offsets are absolute positions in the edge arrays, so a
    window is a pointer offset on the offsets array (n_graph_total — the loss normaliser — stays the call's)."""
    from .binding import LdmRelation

    rel, keep = plan
    if lo == 0:
        return plan
    win = LdmRelation()
    C.memmove(C.byref(win), C.byref(rel), C.sizeof(LdmRelation))
    win.d_edge_offsets = rel.d_edge_offsets + 4 * lo
    return win, keep


def sample_with_relation(inner: HipMaskAndReplaceDiffusion, batch_size: int, cond: Dict, sampling_cfg, tokenizer,
                         update_fn: Optional[Callable] = None, get_intermediate_results: bool = False,
                         seed: Optional[int] = None, first_layout: int = 0, return_device_tensor: bool = False, **_kw):
    """BaseMaskAndReplaceDiffusion.sample for cond["type"] == "relation" (base.py:293-371)."""
    eng = inner.engine
    if getattr(inner, "verified", None) is not None and str(_cfg_get(sampling_cfg, "name")) == "deterministic":
        # fast_verified promises the exact mode's greedy tokens; the near-tie report does not exist for the adjusted
        # steps (ldm_sample_loop refuses it), so greedy cond=relation decoding runs on the exact engine
        eng = inner.verified.exact
    T = inner.num_timesteps
    t_model, t_post = timestep_schedule(T, int(_cfg_get(sampling_cfg, "num_timesteps", T)),
                                        float(_cfg_get(sampling_cfg, "time_difference", 0.0) or 0.0))
    if seed is None:   # (deterministic decoding: no draw from torch's global generator, like the reference — diffusion.sample)
        seed = 0 if str(_cfg_get(sampling_cfg, "name")) == "deterministic" else int(torch.randint(0, 2 ** 62, (1,)).item())
    B = int(batch_size)
    cond = dict(cond)
    for k, v in list(cond.items()):  # duplicate_cond + .to(device), base.py:321-336
        if isinstance(v, torch.Tensor):
            if v.dim() > 0 and v.size(0) == 1 and B > 1:
                v = v.repeat([B] + [1] * (v.dim() - 1))
            cond[k] = v.to(eng.device)
    tokens = cond["seq"].to(dtype=torch.int32).contiguous().clone()
    hip_cond = {"seq": cond["seq"], "mask": cond.get("mask"), "type": "relation"}

    if update_fn is None:
        # fused path: every stage of every step inside ldm_sample_loop (one hipGraph per max_batch window)
        plan = hip_relation_plan(eng, cond, sampling_cfg, tokenizer, B)
        outs, inters = [], []
        off = 0
        for n in batch_cuts(B, eng.max_batch, eng.batch_round):   # (whole rounds of the chip per call: diffusion.batch_cuts)
            sub = {"seq": hip_cond["seq"][off:off + n],
                   "mask": hip_cond["mask"][off:off + n] if hip_cond["mask"] is not None else None, "type": "relation"}
            tk, inter = eng.sample_loop(tokens[off:off + n].contiguous(), t_model, t_post, sampling_cfg, cond=sub,
                                        seed=seed, first_layout=first_layout + off,
                                        intermediates=get_intermediate_results, use_graph=inner.use_graph,
                                        relation=_relation_window(plan, off))
            outs.append(tk)
            inters.append(inter)
            off += n
        if eng.device.type == "cuda":
            torch.cuda.current_stream(eng.device).synchronize()  # `plan` keep-alives may go out of scope
        if get_intermediate_results:
            inter = torch.cat(inters, dim=1) if len(inters) > 1 else inters[0]
            return [x.long().cpu() for x in inter]
        out = torch.cat(outs) if len(outs) > 1 else outs[0]
        return out if return_device_tensor else out.long().cpu()

    # split-step path for a caller-supplied update function
    if B > eng.max_batch:
        raise ValueError(f"split-step cond=relation sampling is limited to max_batch={eng.max_batch} layouts per call")
    inter = []
    for i, (tm, tp) in enumerate(zip(t_model, t_post)):
        logits = eng.denoise_logits(tokens, tm)
        # strong mask applied by ldm_posterior (type "partial": no [PAD] disabling yet — base.py applies it after
        # update(), which ldm_sample_tokens does below)
        logp = eng.posterior(logits, tokens, tp, {"seq": hip_cond["seq"], "mask": hip_cond["mask"], "type": "partial"})
        logp = update_fn(t=tm, cond=cond, model_log_prob=logp, tokenizer=tokenizer, sampling_cfg=sampling_cfg)
        logp = logp.detach().float().contiguous()
        tokens = eng.sample_tokens(logp, sampling_cfg, seed=seed, first_layout=first_layout, step=i,
                                   cond={"seq": hip_cond["seq"], "type": "relation"})
        if get_intermediate_results:
            inter.append(tokens.long().cpu())
    if get_intermediate_results:
        return inter
    return tokens if return_device_tensor else tokens.long().cpu()
