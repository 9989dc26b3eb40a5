"""cond=relation through the split-step API.

The reference interleaves an autograd-driven logit adjustment between the posterior and the draw
(`update()`, trainer/models/categorical_diffusion/logit_adjustment.py:88-126, called at
base.py:261-269).  That optimiser stays in PyTorch; the denoiser forward, the posterior + cond
overrides and the categorical draw still run in libldm_hip.so through the three parity hooks
(ldm_denoise_logits / ldm_posterior / ldm_sample_tokens), all on device tensors, no host copies.

`update_fn(t, cond, model_log_prob, tokenizer, sampling_cfg) -> model_log_prob` defaults to the
reference's own function (the reference package is installed in a drop-in deployment); any callable
with that signature can be injected.
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


def sample_with_relation(inner: HipMaskAndReplaceDiffusion, batch_size: int, cond: Dict, sampling_cfg, tokenizer,
                         update_fn: Optional[Callable] = None, get_intermediate_results: bool = False,
                         seed: Optional[int] = None, first_layout: int = 0, **_kw):
    """BaseMaskAndReplaceDiffusion.sample for cond["type"] == "relation" (base.py:293-371)."""
    eng = inner.engine
    update_fn = update_fn or _reference_update()
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
        logp = update_fn(t=tm, cond=cond, model_log_prob=logp, tokenizer=tokenizer, sampling_cfg=sampling_cfg)
        with torch.no_grad():
            logp = logp.detach().float().contiguous()
            logp[:, eng.pad_id, :] = torch.where(pad_mask, torch.full_like(logp[:, eng.pad_id, :], LOG_EPS),
                                                 logp[:, eng.pad_id, :])
            tokens = eng.sample_tokens(logp, sampling_cfg, seed=seed, first_layout=first_layout, step=i)
        if get_intermediate_results:
            inter.append(tokens.long().cpu())
    return inter if get_intermediate_results else tokens.long().cpu()
