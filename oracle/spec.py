"""TEST INFRASTRUCTURE ONLY — shapes/constants of the LayoutDM sampling hot path.

Mirrors (does not import) the reference:
  vocab layout     trainer/helpers/layout_tokenizer.py:79-82,152-153,429-467
  sub-vocab sizes  trainer/models/categorical_diffusion/constrained.py:51-54
  backbone dims    trainer/config/backbone/medium.yaml + models/layoutdm.py:54 (shrink 29/32)
  schedule         trainer/models/categorical_diffusion/util.py:47-70, base.py:44-47
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np

LOG_EPS = math.log(1e-30)  # util.py:8
VAR_NAMES = ("c", "x", "y", "w", "h")


@dataclasses.dataclass(frozen=True)
class ModelSpec:
    name: str
    n_category: int
    n_bin: int = 32
    max_elem: int = 25
    n_attr: int = 5
    d_model: int = 464
    n_head: int = 8
    d_ff: int = 1856
    n_layer: int = 4
    n_step: int = 100  # T

    @property
    def seq_len(self) -> int:  # S
        return self.max_elem * self.n_attr

    @property
    def n_bbox(self) -> int:
        return self.n_bin * 4

    @property
    def pad_id(self) -> int:
        return self.n_category + self.n_bbox

    @property
    def mask_id(self) -> int:
        return self.pad_id + 1

    @property
    def n_class(self) -> int:  # C
        return self.mask_id + 1

    @property
    def d_head(self) -> int:
        return self.d_model // self.n_head

    def sub_vocab_size(self, attr: int) -> int:  # K, constrained.py:51-54
        return (self.n_category if attr == 0 else self.n_bin) + 2

    def full_ids(self, attr: int) -> np.ndarray:
        """partial index -> full vocabulary id (layout_tokenizer.py:429-467)."""
        if attr == 0:
            body = np.arange(self.n_category)
        else:
            start = self.n_category + (attr - 1) * self.n_bin
            body = np.arange(start, start + self.n_bin)
        return np.concatenate([body, [self.pad_id, self.mask_id]]).astype(np.int64)


RICO25 = ModelSpec("rico25", n_category=25)
PUBLAYNET = ModelSpec("publaynet", n_category=5)
SPECS = {"rico25": RICO25, "publaynet": PUBLAYNET}


def alpha_schedule(num_timesteps, N, att_1=0.99999, att_T=0.000009, ctt_1=0.000009, ctt_T=0.99999):
    """Restatement of util.py:47-70 (float64 numpy)."""
    att = np.arange(0, num_timesteps) / (num_timesteps - 1) * (att_T - att_1) + att_1
    att = np.concatenate(([1], att))
    at = att[1:] / att[:-1]
    ctt = np.arange(0, num_timesteps) / (num_timesteps - 1) * (ctt_T - ctt_1) + ctt_1
    ctt = np.concatenate(([0], ctt))
    one_minus_ctt = 1 - ctt
    one_minus_ct = one_minus_ctt[1:] / one_minus_ctt[:-1]
    ct = 1 - one_minus_ct
    bt = (1 - at - ct) / N
    att = np.concatenate((att[1:], [1]))
    ctt = np.concatenate((ctt[1:], [0]))
    btt = (1 - att - ctt) / N
    return at, bt, ct, att, btt, ctt


SCHEDULE_NAMES = (
    "log_at", "log_bt", "log_ct", "log_cumprod_at", "log_cumprod_bt", "log_cumprod_ct",
    "log_1_min_ct", "log_1_min_cumprod_ct",
)


def schedule_buffers(spec: ModelSpec, q_type: str = "constrained"):
    """{f"{key}_{name}": float32 array} as registered at constrained.py:56-90, or — q_type="vanilla" — the
    un-prefixed single-vocabulary buffers of vanilla.py:42-72 (N = C - 1).

    torch.log / log_1_min_a (util.py:15-16) are evaluated in float64 then cast to
    float32, exactly as the reference does (torch.tensor(float64) -> .float()).
    """
    out = {}
    groups = [(f"{key}_", spec.sub_vocab_size(a) - 1) for a, key in enumerate(VAR_NAMES)]
    if q_type == "vanilla":
        groups = [("", spec.n_class - 1)]
    with np.errstate(divide="ignore"):
        for prefix, N in groups:
            at, bt, ct, att, btt, ctt = alpha_schedule(spec.n_step, N)
            log_at, log_bt, log_ct = np.log(at), np.log(bt), np.log(ct)
            l_att, l_btt, l_ctt = np.log(att), np.log(btt), np.log(ctt)
            log_1_min_ct = np.log(1 - np.exp(log_ct) + 1e-40)
            log_1_min_cumprod_ct = np.log(1 - np.exp(l_ctt) + 1e-40)
            vals = (log_at, log_bt, log_ct, l_att, l_btt, l_ctt, log_1_min_ct, log_1_min_cumprod_ct)
            for n, v in zip(SCHEDULE_NAMES, vals):
                out[f"{prefix}{n}"] = v.astype(np.float32)
    return out
