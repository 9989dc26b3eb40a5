"""`fast_verified`: greedy decoding at (almost) the fp16 mode's speed with the EXACT mode's tokens.

north star: "token indices bit-exact under greedy/argmax decoding".  The fast numerics mode (fp16 operands, fp32
accumulate) carries <= 1e-3 relative logits error, so its argmax (helpers/sampling.py:88-90) can differ from the
reference's where two classes are closer than that error can move them — 1 token in ~30 000 on the reference's own
trajectories.  Those places are detectable from inside the fast pass: with the near-tie report enabled
(`ldm_set_tie_report`, include/ldm_hip.h) every deterministic step marks, per (step, layout), whether some token was decided
with a log-probability lead over the runner-up below `tie_rel * max |logit|`.  The lead's error is bounded by 6x the
largest logit error (2 from the difference of two log-softmax values, the rest from the posterior's
log-sum-exp terms with derivative <= 1), so with tie_rel = 6 x the mode's relative logits tolerance an UNMARKED token is the
exact mode's token.  Greedy decoding is RNG-free and layouts are independent, so a marked layout is simply re-decided in
the exact mode from its state before its first marked step, and spliced in.

The reference-side contract (base.py:205-291,293-371) is unchanged: tokens in, tokens out.  Cost = the fast loop + the
exact mode on (layouts marked) x (steps after their first mark).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch

from .binding import Engine

# 6 x (1e-3 relative logits tolerance of LDM_PREC_FAST_F16, tests/test_hip_parity.py LOGIT_REL_TOL); the measured logits
# error is 3.6e-4 .. 4.6e-4, and the largest reference top-2 margin ever seen among the fast mode's mismatches is 3.4e-4
# of log-probability on logits of magnitude ~2
DEFAULT_TIE_REL = 6e-3
GREEDY = {"name": "deterministic"}


class VerifiedGreedy:
    """A fast-mode engine and an exact-mode engine of the same model; deterministic decoding only."""

    def __init__(self, fast: Engine, exact: Engine, tie_rel: float = DEFAULT_TIE_REL):
        assert fast.S == exact.S and fast.C == exact.C and fast.device == exact.device
        self.fast, self.exact, self.tie_rel = fast, exact, float(tie_rel)
        self.last_stats: Dict[str, float] = {}

    @staticmethod
    def _sub(cond: Optional[dict], idx: torch.Tensor, B: int) -> Optional[dict]:
        if not cond:
            return None
        out = {}
        for k, v in cond.items():
            if isinstance(v, torch.Tensor) and v.dim() > 0 and v.size(0) == B:
                out[k] = v[idx.to(v.device)]
            elif hasattr(v, "shape") and getattr(v, "ndim", 0) > 0 and v.shape[0] == B:  # numpy
                out[k] = v[idx.cpu().numpy()]
            else:
                out[k] = v
        return out

    def sample_step(self, tokens: torch.Tensor, t_model: int, t_post: Optional[int] = None, cond: Optional[dict] = None,
                    step: int = 0) -> torch.Tensor:
        """One greedy reverse step (_sample_single_step, base.py:205-291): fast everywhere, exact on the marked layouts."""
        f = self.fast
        tokens = f._tok(tokens)
        B = tokens.shape[0]
        f.set_tie_report(self.tie_rel)
        out = f.sample_step(tokens, t_model, GREEDY, t_post=t_post, cond=cond, step=step)
        idx = f.tie_flags(1, B)[0].nonzero().flatten()
        if idx.numel():
            out[idx] = self.exact.sample_step(tokens[idx].contiguous(), t_model, GREEDY, t_post=t_post,
                                              cond=self._sub(cond, idx, B), step=step)
        self.last_stats = {"layouts": B, "marked_layout_steps": int(idx.numel()), "steps": 1}
        return out

    def sample_loop(self, tokens: torch.Tensor, t_model: Sequence[int], t_post: Sequence[int], cond: Optional[dict] = None,
                    intermediates: bool = False):
        """The greedy T-step loop (base.py:293-371), in place on `tokens` (B,S) int32 cuda -> (tokens, intermediates|None)."""
        f = self.fast
        tokens = f._tok(tokens)
        B, n = tokens.shape[0], len(t_model)
        init = tokens.clone()
        f.set_tie_report(self.tie_rel)
        out, inter = f.sample_loop(tokens, t_model, t_post, GREEDY, cond=cond, intermediates=True)
        flags = f.tie_flags(n, B).bool()                         # (n, B)
        marked = flags.any(dim=0)
        first = torch.where(marked, flags.to(torch.int32).argmax(dim=0), torch.full_like(flags[0], n, dtype=torch.int64))
        redo_steps = 0
        for i0 in torch.unique(first[marked]).tolist():          # one exact call per distinct first marked step
            idx = (first == i0).nonzero().flatten()
            start = (init if i0 == 0 else inter[i0 - 1])[idx].contiguous()
            tk, it = self.exact.sample_loop(start, list(t_model[i0:]), list(t_post[i0:]), GREEDY,
                                            cond=self._sub(cond, idx, B), intermediates=intermediates)
            out[idx] = tk
            if intermediates:
                inter[i0:, idx] = it
            redo_steps += int(idx.numel()) * (n - i0)
        self.last_stats = {"layouts": B, "steps": n, "marked_layouts": int(marked.sum()),
                           "marked_layout_steps": int(flags.sum()), "exact_layout_steps": redo_steps,
                           "exact_fraction": redo_steps / float(max(B * n, 1))}
        return out, (inter if intermediates else None)
