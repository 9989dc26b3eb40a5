"""`fast_verified`: greedy decoding at (almost) the fp16 mode's speed with the EXACT mode's tokens.

north star: "token indices bit-exact under greedy/argmax decoding".  The fast numerics mode (fp16 operands, fp32
accumulate) carries a logits error that depends on the checkpoint (3e-4 of max |logit| on the reference's init, 1e-3 on
wider weights, percents once attention rows saturate — DESIGN.md section 3.5), so its argmax (helpers/sampling.py:88-90) can
differ from the reference's where two classes are closer than that error can move them.  Those places are detectable
from inside the fast pass: with the near-tie report enabled (`ldm_set_tie_report`, include/ldm_hip.h) every
deterministic step marks, per (step, layout), whether some token was decided with a log-probability lead over the
runner-up below max(tie_rel * max |logit of the token|, tie_abs).  The lead's error is bounded by 6x the largest ABSOLUTE
logits error of the token (2 from the difference of two log-softmax values, the rest from the
posterior's log-sum-exp terms with derivative <= 1), so with tie_abs = 6 x a bound on that error an UNMARKED token is the
exact mode's token.  The bound is MEASURED on the checkpoint (`calibrate`: fast vs exact logits on probe states over the
whole timestep range, times a safety factor) — it is an observation on this checkpoint, not a theorem; `audit` re-checks
a random sample of unmarked (step, layout) pairs in the exact mode and reports what it finds.

r04 algorithm — the cost is what the marks cost, not what the tail of the loop costs:

  1. run the fast loop with the intermediates of every step and the near-tie flags;
  2. every marked (step, layout) is re-decided by ONE exact `ldm_sample_step` from the state before that step (all marks
     of a step in one call) and compared with the fast result;
  3. only layouts whose exact tokens DIFFER (measured: ~0.3 % of the marked pairs) are re-launched — in the fast mode —
     from the corrected state, for the remaining steps, and the procedure repeats on their remainder.

Greedy decoding is RNG-free and layouts are independent, so splicing is exact.  By induction over the steps the result is
the exact engine's greedy trajectory wherever the report is sound.  The reference-side contract (base.py:205-291,293-371)
is unchanged: tokens in, tokens out.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from .binding import Engine

# 6 x (1e-3 relative logits tolerance of LDM_PREC_FAST_F16, tests/test_hip_parity.py LOGIT_REL_TOL) on the token's own
# max |logit|: the r03 criterion, kept as a second, scale-following term beside the calibrated absolute floor
DEFAULT_TIE_REL = 6e-3
LEAD_LIPSCHITZ = 6.0     # |delta lead| <= 6 x max |delta logit|
GREEDY = {"name": "deterministic"}


def probe_states(eng: Engine, n_layouts: int = 8, ts: Optional[Sequence[int]] = None, seed: int = 0):
    """[(tokens (n,S) int32 cpu, t)] covering the timestep range: valid tokens of each attribute's sub-vocabulary
    ([PAD] included), masked with probability t / (T - 1) — what states of a reverse trajectory look like at step t."""
    T = eng.T
    if ts is None:
        ts = sorted({T - 1, (3 * T) // 4, T // 2, T // 4, min(2, T - 1), min(1, T - 1), 0}, reverse=True)
    g = torch.Generator().manual_seed(seed)
    out = []
    for t in ts:
        tok = torch.empty((n_layouts, eng.S), dtype=torch.int64)
        for a in range(eng.n_attr):
            if eng.q_type == "vanilla":
                lo, cnt = 0, eng.C - 2
            else:
                lo, cnt = (0, eng.n_category) if a == 0 else (eng.n_category + (a - 1) * eng.n_bin, eng.n_bin)
            r = torch.randint(0, cnt + 1, (n_layouts, eng.S // eng.n_attr), generator=g)
            tok[:, a::eng.n_attr] = torch.where(r == cnt, torch.full_like(r, eng.pad_id), lo + r)
        tok[torch.rand((n_layouts, eng.S), generator=g) < t / max(T - 1, 1)] = eng.mask_id
        out.append((tok.int(), int(t)))
    return out


def trajectory_states(eng: Engine, n_layouts: int = 8, n_states: int = 6, seed: int = 0):
    """[(tokens (n,S) int32 cpu, t)] taken from a REAL reverse trajectory of this checkpoint: one `random` run of the
    engine's own sampling loop (the kernel path that decodes — the one-launch loop in the fast mode) from all-[MASK],
    the states BEFORE `n_states` steps spread over the schedule, the last two steps included (greedy decoding decides
    its tokens there: DESIGN.md section 3.5).  ADVICE r4: probe states drawn uniformly from the sub-vocabularies are
    not what a trajectory visits once the model has started to commit tokens."""
    T = eng.T
    t_model = list(range(T - 1, -1, -1))
    tok = torch.full((n_layouts, eng.S), eng.mask_id, dtype=torch.int32, device=eng.device)
    _, inter = eng.sample_loop(tok, t_model, t_model, {"name": "random", "temperature": 1.0}, seed=seed,
                               intermediates=True)
    picks = {int(round(i * (T - 1) / max(n_states - 1, 1))) for i in range(n_states)} | {T - 2, T - 1}
    # (the state before step 0 is all-[MASK]: probe_states covers it)
    return [(inter[i - 1].cpu().clone(), t_model[i]) for i in sorted(picks) if 0 < i < T]


def measure_fast_error(fast: Engine, exact: Engine, states=None) -> Dict[str, float]:
    """Largest logits error of the fast engine against the exact engine of the same weights on `states` (default:
    probe_states + states of a real trajectory of the fast engine's own sampling loop): absolute, relative to the
    largest |logit| of the probe, and relative per row.  Non-finite logits of either engine are reported (`finite`)
    and make every error infinite: a NaN must never compare as "inside the tolerance"."""
    if states is None:
        n = max(1, min(8, fast.max_batch, exact.max_batch))
        states = probe_states(fast, n_layouts=n) + trajectory_states(fast, n_layouts=n)
    e_abs = e_rel = e_row = absmax = 0.0
    finite = True
    for tok, t in states:
        lf, le = fast.denoise_logits(tok, t), exact.denoise_logits(tok, t)
        if not (bool(torch.isfinite(lf).all()) and bool(torch.isfinite(le).all())):
            finite = False
            continue
        d = (lf - le).abs()
        m = le.abs().max().item()
        e_abs = max(e_abs, d.max().item())
        e_rel = max(e_rel, d.max().item() / max(m, 1e-30))
        e_row = max(e_row, (d.amax(-1) / le.abs().amax(-1).clamp_min(1e-30)).max().item())
        absmax = max(absmax, m)
    if not finite:
        e_abs = e_rel = e_row = float("inf")
    return {"err_abs": e_abs, "err_rel": e_rel, "err_rel_row": e_row, "absmax": absmax, "n_states": len(states),
            "finite": finite}


class VerifiedGreedy:
    """A fast-mode engine and an exact-mode engine of the same model; deterministic decoding only."""

    # audit: fraction of the UNMARKED (step, layout) pairs re-decided by the reference-precision engine anyway.  The
    # calibrated threshold is an observation on probe states, not a bound, so a small sample is checked by default
    # (ADVICE r4): 0.5 % of the pairs = 256 layout-steps of a 512 x 100 call, one extra reference-precision step per
    # audited timestep.  A mismatch there is a SOUNDNESS failure of the report for this checkpoint: it is counted in
    # last_stats / audit_mismatch_total, warned about, and raised with strict_audit=True.
    DEFAULT_AUDIT = 0.005

    def __init__(self, fast: Engine, exact: Engine, tie_rel: float = DEFAULT_TIE_REL, tie_abs: float = 0.0,
                 safety: float = 2.0, audit: Optional[float] = None, strict_audit: bool = False):
        assert fast.S == exact.S and fast.C == exact.C and fast.device == exact.device
        self.fast, self.exact = fast, exact
        audit = self.DEFAULT_AUDIT if audit is None else audit
        self.tie_rel, self.tie_abs, self.safety, self.audit = float(tie_rel), float(tie_abs), float(safety), float(audit)
        self.strict_audit = bool(strict_audit)
        self.audit_mismatch_total = 0
        self.calibration: Dict[str, float] = {}
        self.last_stats: Dict[str, float] = {}
        self._audit_gen = torch.Generator().manual_seed(0)

    # ------------------------------------------------------------------ the measured error bound
    def calibrate(self, states=None) -> Dict[str, float]:
        """tie_abs <- 6 x safety x (largest absolute fast-vs-exact logits error on the probe states).  Call after the
        weights are loaded (HipMaskAndReplaceDiffusion.load_state_dict does)."""
        c = measure_fast_error(self.fast, self.exact, states)
        if not c["finite"]:
            raise FloatingPointError("fast_verified / auto: non-finite logits on the probe states of this checkpoint "
                                     "(fp16 overflow of an operand, or a broken checkpoint): no threshold can be calibrated")
        self.tie_abs = LEAD_LIPSCHITZ * self.safety * c["err_abs"]
        c["tie_abs"], c["tie_rel"], c["safety"] = self.tie_abs, self.tie_rel, self.safety
        self.calibration = c
        return c

    @staticmethod
    def _refuse_relation(cond: Optional[dict]):
        if cond and cond.get("type") == "relation":
            raise NotImplementedError("fast_verified: cond=relation has no near-tie report (its draw follows an SGD on the "
                                      "log-probabilities); decode it with the reference-precision engine "
                                      "(relation.sample_with_relation does)")

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

    AUDIT_STEPS = 4  # audited pairs are concentrated on this many random steps per pass: every audited step costs one
                     # reference-precision launch sequence whatever the number of layouts in it

    def _audit_mask(self, flags: torch.Tensor) -> Optional[torch.Tensor]:
        """(m, b) bool: ~audit * m * b UNMARKED pairs, on AUDIT_STEPS random steps of the pass."""
        if self.audit <= 0.0:
            return None
        m, b = flags.shape
        k = min(m, max(self.AUDIT_STEPS, int(-(-self.audit * m // 1))))   # (audit = 1: every step, every layout)
        per_step = min(b, max(1, int(round(self.audit * m * b / k))))
        pick = torch.zeros((m, b), dtype=torch.bool)
        for s in torch.randperm(m, generator=self._audit_gen)[:k].tolist():
            pick[s, torch.randperm(b, generator=self._audit_gen)[:per_step]] = True
        return pick.to(flags.device) & ~flags

    def sample_step(self, tokens: torch.Tensor, t_model: int, t_post: Optional[int] = None, cond: Optional[dict] = None,
                    step: int = 0) -> torch.Tensor:
        """One greedy reverse step (_sample_single_step, base.py:205-291): fast everywhere, exact on the marked layouts."""
        self._refuse_relation(cond)
        f = self.fast
        tokens = f._tok(tokens)
        B = tokens.shape[0]
        f.set_tie_report(self.tie_rel, self.tie_abs)
        out = f.sample_step(tokens, t_model, GREEDY, t_post=t_post, cond=cond, step=step)
        idx = f.tie_flags(1, B)[0].nonzero().flatten()
        changed = 0
        if idx.numel():
            ex = self.exact.sample_step(tokens[idx].contiguous(), t_model, GREEDY, t_post=t_post,
                                        cond=self._sub(cond, idx, B), step=step)
            changed = int((ex != out[idx]).any(dim=1).sum())
            out[idx] = ex
        self.last_stats = {"layouts": B, "steps": 1, "marked_layout_steps": int(idx.numel()),
                           "exact_layout_steps": int(idx.numel()), "mismatch_layout_steps": changed}
        return out

    def sample_loop(self, tokens: torch.Tensor, t_model: Sequence[int], t_post: Sequence[int], cond: Optional[dict] = None,
                    intermediates: bool = False):
        """The greedy T-step loop (base.py:293-371), in place on `tokens` (B,S) int32 cuda -> (tokens, intermediates|None)."""
        self._refuse_relation(cond)
        f, e = self.fast, self.exact
        tokens = f._tok(tokens)
        B, n, S = tokens.shape[0], len(t_model), f.S
        t_model, t_post = [int(x) for x in t_model], [int(x) for x in t_post]
        f.set_tie_report(self.tie_rel, self.tie_abs)
        inter_all = torch.empty((n, B, S), dtype=torch.int32, device=f.device) if intermediates else None
        final = torch.empty_like(tokens)
        st = {"marked": 0, "checked": 0, "mismatch": 0, "relaunched": 0, "passes": 0, "audited": 0, "audit_mismatch": 0}
        # work item: (global layout ids, index of its first step, state before that step)
        work: List = [(torch.arange(B, device=f.device), 0, tokens.clone())]
        while work:
            idx, i0, start = work.pop()
            b, m = int(idx.numel()), n - i0
            sub = cond if b == B else self._sub(cond, idx, B)
            out, inter = f.sample_loop(start.clone(), t_model[i0:], t_post[i0:], GREEDY, cond=sub, intermediates=True)
            flags = f.tie_flags(m, b).bool()                                  # (m, b)
            aud = self._audit_mask(flags)
            check = flags if aud is None else (flags | aud)
            st["marked"] += int(flags.sum())
            st["passes"] += 1
            bad_at = torch.full((b,), m, dtype=torch.int64, device=f.device)     # first step whose fast tokens are wrong
            for s in check.any(dim=1).nonzero().flatten().tolist():           # steps with something to check, ascending
                j = (check[s] & (bad_at > s)).nonzero().flatten()             # (a layout is void behind its first wrong step)
                if not j.numel():
                    continue
                prev = (start if s == 0 else inter[s - 1])[j].contiguous()
                ex = e.sample_step(prev, t_model[i0 + s], GREEDY, t_post=t_post[i0 + s],
                                   cond=self._sub(sub, j, b) if sub else None, step=i0 + s)
                st["checked"] += int(j.numel())
                diff = (ex != inter[s][j]).any(dim=1)
                if aud is not None:
                    st["audited"] += int(aud[s][j].sum())
                    st["audit_mismatch"] += int((diff & aud[s][j]).sum())
                if bool(diff.any()):
                    jd = j[diff]
                    bad_at[jd] = s
                    inter[s][jd] = ex[diff]                                   # the corrected state after step s
                    st["mismatch"] += int(jd.numel())
            if inter_all is not None:
                inter_all[i0:, idx] = inter        # (rows behind a layout's bad_at are overwritten by its re-launch)
            final[idx] = out
            redo = (bad_at < m).nonzero().flatten()
            if redo.numel():
                for s in torch.unique(bad_at[redo]).tolist():
                    jd = (bad_at == s).nonzero().flatten()
                    fixed = inter[s][jd].contiguous()
                    if s == m - 1:                                            # wrong only at the last step: spliced, done
                        final[idx[jd]] = fixed
                    else:
                        work.append((idx[jd], i0 + s + 1, fixed.clone()))
                        st["relaunched"] += int(jd.numel()) * (m - s - 1)
        tokens.copy_(final)
        if st["audit_mismatch"]:
            self.audit_mismatch_total += st["audit_mismatch"]
            msg = (f"fast_verified: {st['audit_mismatch']} of {st['audited']} audited UNMARKED (step, layout) pairs differ from "
                   f"the reference-precision engine: the calibrated near-tie threshold (tie_abs {self.tie_abs:.3g}) is not "
                   "sound on this checkpoint (the differing pairs were corrected; unaudited pairs are not covered) — use "
                   "precision='split' / 'exact'")
            if self.strict_audit:
                raise RuntimeError(msg)
            import warnings

            warnings.warn(msg, RuntimeWarning)
        tot = float(max(B * n, 1))
        self.last_stats = {"layouts": B, "steps": n, "marked_layout_steps": st["marked"],
                           "exact_layout_steps": st["checked"], "mismatch_layout_steps": st["mismatch"],
                           "relaunched_layout_steps": st["relaunched"], "fast_passes": st["passes"],
                           "exact_fraction": st["checked"] / tot, "relaunched_fraction": st["relaunched"] / tot,
                           "audited_layout_steps": st["audited"], "audit_mismatch_layout_steps": st["audit_mismatch"],
                           "tie_rel": self.tie_rel, "tie_abs": self.tie_abs}
        return tokens, inter_all
