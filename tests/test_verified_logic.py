"""CPU: the splice / re-launch logic of `fast_verified` (layout_dm_amd/verified.py) on FAKE engines.

The exact engine is a deterministic token map next = f(tokens, t); the fast engine is the same map with errors injected
at chosen (step, layout) pairs — always marked, as a sound near-tie report guarantees — plus false-positive marks.  The
verified loop must return the exact engine's trajectory (every intermediate state), check exactly the marked pairs that
are still on a valid trajectory, re-launch only the layouts whose tokens differed, and account for all of it."""
import numpy as np
import pytest
import torch

from layout_dm_amd.verified import VerifiedGreedy

S, C = 20, 50


def exact_next(tokens: torch.Tensor, t: int) -> torch.Tensor:
    """A 'model': the next state depends on the whole layout (like attention does) and on t."""
    mix = tokens.long().sum(dim=1, keepdim=True)
    return ((tokens.long() * 7 + mix * 3 + t * 11 + torch.arange(S)[None] * 5) % C).int()


class FakeEngine:
    def __init__(self, T, errors=None, marks=None):
        self.S, self.C, self.T = S, C, T
        self.device = torch.device("cpu")
        self.errors = errors or set()      # {(t_model, layout_key)}: the fast engine's wrong decisions
        self.marks = marks or set()        # extra marked pairs (false positives)
        self.calls = []
        self._flags = None
        self._keys = None                  # layout identity = its conditioning row, carried through sub-batches

    def _tok(self, t):
        return torch.as_tensor(t).to(torch.int32).contiguous()

    def set_tie_report(self, tie_rel, tie_abs=0.0):
        self.tie = (tie_rel, tie_abs)

    def _step(self, tokens, t, keys):
        out = exact_next(tokens, t)
        flags = torch.zeros(tokens.shape[0], dtype=torch.uint8)
        for j, k in enumerate(keys.tolist()):
            if (t, k) in self.errors:
                out[j, 3] = (out[j, 3] + 1) % C
                flags[j] = 1
            if (t, k) in self.marks:
                flags[j] = 1
        return out, flags

    def sample_step(self, tokens, t_model, cfg, t_post=None, cond=None, seed=0, first_layout=0, step=0, relation=None):
        keys = cond["key"]
        out, flags = self._step(self._tok(tokens), int(t_model), keys)
        self._flags = flags[None]
        self.calls.append(("step", int(t_model), tokens.shape[0]))
        return out

    def sample_loop(self, tokens, t_model, t_post, cfg, cond=None, seed=0, first_layout=0, intermediates=False,
                    use_graph=True, lc_keep=None, relation=None):
        tokens = self._tok(tokens)
        keys = cond["key"]
        inter, flags = [], []
        cur = tokens
        for t in t_model:
            cur, f = self._step(cur, int(t), keys)
            inter.append(cur.clone())
            flags.append(f)
        self._flags = torch.stack(flags)
        self.calls.append(("loop", len(t_model), tokens.shape[0]))
        tokens.copy_(cur)
        return tokens, (torch.stack(inter) if intermediates else None)

    def tie_flags(self, n_steps, B):
        assert self._flags.shape == (n_steps, B)
        return self._flags.clone()


def reference_run(B, steps):
    tok = (torch.arange(B * S).view(B, S) % C).int()
    inter = []
    for t in steps:
        tok = exact_next(tok, t)
        inter.append(tok.clone())
    return torch.stack(inter)


@pytest.mark.parametrize("case", ["no_errors", "errors_spread", "errors_cascade_and_last_step", "everything_marked"])
def test_verified_loop_reproduces_the_exact_trajectory(case):
    B, T = 12, 30
    steps = list(range(T - 1, -1, -1))
    errors, marks = set(), set()
    if case == "errors_spread":
        errors = {(25, 2), (17, 5), (9, 2), (9, 7)}          # layout 2 goes wrong twice: the second time in its re-launch
        marks = {(20, 0), (20, 1), (3, 11)}
    elif case == "errors_cascade_and_last_step":
        errors = {(0, 4), (1, 4), (2, 4), (0, 9), (29, 3)}   # the first step, and three consecutive last steps
    elif case == "everything_marked":
        marks = {(t, k) for t in steps for k in range(B)}
        errors = {(12, 6)}
    fast, exact = FakeEngine(T, errors, marks), FakeEngine(T)
    vg = VerifiedGreedy(fast, exact, tie_rel=6e-3, tie_abs=0.01, audit=0.0)   # (the splice logic alone; audits below)
    start = (torch.arange(B * S).view(B, S) % C).int()
    cond = {"key": torch.arange(B), "type": "c"}             # the layout identity rides along like a cond tensor
    out, inter = vg.sample_loop(start.clone(), steps, steps, cond=cond, intermediates=True)
    ref = reference_run(B, steps)
    assert torch.equal(inter, ref) and torch.equal(out, ref[-1])
    st = vg.last_stats
    assert st["mismatch_layout_steps"] == len(errors)
    n_relaunch_passes = st["fast_passes"] - 1
    assert (n_relaunch_passes == 0) == (len([e for e in errors if e[0] != 0]) == 0 or case == "no_errors")
    # every exact call is a single step on a sub-batch; their total is the number of checked pairs
    assert sum(n for kind, _, n in exact.calls) == st["exact_layout_steps"] and all(k == "step" for k, _, _ in exact.calls)
    # only layouts with an error are re-launched, from the step after it, for the remaining steps
    relaunched = sorted((n, b) for kind, n, b in fast.calls[1:])
    if case == "errors_spread":
        # t = 25 -> 25 remaining steps for layout 2; t = 17 -> 17 for layout 5; t = 9: layout 7 from the first pass, layout 2
        # from its re-launch: 9 steps each (grouped per pass)
        assert st["relaunched_layout_steps"] == 25 + 17 + 9 + 9
        assert sorted(b for _, b in relaunched) == [1, 1, 1, 1]
    if case == "errors_cascade_and_last_step":
        # an error at the LAST step (t = 0) is spliced without a re-launch; t = 29 (first step) re-launches 29 steps
        assert st["relaunched_layout_steps"] == 29 + 2 + 1 and st["mismatch_layout_steps"] == 5
    if case == "everything_marked":
        assert st["exact_fraction"] >= 1.0 and st["mismatch_layout_steps"] == 1
    assert st["exact_layout_steps"] <= st["marked_layout_steps"]


def test_verified_step_and_audit():
    B, T = 8, 10
    fast, exact = FakeEngine(T, errors={(4, 1)}, marks={(4, 6)}), FakeEngine(T)
    vg = VerifiedGreedy(fast, exact, audit=0.5)
    start = (torch.arange(B * S).view(B, S) % C).int()
    cond = {"key": torch.arange(B), "type": "c"}
    out = vg.sample_step(start, 4, cond=cond)
    assert torch.equal(out, exact_next(start, 4)) and vg.last_stats["marked_layout_steps"] == 2
    assert vg.last_stats["mismatch_layout_steps"] == 1
    steps = list(range(T - 1, -1, -1))
    out, inter = vg.sample_loop(start.clone(), steps, steps, cond=cond, intermediates=True)
    assert torch.equal(inter, reference_run(B, steps))
    st = vg.last_stats
    # the audit re-checks unmarked pairs at random; the report is sound here, so none of them differs
    assert st["audited_layout_steps"] > 0 and st["audit_mismatch_layout_steps"] == 0
    # an UNSOUND report (an error that is not marked) is what the audit exists to catch
    class Unsound(FakeEngine):
        def _step(self, tokens, t, keys):
            out, flags = super()._step(tokens, t, keys)
            if t == 6:
                out[0, 0] = (out[0, 0] + 1) % C     # wrong, and NOT marked
            return out, flags
    vg2 = VerifiedGreedy(Unsound(T), FakeEngine(T), audit=1.0)
    with pytest.warns(RuntimeWarning, match="not\\s+sound on this checkpoint"):
        out, _ = vg2.sample_loop(start.clone(), steps, steps, cond=cond)
    assert vg2.last_stats["audit_mismatch_layout_steps"] >= 1 and vg2.audit_mismatch_total >= 1
    assert torch.equal(out, reference_run(B, steps)[-1])    # (with audit = 1 every pair is checked, so it is also repaired)
    with pytest.raises(RuntimeError, match="audited UNMARKED"):
        VerifiedGreedy(Unsound(T), FakeEngine(T), audit=1.0, strict_audit=True).sample_loop(start.clone(), steps, steps, cond=cond)


def test_default_audit_is_on_and_clustered():
    """r05: a small audit runs by default, concentrated on a few steps — each audited step is one
    reference-precision launch sequence, whatever the number of layouts in it."""
    B, T = 64, 40
    fast, exact = FakeEngine(T), FakeEngine(T)
    vg = VerifiedGreedy(fast, exact)
    assert vg.audit == VerifiedGreedy.DEFAULT_AUDIT > 0
    steps = list(range(T - 1, -1, -1))
    start = (torch.arange(B * S).view(B, S) % C).int()
    out, _ = vg.sample_loop(start.clone(), steps, steps, cond={"key": torch.arange(B), "type": "c"})
    assert torch.equal(out, reference_run(B, steps)[-1])
    st = vg.last_stats
    assert 0 < st["audited_layout_steps"] <= max(VerifiedGreedy.AUDIT_STEPS, round(2 * vg.audit * B * T))
    assert len(exact.calls) <= VerifiedGreedy.AUDIT_STEPS and st["audit_mismatch_layout_steps"] == 0


def test_relation_is_refused():
    vg = VerifiedGreedy(FakeEngine(5), FakeEngine(5))
    with pytest.raises(NotImplementedError):
        vg.sample_loop(torch.zeros((4, S), dtype=torch.int32), [4, 3], [4, 3], cond={"type": "relation", "key": torch.arange(4)}, )
