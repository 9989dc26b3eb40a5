"""CPU: `precision="auto"` says what it selected and why — on FAKE engines whose fp16 error is a knob.

HipMaskAndReplaceDiffusion.load_state_dict must (a) emit ONE INFO record on the `layout_dm_amd` logger naming the selected engine, the
measured fp16 logits error, the tolerance and the throughput class, (b) expose the same as `selection_report` (what
`python -m layout_dm_amd.check_checkpoint` prints), and (c) work for an engine built with max_batch < 4 (the verifier
check used a fixed probe batch of 4)."""
import json
import logging

import pytest
import torch

import layout_dm_amd.diffusion as D

S_ATTR = 5


class FakeEngine:
    """binding.Engine's surface as diffusion.py / verified.py use it; logits = a fixed function of (tokens, t) plus `noise` in the
    engines whose precision is 'fast'."""
    fast_noise = 0.0
    mixed_noise = 0.0
    hybrid_noise = 1.0            # (far outside unless a test sets it)
    mixed_available = True
    built = []

    def __init__(self, *, n_category, n_bin=32, max_elem=25, n_attr=5, n_step=100, precision="exact", max_batch=512,
                 q_type="constrained", **_k):
        self.q_type = q_type
        self.S, self.C, self.T = max_elem * n_attr, n_category + 4 * n_bin + 2, n_step
        self.n_attr, self.n_bin, self.n_category = n_attr, n_bin, n_category
        self.pad_id, self.mask_id = self.C - 2, self.C - 1
        self.max_batch, self.batch_round, self.precision = max_batch, 256, precision
        self.device = torch.device("cpu")
        if precision in ("mixed", "hybrid") and not FakeEngine.mixed_available:
            raise RuntimeError("ldm_create: precision mixed: only the reference backbone's geometry ...")
        FakeEngine.built.append((precision, max_batch))

    def load_state_dict(self, sd):
        pass

    def _tok(self, t):
        return torch.as_tensor(t).to(torch.int32).contiguous()

    def denoise_logits(self, tokens, t):
        tokens = self._tok(tokens)
        assert tokens.shape[0] <= self.max_batch, f"batch {tokens.shape[0]} outside [1, max_batch = {self.max_batch}]"
        g = torch.Generator().manual_seed(int(tokens.long().sum()) * 131 + int(t))
        base = torch.randn(tokens.shape[0], self.S, self.C, generator=g)
        noise = {"fast": FakeEngine.fast_noise, "mixed": FakeEngine.mixed_noise, "hybrid": FakeEngine.hybrid_noise}.get(self.precision, 0.0)
        if noise:
            base = base + noise * base.abs().max() * torch.sign(torch.randn(base.shape, generator=g))
        return base

    def sample_loop(self, tokens, t_model, t_post, cfg, cond=None, seed=0, first_layout=0, intermediates=False, use_graph=True,
                    lc_keep=None, relation=None):
        tokens = self._tok(tokens)
        inter = torch.stack([tokens.clone() for _ in t_model]) if intermediates else None
        return tokens, inter

    def set_tie_report(self, *a, **k):
        pass

    def describe(self):
        return {"precision": self.precision}

    def close(self):
        pass


@pytest.fixture()
def fake(monkeypatch):
    monkeypatch.setattr(D, "Engine", FakeEngine)
    import layout_dm_amd.verified as V

    monkeypatch.setattr(V, "Engine", FakeEngine, raising=False)
    FakeEngine.built = []
    FakeEngine.fast_noise, FakeEngine.mixed_noise, FakeEngine.hybrid_noise, FakeEngine.mixed_available = 0.0, 5e-3, 6e-3, True
    return FakeEngine


@pytest.mark.parametrize("noise,hybrid,mixed,want", [(1e-4, 6e-3, 5e-3, "fast_verified"), (5e-3, 6e-3, 4e-3, "split"), (5e-3, 6e-3, 2e-4, "mixed_verified"),
                                                     (5e-3, 4e-4, 2e-4, "hybrid_verified"), (5e-3, None, None, "split")])
def test_auto_logs_engine_error_tolerance_and_throughput_class(fake, caplog, noise, hybrid, mixed, want):
    """The ladder: fp16 engine inside the tolerance -> fast_verified (no other engine is even built); else the hybrid engine (attention path
    hi + lo activations x fp16 weights, FFN / head plain fp16) is built and measured -> hybrid_verified if inside; else the mixed engine
    (hi + lo activations x fp16 weights everywhere) -> mixed_verified; else — or where the library has no such kernels for the geometry —
    the reference-precision engine."""
    fake.fast_noise = noise
    fake.mixed_noise, fake.hybrid_noise, fake.mixed_available = (mixed or 0.0), (hybrid or 0.0), mixed is not None
    m = D.HipMaskAndReplaceDiffusion(n_category=25, precision="auto", max_batch=8)
    with caplog.at_level(logging.INFO, logger="layout_dm_amd"):
        m.load_state_dict({})
    recs = [r for r in caplog.records if r.name == "layout_dm_amd" and r.levelno == logging.INFO]
    assert len(recs) == 1, [r.getMessage() for r in recs]
    msg = recs[0].getMessage()
    assert m.selected_precision == want and f"'{want}'" in msg
    assert "0.001" in msg and ("inside" in msg if want == "fast_verified" else "OUTSIDE" in msg)
    assert D.THROUGHPUT_CLASS[want] in msg
    rep = m.selection_report
    assert rep["engine_selected"] == want and rep["tolerance"] == 1e-3 and rep["verifier"] == "split"
    assert abs(rep["fast_logits_err_rel"] - noise) < 0.2 * noise       # the record carries the MEASURED error
    assert f"{rep['fast_logits_err_rel']:.2e}" in msg
    built = [p for p, _ in FakeEngine.built]
    if want == "fast_verified":
        assert "mixed" not in built and "hybrid" not in built and "mixed_logits_err_rel" not in rep and "hybrid_logits_err_rel" not in rep
    elif mixed is None:
        assert "mixed_unavailable" in rep and "hybrid_unavailable" in rep and "no mixed engine" in msg and "no hybrid engine" in msg
    else:
        assert built.count("hybrid") == 1 and abs(rep["hybrid_logits_err_rel"] - hybrid) < 0.2 * hybrid and "hybrid engine" in msg
        if want == "hybrid_verified":
            assert "mixed" not in built and "mixed_logits_err_rel" not in rep           # the ladder stops at the first rung that holds
        else:
            assert built.count("mixed") == 1 and abs(rep["mixed_logits_err_rel"] - mixed) < 0.2 * mixed
            assert f"{rep['mixed_logits_err_rel']:.2e}" in msg and "mixed engine" in msg
        rung = want[:-len("_verified")] if want.endswith("_verified") else None
        assert (m.engine.precision == rung) == (rung is not None)
        assert (m.verified.fast is m.engine) == (rung is not None)
    json.dumps(rep, default=str)                                       # (check_checkpoint prints it)
    # a second checkpoint on the same object starts from the fp16 engine again
    fake.fast_noise = 1e-4
    m.load_state_dict({})
    assert m.selected_precision == "fast_verified" and m.engine.precision == "fast" and m.verified.fast is m.engine
    m.close()


@pytest.mark.parametrize("rung", ["mixed", "hybrid"])
def test_rung_verified_as_requested(fake, caplog, rung):
    fake.mixed_noise = fake.hybrid_noise = 3e-4
    m = D.HipMaskAndReplaceDiffusion(n_category=25, precision=rung + "_verified", max_batch=8)
    with caplog.at_level(logging.INFO, logger="layout_dm_amd"):
        m.load_state_dict({})
    msgs = [r.getMessage() for r in caplog.records if r.name == "layout_dm_amd"]
    assert len(msgs) == 1 and f"'{rung}_verified'" in msgs[0] and "as requested" in msgs[0]
    assert m.engine.precision == rung and m.verified.fast is m.engine and m.verified.exact.precision == "split"
    assert abs(m.selection_report[rung + "_logits_err_rel"] - 3e-4) < 1e-4 and abs(m.calibration["err_rel"] - 3e-4) < 1e-4


def test_explicit_precision_is_logged_too(fake, caplog):
    m = D.HipMaskAndReplaceDiffusion(n_category=25, precision="split", max_batch=8)
    with caplog.at_level(logging.INFO, logger="layout_dm_amd"):
        m.load_state_dict({})
    msgs = [r.getMessage() for r in caplog.records if r.name == "layout_dm_amd"]
    assert len(msgs) == 1 and "'split'" in msgs[0] and "as requested" in msgs[0]


@pytest.mark.parametrize("max_batch", [1, 2, 3])
def test_auto_with_max_batch_below_the_old_probe_batch(fake, max_batch):
    """_check_verifier probed with 4 layouts whatever max_batch was; ldm_denoise_logits rejects B > max_batch,
    so LayoutDM(max_batch=1) — precision='auto' is its default — raised inside load_state_dict."""
    fake.fast_noise = 1e-4
    m = D.HipMaskAndReplaceDiffusion(n_category=25, precision="auto", max_batch=max_batch)
    m.load_state_dict({})
    assert m.selected_precision == "fast_verified"
    assert all(mb <= max_batch for _, mb in FakeEngine.built)           # the small fp32 probe engine as well
    assert m.verifier_check["finite"]
