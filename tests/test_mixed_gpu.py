"""The MIXED and HYBRID numerics modes (LDM_PREC_MIXED_F16 / LDM_PREC_HYBRID_F16, r06): the split mode's launches with fp16-ONLY weights —
activations, q / k / v and the attention probabilities keep their hi + lo halves, the four weight GEMMs of a block and the head drop the W_lo
product (two matrix passes instead of three; kernels_lngemm.hip NPM / NPP, kernels_attnout.hip W2) — and, hybrid, with the FFN and the head in
plain fp16 on top (one pass: LayerNorm-2 / head-LayerNorm outputs and the hidden activations rounded once; the attention path keeps hi + lo).
tools/two_product_emulation.py predicted their logits errors on the CPU (mixed: fitted checkpoint 2.2e-4, init 3.0e-4, "mid" 5.7e-4; hybrid:
2.8e-4, 4.8e-4, 9.4e-4; "wide" percents for both); here the kernels are held to it:

  * logits against the float64 oracle on init / mid weights, against the reference's own logits on the fitted checkpoint — inside the
    north star's 1e-3, and an order of magnitude away from the split mode's (it must not silently BE the split mode or the fp16 mode);
  * bit-repeatable; the describe string names it; geometries without the two-product kernels are refused at create time;
  * `precision="auto"` on the fitted checkpoint (fp16 engine outside 1e-3) takes the middle rung, and its greedy loop — re-checked by the
    split engine — is the oracle's, token for token; `mixed_verified` greedy == the split engine's greedy on mid-trajectory states."""
import os

import numpy as np
import pytest
import torch

from oracle import restatement as R
from oracle import spec as SP
from oracle import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WEIGHTS = os.path.join(ROOT, "oracle", "_fit", "rico25_fitted.npz")
GOLDEN = os.path.join(ROOT, "tests", "golden", "rico25_fitted.npz")
pytestmark = pytest.mark.gpu
SPEC = SP.RICO25


@pytest.fixture(scope="module")
def cuda():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device (no CPU fallback exists)")
    return torch.device("cuda", 0)


def _rel(a, ref):
    return ((a.double() - ref.double()).abs().max() / ref.abs().max()).item()


def _states(seed=0, n=4):
    g = torch.Generator().manual_seed(seed)
    out = []
    for t in (50, 90, 5):
        tokens = torch.empty(n, SPEC.seq_len, dtype=torch.long)
        for a in range(SPEC.n_attr):
            ids = torch.as_tensor(SPEC.full_ids(a))
            tokens[:, a::SPEC.n_attr] = ids[torch.randint(0, len(ids) - 1, (n, SPEC.max_elem), generator=g)]
        tokens[torch.rand(n, SPEC.seq_len, generator=g) < t / 99] = SPEC.mask_id
        out.append((tokens, t))
    return out


@pytest.mark.parametrize("mode,point,lo,hi", [("mixed", "init", 5e-5, 6e-4), ("mixed", "mid", 1e-4, 1e-3), ("hybrid", "init", 2e-4, 8e-4),
                                              ("hybrid", "mid", 5e-4, 1.5e-3)])
def test_mixed_logits_against_the_float64_oracle(cuda, mode, point, lo, hi):
    from layout_dm_amd.binding import Engine

    sd = synth.synth_state_dict(SPEC, seed=0) if point == "init" else synth.trained_like_state_dict(SPEC, point, seed=3)
    W64 = R.as_torch_weights(sd, torch.float64)
    e = Engine(n_category=SPEC.n_category, precision=mode, max_batch=8)
    s = Engine(n_category=SPEC.n_category, precision="split", max_batch=8)
    e.load_state_dict(sd)
    s.load_state_dict(sd)
    assert e.describe()["precision"] == mode + "_f16" and s.describe()["precision"] == "split_f16"
    worst = 0.0
    for tokens, t in _states():
        ref = R.denoiser_logits(W64, SPEC, tokens, t, dtype=torch.float64)
        a = e.denoise_logits(tokens.int(), t)
        b = e.denoise_logits(tokens.int(), t)
        assert torch.equal(a, b), f"{mode}-mode logits are not bit-repeatable"
        em = _rel(a.cpu()[..., :SPEC.n_class], ref)
        es = _rel(s.denoise_logits(tokens.int(), t).cpu()[..., :SPEC.n_class], ref)
        assert es < 1e-5 < em, (point, t, em, es)          # the weights' lo halves are really gone — and only they
        worst = max(worst, em)
    print(f"[{mode}/{point}] max rel logits error vs the float64 oracle {worst:.3e} (tools/two_product_emulation.py: mixed init 3.0e-4, mid 5.7e-4; "
          f"hybrid init 4.8e-4, mid 9.4e-4)")
    assert lo < worst < hi, (point, worst)
    e.close()
    s.close()


def _fitted():
    if not os.path.exists(WEIGHTS):
        pytest.skip("oracle/_fit/rico25_fitted.npz absent (python -m oracle.make_trained_fixture)")
    w = np.load(WEIGHTS)
    return {k: w[k] for k in w.files}, np.load(GOLDEN)


def test_mixed_on_the_fitted_checkpoint_and_autos_middle_rung(cuda):
    from layout_dm_amd.binding import Engine
    from layout_dm_amd.diffusion import HipMaskAndReplaceDiffusion

    sd, g = _fitted()
    for mode, bound in (("mixed", 5e-4), ("hybrid", 6e-4)):
        e = Engine(n_category=SPEC.n_category, precision=mode, max_batch=8)
        e.load_state_dict(sd)
        worst = max(_rel(e.denoise_logits(torch.from_numpy(g[f"tokens_{int(t)}"].astype(np.int32)), int(t)).cpu(), torch.from_numpy(g[f"logits_{int(t)}"]))
                    for t in g["ts"])
        e.close()
        print(f"[{mode}/fitted] max rel logits error vs the reference {worst:.3e} (CPU emulation: mixed 2.2e-4, hybrid 2.8e-4; fp16 engine 1.2e-3, split 4e-6)")
        assert 2e-5 < worst < bound
    m = HipMaskAndReplaceDiffusion(n_category=SPEC.n_category, precision="auto", max_batch=8)
    m.load_state_dict(sd)
    rep = m.selection_report
    print(f"[auto on fitted] fp16 {rep['fast_logits_err_rel']:.3e}, hybrid {rep.get('hybrid_logits_err_rel')}, mixed {rep.get('mixed_logits_err_rel')} "
          f"-> '{m.selected_precision}'")
    if rep["fast_logits_err_rel"] <= 1e-3:
        assert m.selected_precision == "fast_verified"
    else:   # the ladder: the first rung inside the tolerance
        want = "hybrid" if rep["hybrid_logits_err_rel"] <= 1e-3 else "mixed" if rep["mixed_logits_err_rel"] <= 1e-3 else None
        assert want is not None and m.selected_precision == want + "_verified"
        assert m.engine.describe()["precision"] == want + "_f16" and m.verified.fast is m.engine
        assert m.verified.exact.describe()["precision"] == "split_f16"
    # greedy decoding: the verified pair answers with the reference's tokens on the fixture's 100-state trajectory ...
    before = torch.from_numpy(g["states_before"].astype(np.int32))
    bad = 0
    for i, t in enumerate(g["steps"]):
        out = m.verified.sample_step(before[i], int(t), step=i).cpu()
        bad += int((out.numpy() != g["greedy_next"][i]).sum())
    assert bad == 0, bad
    # ... and a greedy loop of the default path == the oracle's
    cfg = {"name": "deterministic", "num_timesteps": 10}
    assert torch.equal(m.sample(batch_size=2, sampling_cfg=cfg), R.sample_loop(R.as_torch_weights(sd), SPEC, 2, cfg))
    # stochastic sampling runs on the selected engine and leaves no [MASK]
    out = m.sample(batch_size=8, sampling_cfg={"name": "random", "temperature": 1.0, "num_timesteps": 100}, seed=3)
    assert out.shape == (8, SPEC.seq_len) and bool((out != SPEC.mask_id).all())
    m.close()


@pytest.mark.parametrize("mode", ["mixed", "hybrid"])
def test_mixed_verified_greedy_equals_the_split_engines_on_mid_trajectory_states(cuda, mode):
    """Greedy loops started where a stochastic run stands after 50 of 100 steps (near-ties are NOT confined to the last steps there)."""
    from layout_dm_amd.diffusion import HipMaskAndReplaceDiffusion, timestep_schedule

    sd = synth.trained_like_state_dict(SPEC, "mid", seed=3)
    B = 64
    mv = HipMaskAndReplaceDiffusion(n_category=SPEC.n_category, precision=mode + "_verified", max_batch=B)
    mv.load_state_dict(sd)
    assert mv.selection_report[mode + "_logits_err_rel"] < 2e-3
    tm, tp = timestep_schedule(100, 100)
    split = mv.verified.exact
    tok = torch.full((B, SPEC.seq_len), SPEC.mask_id, dtype=torch.int32, device=split.device)
    mid, _ = split.sample_loop(tok, tm[:50], tp[:50], {"name": "random", "temperature": 1.0}, seed=11)
    a, _ = mv.verified.sample_loop(mid.clone(), tm[50:], tp[50:])
    b, _ = split.sample_loop(mid.clone(), tm[50:], tp[50:], {"name": "deterministic"})
    st = mv.verified.last_stats
    print(f"[{mode}_verified] {B} layouts x 50 greedy steps from mid-trajectory states: marked {st['marked_layout_steps']}, re-checked "
          f"{st['exact_layout_steps']}, corrected {st['mismatch_layout_steps']}, audit mismatches {st['audit_mismatch_layout_steps']}")
    assert torch.equal(a, b) and st["audit_mismatch_layout_steps"] == 0
    mv.close()


def test_mixed_is_refused_where_its_kernels_do_not_exist(cuda):
    from layout_dm_amd.binding import Engine

    for mode in ("mixed", "hybrid"):
        with pytest.raises(RuntimeError, match="precision mixed / hybrid"):
            Engine(n_category=5, d_model=256, n_head=8, d_ff=1024, precision=mode, max_batch=4)


def test_hybrid_fused_ffn_against_its_two_launch_form_and_the_emulation(cuda, tmp_path):
    """The hybrid mode's FFN ships as ONE plain-fp16 launch (kernels_ffn16.hip: the fast mode's chunk stream as a row kernel).  Its first form —
    linear1 writing plain-fp16 panels, linear2 as the one-product GEMM prologue of the next launch (kernels_lngemm.hip OUT = 3 / NPP = 1, kept behind
    LDM_DEV=1 LDM_HYB_FFN=0) — rounds the same operands at the same places.  They can NOT be asked to agree digit for digit: a plain-fp16 activation
    format is chaotic at its own error level — a 1e-7 relative perturbation anywhere upstream (another fp32 summation order is enough) flips one
    rounding in 10^4, each flip moves a residual row by ~1e-5, which flips 2 % of the roundings behind it, and two blocks later the rounding pattern
    is a different draw (tools/two_product_emulation.py `jitter`: the CPU emulation moves by 7e-4 under such a jitter; the mixed format, whose
    activation rounding unit is 2^-22, by 3e-5 — and the mixed ENGINE matches its emulation to 1e-6).  What is checked: each form's logits error
    against the float64 oracle is the emulation's error LEVEL (within a factor 1.6, per timestep), and the forms differ by no more than their errors
    add up to.  One process per form: the knob is read at create."""
    import subprocess
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import two_product_emulation as E

    code = f'''
import sys
sys.path.insert(0, {ROOT!r})
import numpy as np, torch
from oracle import spec as SP, synth
from layout_dm_amd.binding import Engine
spec = SP.RICO25
sd = synth.trained_like_state_dict(spec, "mid", seed=3)
g = torch.Generator().manual_seed(9)
tokens = torch.randint(0, spec.n_class, (4, spec.seq_len), generator=g).int()
e = Engine(n_category=spec.n_category, precision="hybrid", max_batch=8)
e.load_state_dict(sd)
assert ("ffn_fused_fp16" in e.describe()["kernels"]) == (sys.argv[2] == "fused"), e.describe()
np.save(sys.argv[1], torch.stack([e.denoise_logits(tokens, t).cpu()[..., :spec.n_class] for t in (90, 40, 3)]).numpy())
'''
    outs = {}
    for form, env in (("fused", {}), ("two_launch", {"LDM_DEV": "1", "LDM_HYB_FFN": "0"})):
        path = str(tmp_path / f"{form}.npy")
        p = subprocess.run([sys.executable, "-c", code, path, form], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                           timeout=300)
        assert p.returncode == 0, p.stdout[-2000:]
        outs[form] = torch.from_numpy(np.load(path)).double()
    sd = synth.trained_like_state_dict(SPEC, "mid", seed=3)
    W, W64 = R.as_torch_weights(sd), R.as_torch_weights(sd, torch.float64)
    tokens = torch.randint(0, SPEC.n_class, (4, SPEC.seq_len), generator=torch.Generator().manual_seed(9))
    f = {s: (E.h if s in dict(E.FORMATS)["hybrid: weights + ln2, hid, hln fp16"] else E.h2) for s in E.SITES}
    for i, t in enumerate((90, 40, 3)):
        ref = R.denoiser_logits(W64, SPEC, tokens, t, dtype=torch.float64)
        m = ref.abs().max()
        emu = ((E.fwd(W, tokens, t, f).double() - ref).abs().max() / m).item()
        ea, eb = (((outs[k][i] - ref).abs().max() / m).item() for k in ("fused", "two_launch"))
        d = ((outs["fused"][i] - outs["two_launch"][i]).abs().max() / m).item()
        print(f"[hybrid, t = {t}] error vs float64: fused FFN {ea:.2e}, two-launch form {eb:.2e}, CPU emulation of the format {emu:.2e}; fused vs two-launch {d:.2e}")
        assert emu / 1.6 <= ea <= 1.6 * emu and emu / 1.6 <= eb <= 1.6 * emu, (t, ea, eb, emu)
        assert d <= ea + eb, (t, d, ea, eb)
