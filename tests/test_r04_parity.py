"""GPU parity OFF the easy point: the HIP path through the C-ABI against fixtures the REAL reference
produced on

  (a) "trained-like" weight distributions (oracle/synth.py TRAINED_LIKE: sigma 0.02 / 0.06 / 0.15, LayerNorm gains
      1 +- 0.5, outlier channels x8, AdaLN embeddings x5) — tests/golden/rico25_trained_like.npz;
  (b) BASELINE config 5's shape: a T = 200 model sampled with cond=refinement and cond=relation —
      tests/golden/rico25_config5_T200.npz;

and `fast_verified` where it hurts: free-running greedy loops at B = 512 from mid-trajectory states.

Tolerances, each derived from something measured rather than picked:
  * exact mode logits: max(2e-5, 3 x the reference's OWN float32 noise floor at that point) — the fixture stores the
    reference's float32 forward against its float64 forward; two float32 evaluations with different summation orders
    can each sit one floor from the truth (at sigma = 0.15 the floor is 1.2e-4: "2e-5" is not a meaningful bar there);
  * exact mode greedy tokens: bit-exact, every point, every step;
  * fast mode logits: the north star's 1e-3 holds on the init-like point only; the measured error of every point is
    printed and bounded by an envelope (FAST_ENVELOPE), and `precision="auto"` must refuse the fast engine exactly
    where its measured error exceeds 1e-3;
  * fast mode greedy tokens: may differ from the reference only where the reference's top-2 margin is below 6 x the
    calibrated absolute logits error (the bound the near-tie report is built on);
  * fast_verified: bit-exact, every point, every step, with the calibrated threshold.
"""
import dataclasses
import os

import numpy as np
import pytest
import torch

from oracle import restatement as R
from oracle import spec as SP
from oracle import synth

pytestmark = pytest.mark.gpu

POINTS = list(synth.TRAINED_LIKE)
# measured r04 (profiles/r04_call1_*): init 4.4e-4, mid 1.27e-3, wide 8.5e-2 against the reference (1.5e-1 on the probe
# states; max |attention score| > 200 there: saturated softmax rows amplify the fp16 rounding of q and k, and the fast mode
# is simply not usable — `auto` runs that checkpoint in the exact mode); the envelope is ~2x the measurement
FAST_ENVELOPE = {"init": 1e-3, "mid": 2.5e-3, "wide": 2.5e-1}
GREEDY = {"name": "deterministic"}


def _rel(a, ref):
    return ((a - ref).abs().max() / ref.abs().max()).item()


@pytest.fixture(scope="module")
def cuda():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device (no CPU fallback exists)")
    return torch.device("cuda", 0)


_ENG = {}


def _engine(spec, sd_key, sd_fn, precision, max_batch=8):
    from layout_dm_amd.binding import Engine

    key = (spec.name, sd_key, precision, max_batch)
    if key not in _ENG:
        e = Engine(n_category=spec.n_category, n_bin=spec.n_bin, max_elem=spec.max_elem, d_model=spec.d_model,
                   n_head=spec.n_head, d_ff=spec.d_ff, n_layer=spec.n_layer, n_step=spec.n_step, precision=precision,
                   max_batch=max_batch)
        e.load_state_dict(sd_fn())
        _ENG[key] = e
    return _ENG[key]


def point_engine(point, precision, max_batch=8):
    spec = SP.RICO25
    return _engine(spec, point, lambda: synth.trained_like_state_dict(spec, point, seed=2), precision, max_batch)


def _sub(g, prefix):
    return {k[len(prefix):]: g[k] for k in g.files if k.startswith(prefix)}


def _traj(e, g, cond=None, vg=None, relation=None):
    """teacher-forced greedy steps over a reference trajectory -> (#tokens differing, #tokens, worst reference margin
    among the differing ones, marked layout-steps)"""
    steps = g["steps"]
    before = torch.from_numpy(g["states_before"].astype(np.int32))
    ref_next = torch.from_numpy(g["greedy_next"].astype(np.int32))
    margin = torch.from_numpy(g["greedy_margin"])
    bad, worst, marked = 0, 0.0, 0
    for i, t in enumerate(steps):
        if vg is not None:
            out = vg.sample_step(before[i], int(t), cond=cond, step=i).cpu()
            marked += vg.last_stats["marked_layout_steps"]
        else:
            out = e.sample_step(before[i], int(t), GREEDY, cond=cond, step=i, relation=relation).cpu()
        mism = out != ref_next[i]
        if mism.any():
            bad += int(mism.sum())
            worst = max(worst, margin[i][mism].max().item())
    return bad, ref_next.numel(), worst, marked


# ----------------------------------------------------------------------------- (a) trained-like weight distributions
@pytest.mark.parametrize("point", POINTS)
def test_trained_like_logits_exact_mode(cuda, golden_dir, point):
    g = np.load(os.path.join(golden_dir, "rico25_trained_like.npz"))
    floor = float(g[f"{point}_f32_noise_floor"])
    e = point_engine(point, "exact")
    worst = 0.0
    for t in g["ts"]:
        t = int(t)
        tokens = torch.from_numpy(g[f"{point}_tokens_{t}"].astype(np.int32))
        ref = torch.from_numpy(g[f"{point}_logits_{t}"])
        worst = max(worst, _rel(e.denoise_logits(tokens, t).cpu(), ref))
        post = e.posterior(ref, tokens, t).cpu()
        ref_post = torch.from_numpy(g[f"{point}_post_{t}"])
        assert (post - ref_post).abs().max().item() <= 2e-4, (point, t)
        assert torch.equal(post.argmax(1), ref_post.argmax(1))
    print(f"[trained-like/{point}/exact] max rel logits error vs the reference {worst:.3e} "
          f"(the reference's own float32 noise floor: {floor:.3e}; max |logit| "
          f"{max(float(np.abs(g[f'{point}_logits_{int(t)}']).max()) for t in g['ts']):.2f})")
    assert worst <= max(2e-5, 3 * floor)


@pytest.mark.parametrize("precision", ["exact", "split", "fast"])
def test_trained_like_publaynet_mid(cuda, golden_dir, precision):
    """The other vocabulary off the init distribution: PubLayNet (5 categories, C = 135) on the "mid" weights, fixture from the real
    reference (tests/golden/publaynet_trained_like.npz).  Reference-precision modes: logits within max(2e-5, 3 x the reference's own
    float32 noise floor), posterior, and the reference's greedy tokens on all 100 states of its stochastic trajectory; fast
    mode: logits inside the envelope of the mid point, greedy tokens differ only where the reference's own top-2 margin is tiny."""
    spec = SP.SPECS["publaynet"]
    g = np.load(os.path.join(golden_dir, "publaynet_trained_like.npz"))
    e = _engine(spec, "mid", lambda: synth.trained_like_state_dict(spec, "mid", seed=2), precision, 8)
    floor = float(g["mid_f32_noise_floor"])
    worst = 0.0
    for t in g["ts"]:
        t = int(t)
        tokens = torch.from_numpy(g[f"mid_tokens_{t}"].astype(np.int32))
        ref = torch.from_numpy(g[f"mid_logits_{t}"])
        worst = max(worst, _rel(e.denoise_logits(tokens, t).cpu(), ref))
        if precision != "fast":
            post = e.posterior(ref, tokens, t).cpu()
            ref_post = torch.from_numpy(g[f"mid_post_{t}"])
            assert (post - ref_post).abs().max().item() <= 2e-4, t
            assert torch.equal(post.argmax(1), ref_post.argmax(1))
    bad, n, margin, _ = _traj(e, _sub(g, "mid_"))
    print(f"[trained-like/publaynet mid/{precision}] max rel logits error vs the reference {worst:.3e} (noise floor {floor:.3e}); "
          f"greedy tokens differing {bad}/{n}" + (f" (largest reference margin among them {margin:.3e})" if bad else ""))
    if precision == "fast":
        assert worst <= FAST_ENVELOPE["mid"]
        assert bad == 0 or margin < 2e-2, (bad, margin)
    else:
        assert worst <= max(2e-5, 3 * floor)
        assert bad == 0


def test_default_path_publaynet_mid(cuda, golden_dir):
    """The other vocabulary: what the default (precision="auto") runs on the PubLayNet "mid" checkpoint is inside 1e-3 of the
    reference (the plain fp16 engine measures 1.5e-3 there and is refused)."""
    from layout_dm_amd.diffusion import HipMaskAndReplaceDiffusion

    spec = SP.SPECS["publaynet"]
    g = np.load(os.path.join(golden_dir, "publaynet_trained_like.npz"))
    m = HipMaskAndReplaceDiffusion(n_category=spec.n_category, precision="auto", max_batch=8)
    m.load_state_dict(synth.trained_like_state_dict(spec, "mid", seed=2))
    worst = 0.0
    for t in g["ts"]:
        t = int(t)
        tokens = torch.from_numpy(g[f"mid_tokens_{t}"].astype(np.int32))
        worst = max(worst, _rel(m.engine.denoise_logits(tokens, t).cpu(), torch.from_numpy(g[f"mid_logits_{t}"])))
    print(f"[trained-like/publaynet mid/default = auto -> {m.selected_precision}] fast engine measured at load "
          f"{m.calibration['err_rel']:.3e}; selected engine vs the reference {worst:.3e}")
    assert worst <= 1e-3
    assert (m.selected_precision == "fast_verified") == (m.calibration["err_rel"] <= 1e-3)
    m.verified.fast.close()
    m.verified.exact.close()


@pytest.mark.parametrize("point", POINTS)
def test_trained_like_logits_fast_mode_and_auto_selection(cuda, golden_dir, point):
    """The fast mode's measured error at every point, against the reference; `precision="auto"` keeps the fp16 engine
    exactly where its error measured at load time (fast vs exact engine, probe states) is inside 1e-3."""
    from layout_dm_amd.diffusion import HipMaskAndReplaceDiffusion

    g = np.load(os.path.join(golden_dir, "rico25_trained_like.npz"))
    e = point_engine(point, "fast")
    worst = 0.0
    for t in g["ts"]:
        t = int(t)
        tokens = torch.from_numpy(g[f"{point}_tokens_{t}"].astype(np.int32))
        worst = max(worst, _rel(e.denoise_logits(tokens, t).cpu(), torch.from_numpy(g[f"{point}_logits_{t}"])))
    spec = SP.RICO25
    m = HipMaskAndReplaceDiffusion(n_category=spec.n_category, precision="auto", max_batch=8)
    m.load_state_dict(synth.trained_like_state_dict(spec, point, seed=2))
    cal = m.calibration
    print(f"[trained-like/{point}/fast] max rel logits error vs the reference {worst:.3e}; calibration at load: "
          f"err_rel {cal['err_rel']:.3e} err_abs {cal['err_abs']:.3e} absmax {cal['absmax']:.2f} -> auto selects "
          f"{m.selected_precision}")
    assert worst <= FAST_ENVELOPE[point]                        # (a sanity envelope around the MEASUREMENT printed above)
    if point == "init":
        assert worst <= 1e-3                                    # the north star's bound where the fp16 engine claims it
    # r05: what the DEFAULT path runs — precision="auto" is LayoutDM's default — is inside the north
    # star's 1e-3 against the reference at EVERY point, whichever engine the load-time measurement selected
    dflt = 0.0
    for t in g["ts"]:
        t = int(t)
        tokens = torch.from_numpy(g[f"{point}_tokens_{t}"].astype(np.int32))
        dflt = max(dflt, _rel(m.engine.denoise_logits(tokens, t).cpu(), torch.from_numpy(g[f"{point}_logits_{t}"])))
    print(f"[trained-like/{point}/default = auto -> {m.selected_precision}] max rel logits error vs the reference {dflt:.3e}; "
          f"verifier check {m.verifier_check}")
    assert dflt <= 1e-3, (point, m.selected_precision, dflt)
    # the probe's verdict agrees with the error against the reference (within the spread between probe and fixture states)
    assert 0.4 * worst <= cal["err_rel"] <= 2.5 * worst, (worst, cal)
    # (r06: the mixed engine — hi + lo activations x fp16 weights — is auto's rung between the two: tests/test_mixed_gpu.py)
    mixed = m.selection_report.get("mixed_logits_err_rel")
    hybrid = m.selection_report.get("hybrid_logits_err_rel")
    assert m.selected_precision == ("fast_verified" if cal["err_rel"] <= 1e-3 else "hybrid_verified" if hybrid is not None and hybrid <= 1e-3
                                    else "mixed_verified" if mixed is not None and mixed <= 1e-3 else "split")
    if m.selected_precision in ("mixed_verified", "hybrid_verified"):
        assert m.engine.describe()["precision"] == m.selected_precision[:-len("_verified")] + "_f16" and m.verified.exact.describe()["precision"] == "split_f16"
    if point == "wide":
        assert m.selected_precision == "split" and m.engine is m.verified.exact   # (the reference-precision engine: split)
        out = m.sample(batch_size=2, sampling_cfg={"name": "deterministic", "num_timesteps": 5})
        ref = R.sample_loop(R.as_torch_weights(synth.trained_like_state_dict(spec, point, seed=2)), spec, 2,
                            {"name": "deterministic", "num_timesteps": 5})
        assert torch.equal(out, ref)
    if point == "init":
        assert m.selected_precision == "fast_verified"
    m.close()


@pytest.mark.parametrize("point", POINTS)
def test_trained_like_greedy_trajectory(cuda, golden_dir, point):
    """All 100 states of a stochastic reference trajectory per point: exact mode bit-exact; fast mode differs only inside
    the calibrated band; fast_verified (calibrated near-tie report + exact re-check) bit-exact."""
    from layout_dm_amd.verified import LEAD_LIPSCHITZ, VerifiedGreedy

    g = _sub(np.load(os.path.join(golden_dir, "rico25_trained_like.npz")), point + "_")
    ex, fa = point_engine(point, "exact"), point_engine(point, "fast")
    bad, n, worst, _ = _traj(ex, g)
    print(f"[trained-like/{point}/exact] greedy tokens differing from the reference: {bad}/{n}")
    assert bad == 0
    vg = VerifiedGreedy(fa, ex)
    cal = vg.calibrate()
    fa.set_tie_report(0.0, 0.0)
    bad, n, worst, _ = _traj(fa, g)
    print(f"[trained-like/{point}/fast] greedy tokens differing from the reference: {bad}/{n}"
          + (f" (largest reference margin among them {worst:.3e}; calibrated band {cal['tie_abs']:.3e} = "
             f"{LEAD_LIPSCHITZ:.0f} x {vg.safety:.0f} x {cal['err_abs']:.3e})" if bad else ""))
    assert bad == 0 or worst < cal["tie_abs"], (bad, worst, cal)
    bad, n, _, marked = _traj(None, g, vg=vg)
    n_ls = g["states_before"].shape[0] * g["states_before"].shape[1]
    print(f"[trained-like/{point}/fast_verified] greedy tokens differing from the reference: {bad}/{n}; layout-steps "
          f"re-checked in the exact mode {marked}/{n_ls}")
    assert bad == 0
    fa.set_tie_report(0.0, 0.0)


@pytest.mark.parametrize("point", POINTS)
def test_tie_report_is_sound_on_trained_like_points(cuda, point):
    """B = 128 on states a stochastic run of THAT checkpoint visits: every layout whose fast greedy tokens differ from the
    exact mode's was marked (calibrated threshold), and the verified loop from mid-trajectory states equals the exact
    engine's greedy loop, token for token, at every intermediate step."""
    from layout_dm_amd.verified import VerifiedGreedy

    spec, B = SP.RICO25, 128
    fa, ex = point_engine(point, "fast", B), point_engine(point, "exact", B)
    vg = VerifiedGreedy(fa, ex, audit=0.02)
    cal = vg.calibrate()
    steps = R.timestep_list(spec.n_step, 100)
    tok = torch.full((B, spec.seq_len), spec.mask_id, dtype=torch.int32, device=cuda)
    _, inter = ex.sample_loop(tok, steps, steps, {"name": "random", "temperature": 1.0}, seed=5, intermediates=True)
    inter = inter.clone()
    n_marked = n_diff = 0
    for i in (9, 49, 89, 96, 97, 98, 99):
        before, t = inter[i - 1], steps[i]
        e_out = ex.sample_step(before, t, GREEDY, step=i)
        fa.set_tie_report(vg.tie_rel, vg.tie_abs)
        f_out = fa.sample_step(before, t, GREEDY, step=i)
        flags = fa.tie_flags(1, B)[0].bool()
        diff = (f_out != e_out).any(dim=1)
        assert not (diff & ~flags).any(), f"{point} step {i}: a layout with differing tokens was not marked"
        n_marked += int(flags.sum())
        n_diff += int(diff.sum())
    print(f"[tie report/{point}, B={B}, 7 steps] calibration err_abs {cal['err_abs']:.3e} -> tie_abs {cal['tie_abs']:.3e}; "
          f"marked layout-steps {n_marked}/{7 * B}; layouts whose fast tokens differ from the exact mode's: {n_diff}")
    for i0 in (50, 80):                                          # free-running greedy from a mid-trajectory state
        start = inter[i0 - 1].clone()
        want, want_inter = ex.sample_loop(start.clone(), steps[i0:], steps[i0:], GREEDY, intermediates=True, use_graph=False)
        got, got_inter = vg.sample_loop(start.clone(), steps[i0:], steps[i0:], intermediates=True)
        st = vg.last_stats
        print(f"[fast_verified loop/{point} from step {i0}] {st}")
        assert torch.equal(got_inter, want_inter) and torch.equal(got, want)
        assert st["audit_mismatch_layout_steps"] == 0, st      # unmarked layout-steps re-checked at random: all equal
    fa.set_tie_report(0.0, 0.0)


# ----------------------------------------------------------------------------- (b) config 5's shape: T = 200
def _config5(golden_dir):
    spec = dataclasses.replace(SP.RICO25, name="rico25_t200", n_step=200)
    g = np.load(os.path.join(golden_dir, "rico25_config5_T200.npz"))
    return spec, g, (lambda: synth.synth_state_dict(spec, seed=1, perturb=True))


def _refinement_cond(sub, spec):
    table = torch.from_numpy(sub["weak_table"])
    seq_orig = torch.from_numpy(sub["seq_orig"].astype(np.int64))
    return {"seq": sub["cond_seq"].astype(np.int64), "mask": sub["cond_mask"], "type": "refinement",
            "weak_logits": table[seq_orig].permute(0, 2, 1).contiguous()}


@pytest.mark.parametrize("precision", ["exact", "fast"])
def test_config5_T200_refinement_teacher_forced(cuda, golden_dir, precision):
    from test_hip_parity import MARGIN_BOUND, MISMATCH_COUNT_BOUND

    spec, g, sd = _config5(golden_dir)
    sub = _sub(g, "ref_")
    e = _engine(spec, "t200", sd, precision)
    bad, n, worst, _ = _traj(e, sub, cond=_refinement_cond(sub, spec))
    print(f"[config 5 shape / refinement T=200 / {precision}] greedy tokens differing from the reference: {bad}/{n}"
          + (f" (largest reference margin among them {worst:.3e})" if bad else ""))
    if precision == "exact":
        assert bad == 0
    else:
        assert bad == 0 or worst < MARGIN_BOUND["fast"]
        assert bad <= MISMATCH_COUNT_BOUND["fast"] * n


def _relation_plan(e, sub, B):
    graph = {k: torch.from_numpy(sub[k]) for k in ("y", "edge_index", "edge_attr", "batch")}
    return e.make_relation(graph, sub["centres"], sub["canvas_bins"], 3e6, 3, B)


@pytest.mark.parametrize("precision", ["exact", "fast"])
def test_config5_T200_relation_teacher_forced(cuda, golden_dir, precision):
    """cond=relation on all 200 states of the reference's own T = 200 run: posterior -> strong mask -> logit adjustment
    (t >= 10; lr 3e6, 3 updates: the reference's defaults) -> [PAD] disable -> argmax.  The SGD steps are O(1e4) in
    log-probability and its hinges switch on the last bits of an expected box, so the reference's own float32 token can be
    an artefact of its summation order: the fixture annotates those (exact: every other token bit-exact; fast: counted and
    bounded).  (The final `bad_plain > bad` line only shows the adjustment is on the path; its STRENGTH is pinned by
    test_relation_update_vs_reference_and_oracle against the reference's autograd update.)"""
    spec, g, sd = _config5(golden_dir)
    sub = _sub(g, "rel_")
    e = _engine(spec, "t200", sd, precision)
    B = sub["cond_seq"].shape[0]
    cond = {"seq": sub["cond_seq"].astype(np.int64), "mask": sub["cond_mask"], "type": "relation"}
    plan = _relation_plan(e, sub, B)
    bad, n, worst, _ = _traj(e, sub, cond=cond, relation=plan)
    print(f"[config 5 shape / relation T=200 / {precision}] greedy tokens differing from the reference: {bad}/{n}"
          + (f" (largest reference margin among them {worst:.3e})" if bad else ""))
    if precision == "exact":
        # r05: bit-exact, except on the tokens the fixture ANNOTATES as decided by rounding inside the
        # reference's own update — `rel_order_dependent`: the reference's float32 update and the same update in float64 on
        # identical inputs pick different tokens there (oracle/make_golden.py relation_order_dependent_tokens; 1 of 50 000:
        # state 3, float32 -> 152 with a margin of 0.18 nat, float64 -> 145; profiles/r05_relation_order_dependence.txt).
        # On those the engine must return one of the two answers; everywhere else the reference's token.
        od = {(int(r[0]), int(r[1]), int(r[2])): (int(r[3]), int(r[4])) for r in sub["order_dependent"]}
        before = torch.from_numpy(sub["states_before"].astype(np.int32))
        ref_next = torch.from_numpy(sub["greedy_next"].astype(np.int32))
        unexplained, at_annotated = [], []
        for i, t in enumerate(sub["steps"]):
            out = e.sample_step(before[i], int(t), GREEDY, cond=cond, step=i, relation=plan).cpu()
            for b_, s_ in (out != ref_next[i]).nonzero().tolist():
                if (i, b_, s_) in od and int(out[b_, s_]) in od[(i, b_, s_)]:
                    at_annotated.append((i, b_, s_, int(out[b_, s_])))
                else:
                    unexplained.append((i, b_, s_, int(out[b_, s_]), int(ref_next[i, b_, s_])))
        print(f"[config 5 shape / relation T=200 / exact] differing tokens at annotated order-dependent positions: {at_annotated} "
              f"(annotation: {sorted(od.items())}); unexplained: {unexplained}")
        assert not unexplained and len(at_annotated) == bad <= len(od)     # nothing differs outside the annotation
    else:
        assert bad <= 1e-3 * n, (bad, n, worst)
    # and the adjustment is on the path: without it tokens differ
    bad_plain, _, _, _ = _traj(e, sub, cond=dict(cond, type="c"))
    assert bad_plain > bad


def test_config5_T200_loops_equal_their_steps(cuda, golden_dir):
    """The T = 200 loops against the same steps one ldm_sample_step at a time, on config 5's conds:
    refinement — fast: the one-launch loop (two launches: 128 + 72 steps) and, in dev mode, the per-step hipGraph path;
    relation — the per-step hipGraph path (posterior / relation_update / draw per chunk-step, 200 steps captured), fast
    and exact; graph == eager."""
    spec, g, sd = _config5(golden_dir)
    ref_sub, rel_sub = _sub(g, "ref_"), _sub(g, "rel_")
    steps = R.timestep_list(spec.n_step, 200)
    cfg = {"name": "random", "temperature": 1.0}
    for precision in ("fast", "exact"):
        e = _engine(spec, "t200", sd, precision)
        # refinement
        cond = _refinement_cond(ref_sub, spec)
        start = torch.from_numpy(ref_sub["cond_seq"].astype(np.int32)).to(cuda)
        out, inter = e.sample_loop(start.clone(), steps, steps, cfg, cond=cond, seed=3, first_layout=5, intermediates=True)
        cur = start.clone()
        for i, t in enumerate(steps):
            cur = e.sample_step(cur, t, cfg, cond=cond, seed=3, first_layout=5, step=i)
            if i in (0, 1, 100, 127, 128, 129, 198, 199):
                assert torch.equal(cur, inter[i]), (precision, "refinement", i)
        m = torch.from_numpy(ref_sub["cond_mask"])
        assert torch.equal(out.cpu().long()[m], torch.from_numpy(ref_sub["cond_seq"].astype(np.int64))[m])
        assert (out != spec.mask_id).all()
        # relation
        B = rel_sub["cond_seq"].shape[0]
        cond = {"seq": rel_sub["cond_seq"].astype(np.int64), "mask": rel_sub["cond_mask"], "type": "relation"}
        plan = _relation_plan(e, rel_sub, B)
        start = torch.from_numpy(rel_sub["cond_seq"].astype(np.int32)).to(cuda)
        out, inter = e.sample_loop(start.clone(), steps, steps, cfg, cond=cond, seed=4, intermediates=True, relation=plan)
        eager, _ = e.sample_loop(start.clone(), steps, steps, cfg, cond=cond, seed=4, relation=plan, use_graph=False)
        assert torch.equal(out, eager), (precision, "relation graph vs eager")
        cur = start.clone()
        for i, t in enumerate(steps):
            cur = e.sample_step(cur, t, cfg, cond=cond, seed=4, step=i, relation=plan)
            if i in (0, 1, 100, 188, 189, 190, 191, 199):       # both sides of t = 10
                assert torch.equal(cur, inter[i]), (precision, "relation", i)
        assert (out != spec.mask_id).all()
        torch.cuda.synchronize()


def test_relation_loop_general_form_equals_the_packed_form(cuda):
    """The loop kernel stages a layout's graph once per launch in a packed form (src | dst << 6 | attr << 12, ldm_relation_core.h
    RelPersist); a graph that does not fit it takes the general form of the same SGD (edges re-staged from global memory).  An
    attribute bit that no cost reads (1 << 20) forces that path without changing the arithmetic: tokens must be identical."""
    from layout_dm_amd.binding import Engine
    from layout_dm_amd.synthetic import linear_bin_centres, synth_cond_relation
    from layout_dm_amd import synthetic as PS

    spec = SP.RICO25
    B = 96
    e = Engine(n_category=spec.n_category, precision="fast", max_batch=128)
    e.load_state_dict(synth.synth_state_dict(spec, seed=1, perturb=True))
    cond_np, graph = synth_cond_relation(PS.SPECS["rico25"], B, seed=5, edge_ratio=0.3)
    cond = {"seq": cond_np["seq"], "mask": cond_np["mask"], "type": "relation"}
    steps = R.timestep_list(spec.n_step, 100)[:40]
    outs = []
    for high_bit in (False, True):
        plan = e.make_relation(graph, linear_bin_centres(spec.n_bin), [16, 16, 31, 31], 3e6, 3, B)
        if high_bit:
            plan[1][3].bitwise_or_(1 << 20)
        tok = torch.from_numpy(cond_np["seq"]).int().to(cuda)
        out, inter = e.sample_loop(tok, steps, steps, {"name": "random", "temperature": 1.0}, cond=cond, seed=3, intermediates=True,
                                   relation=plan)
        outs.append((out.cpu().clone(), inter.cpu().clone()))
        torch.cuda.synchronize()
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][0], outs[1][0])
    e.close()


def test_relation_in_the_loop_kernel_equals_the_per_step_path(cuda, monkeypatch):
    """r04: cond=relation inside the one-launch loop (stack_stream_k<., 2, true>: posterior -> SGD -> [PAD] disable -> draw behind
    the vocabulary head, ldm_relation_core.h) against the per-step path (LDM_DEV=1 LDM_REL_LOOP=0: stack launch +
    relation_step_k per chunk-step, hipGraph), B = 300 random graphs, T = 100, three samplers.  One source for the SGD and
    for the step's tail; what differs is the summation order of the log-softmax (as for the plain loop:
    test_fused_loop_equals_per_step_path), so tokens may differ only where that moves a tie or a CDF edge — and, behind
    an SGD with steps of O(1e4), where it moves a hinge.  Conditioned categories survive; no [MASK] / [PAD] in free slots."""
    from layout_dm_amd.binding import Engine
    from layout_dm_amd.synthetic import linear_bin_centres, synth_cond_relation
    from layout_dm_amd import synthetic as PS

    spec = SP.RICO25
    sd = synth.synth_state_dict(spec, seed=1, perturb=True)
    B = 300
    cond_np, graph = synth_cond_relation(PS.SPECS["rico25"], B, seed=3, edge_ratio=0.3)
    steps = R.timestep_list(spec.n_step, 100)
    outs = {}
    for loop in ("1", "0"):
        monkeypatch.setenv("LDM_DEV", "1")
        monkeypatch.setenv("LDM_REL_LOOP", loop)
        e = Engine(n_category=spec.n_category, precision="fast", max_batch=512)
        e.load_state_dict(sd)
        assert e.describe()["loop"] == "one_launch"
        cond = {"seq": cond_np["seq"], "mask": cond_np["mask"], "type": "relation"}
        plan = e.make_relation(graph, linear_bin_centres(spec.n_bin), [16, 16, 31, 31], 3e6, 3, B)
        res = {}
        for name in ("deterministic", "random", "top_p"):
            cfg = {"name": name, "temperature": 1.0, "top_p": 0.9}
            tok = torch.from_numpy(cond_np["seq"]).int().to(cuda)
            out, inter = e.sample_loop(tok, steps, steps, cfg, cond=cond, seed=7, first_layout=11, intermediates=True,
                                       relation=plan)
            res[name] = (out.cpu().clone(), inter.cpu().clone())
            if loop == "1" and name == "random":   # the loop == the same steps one launch at a time (same kernel, same state)
                cur = torch.from_numpy(cond_np["seq"]).int().to(cuda)
                for i, t in enumerate(steps[:14]):
                    cur = e.sample_step(cur, t, cfg, cond=cond, seed=7, first_layout=11, step=i, relation=plan)
                    assert torch.equal(cur.cpu(), res[name][1][i]), i
        torch.cuda.synchronize()
        e.close()
        outs[loop] = res
    m = torch.from_numpy(cond_np["mask"])
    seq = torch.from_numpy(cond_np["seq"])
    for name in ("deterministic", "random", "top_p"):
        a, b = outs["1"][name], outs["0"][name]
        final = (a[0] != b[0]).float().mean().item()
        early = (a[1][:10] != b[1][:10]).float().mean().item()
        print(f"[relation in the loop kernel vs per-step path / {name}] tokens differing: first 10 steps {early:.2e}, final {final:.2e}")
        assert early <= 5e-4 and final <= 5e-3, (name, early, final)
        assert torch.equal(a[0].long()[m], seq[m]), "conditioned tokens changed"
        free = ~m
        assert (a[0].long()[free] != spec.mask_id).all() and (a[0].long()[free] != spec.pad_id).all()


# ----------------------------------------------------------------------------- fast_verified where it hurts (B = 512)
def test_fast_verified_free_running_from_mid_trajectory_b512(cuda):
    """greedy free-running loops at B = 512 started from the states a stochastic run visits at step 20,
    50, 80 (t = 79, 49, 19): the verified loop == the exact engine's greedy loop (every intermediate), and its cost is
    what the marks cost: exact_fraction / relaunched_fraction / wall time per remaining step are printed."""
    import time

    from layout_dm_amd.verified import VerifiedGreedy
    from test_hip_parity import engine

    spec, B = SP.RICO25, 512
    fa, ex = engine("rico25", "fast", max_batch=B), engine("rico25", "exact", max_batch=B)
    vg = VerifiedGreedy(fa, ex)
    vg.calibrate()
    steps = R.timestep_list(spec.n_step, 100)
    tok = torch.full((B, spec.seq_len), spec.mask_id, dtype=torch.int32, device=cuda)
    _, inter = fa.sample_loop(tok, steps, steps, {"name": "random", "temperature": 1.0}, seed=9, intermediates=True)
    inter = inter.clone()
    for i0 in (20, 50, 80):
        start = inter[i0 - 1].clone()
        want, want_inter = ex.sample_loop(start.clone(), steps[i0:], steps[i0:], GREEDY, intermediates=True)
        vg.sample_loop(start.clone(), steps[i0:], steps[i0:])       # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got, got_inter = vg.sample_loop(start.clone(), steps[i0:], steps[i0:], intermediates=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = vg.last_stats
        print(f"[fast_verified B=512 from step {i0}] {dt * 1e3:.1f} ms for {100 - i0} steps = "
              f"{B * (100 - i0) / 100 / dt:.0f} layout-equivalents/s; exact_fraction {st['exact_fraction']:.4f} "
              f"relaunched_fraction {st['relaunched_fraction']:.5f} mismatch layout-steps {st['mismatch_layout_steps']} "
              f"fast passes {st['fast_passes']}")
        assert torch.equal(got, want)
        assert torch.equal(got_inter, want_inter)
        assert st["exact_fraction"] < 0.25
    fa.set_tie_report(0.0, 0.0)


# ----------------------------------------------------------------------------- hygiene: knobs, description
def test_dev_knobs_are_refused_outside_dev_mode(cuda, monkeypatch):
    from layout_dm_amd.binding import Engine

    spec = SP.RICO25
    monkeypatch.delenv("LDM_DEV", raising=False)
    monkeypatch.setenv("LDM_STACK_LOOP", "0")
    with pytest.raises(RuntimeError, match="LDM_STACK_LOOP.*LDM_DEV"):
        Engine(n_category=spec.n_category, precision="fast", max_batch=4)
    monkeypatch.delenv("LDM_STACK_LOOP")
    e = Engine(n_category=spec.n_category, precision="fast", max_batch=4)
    e.load_state_dict(synth.synth_state_dict(spec, seed=1))
    d = e.describe()
    assert d["precision"] == "fast_f16" and d["loop"] == "one_launch" and d["kernels"] == "stack", d
    # (d["knobs"] lists what the library honoured so far in this PROCESS — other tests of this session run in dev mode)
    assert d["abi"] == "5" and "knobs" in d
    from layout_dm_amd import build

    assert d["src_digest"] == build.source_digest()   # the library that runs was built from this tree
    e.close()


def test_tie_report_refused_for_relation(cuda, golden_dir):
    """greedy cond=relation with the near-tie report enabled must not silently return unverified tokens."""
    spec, g, sd = _config5(golden_dir)
    sub = _sub(g, "rel_")
    e = _engine(spec, "t200", sd, "fast")
    B = sub["cond_seq"].shape[0]
    cond = {"seq": sub["cond_seq"].astype(np.int64), "mask": sub["cond_mask"], "type": "relation"}
    plan = _relation_plan(e, sub, B)
    e.set_tie_report(6e-3, 1e-2)
    with pytest.raises(RuntimeError, match="not defined for cond=relation"):
        e.sample_step(torch.from_numpy(sub["cond_seq"].astype(np.int32)), 150, GREEDY, cond=cond, relation=plan)
    e.set_tie_report(0.0, 0.0)
    e.sample_step(torch.from_numpy(sub["cond_seq"].astype(np.int32)), 150, GREEDY, cond=cond, relation=plan)
