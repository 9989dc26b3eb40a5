"""GPU parity tests: the HIP path (through the C-ABI, via layout_dm_amd.binding) against the
oracle restatement and the reference-produced golden fixtures.

Tolerances (north star): token indices bit-exact under greedy decoding in the `exact` numerics
mode (fp32 MFMA); logits within 1e-3 relative (max |diff| / max |ref logit|) in every mode.
"""
import os

import numpy as np
import pytest
import torch

from oracle import restatement as R
from oracle import spec as SP
from oracle import synth

pytestmark = pytest.mark.gpu

WEIGHT_SEED = 1
LOGIT_REL_TOL = {"exact": 2e-5, "split": 5e-5, "fast": 1e-3, "mixed": 1e-3, "hybrid": 1e-3}   # (mixed: the split mode with fp16-only weights, r06)


def _rel(a: torch.Tensor, ref: torch.Tensor) -> float:
    return ((a - ref).abs().max() / ref.abs().max()).item()


@pytest.fixture(scope="module")
def cuda():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device (no CPU fallback exists)")
    return torch.device("cuda", 0)


_ENG = {}


def engine(ds: str, precision: str, max_batch: int = 8, chunk: int = 0):
    from layout_dm_amd.binding import Engine

    key = (ds, precision, max_batch, chunk)
    if key not in _ENG:
        spec = SP.SPECS[ds]
        e = Engine(n_category=spec.n_category, n_bin=spec.n_bin, max_elem=spec.max_elem, d_model=spec.d_model,
                   n_head=spec.n_head, d_ff=spec.d_ff, n_layer=spec.n_layer, n_step=spec.n_step,
                   precision=precision, max_batch=max_batch, chunk=chunk)
        e.load_state_dict(synth.synth_state_dict(spec, seed=WEIGHT_SEED, perturb=True))
        _ENG[key] = e
    return _ENG[key]


def weights(ds):
    spec = SP.SPECS[ds]
    return spec, R.as_torch_weights(synth.synth_state_dict(spec, seed=WEIGHT_SEED, perturb=True))


# ----------------------------------------------------------------------------- denoiser
@pytest.mark.parametrize("precision", ["exact", "split", "mixed", "hybrid", "fast"])
@pytest.mark.parametrize("ds", ["rico25", "publaynet"])
def test_denoiser_logits_vs_reference_golden(cuda, golden_dir, ds, precision):
    """ldm_denoise_logits == reference CategoricalTransformer.forward (golden, B=2)."""
    e = engine(ds, precision)
    g = np.load(os.path.join(golden_dir, f"{ds}_step_cases.npz"))
    worst = 0.0
    for t in g["ts"]:
        tokens = torch.from_numpy(g[f"tokens_{int(t)}"].astype(np.int32))
        ref = torch.from_numpy(g[f"logits_{int(t)}"])
        out = e.denoise_logits(tokens, int(t)).cpu()
        worst = max(worst, _rel(out, ref))
    print(f"[{ds}/{precision}] max rel logits error vs reference: {worst:.3e}")
    assert worst <= LOGIT_REL_TOL[precision]


def test_stack_kernel_agrees_with_generic_kernels(cuda, monkeypatch):
    """The layout-resident stack kernel (kernels_stack.hip: all layers + vocabulary head per launch, rows in the
    accumulators) against the generic tiled kernels every other geometry runs on (LDM_FUSED_ATTN=0: LayerNorm -> gemm16 ->
    attention16 -> ...): same weights, same tokens; both within the reference tolerance of the fp64-softmax oracle.  The two
    differ only in fp16 rounding points (V bias folded into the out-projection bias, packed softmax arithmetic, v_rcp_f32,
    row statistics recomputed from the accumulators, ReLU after the fp16 cast)."""
    from layout_dm_amd.binding import Engine

    spec, W = weights("rico25")
    sd = synth.synth_state_dict(spec, seed=WEIGHT_SEED, perturb=True)
    g = torch.Generator().manual_seed(11)
    tokens = torch.randint(0, spec.n_class, (6, spec.seq_len), generator=g).int()
    ref = R.denoiser_logits(W, spec, tokens.long(), 23)
    outs = {}
    for gen in ("6", "0"):
        monkeypatch.setenv("LDM_DEV", "1")  # development knobs are honoured in dev mode only (csrc/ldm_knobs.h)
        monkeypatch.setenv("LDM_FUSED_ATTN", gen)
        e = Engine(n_category=spec.n_category, n_bin=spec.n_bin, max_elem=spec.max_elem, d_model=spec.d_model,
                   n_head=spec.n_head, d_ff=spec.d_ff, n_layer=spec.n_layer, n_step=spec.n_step, precision="fast",
                   max_batch=8)
        e.load_state_dict(sd)
        outs[gen] = e.denoise_logits(tokens, 23).cpu()
        # the generic path also runs the whole loop (per-step launches in hipGraphs)
        if gen == "0":
            steps = R.timestep_list(spec.n_step, 10)
            tok = torch.full((5, spec.seq_len), spec.mask_id, dtype=torch.int32, device=cuda)
            out = e.sample_loop(tok, steps, steps, {"name": "random", "temperature": 1.0}, seed=3)[0]
            assert (out != spec.mask_id).all()
        e.close()
        assert _rel(outs[gen], ref) <= LOGIT_REL_TOL["fast"], gen
    assert _rel(outs["6"], outs["0"]) <= 5e-4


@pytest.mark.parametrize("precision", ["exact", "split", "fast"])
@pytest.mark.parametrize("shape", ["short_sequence", "mid_sequence", "long_sequence", "small_backbone"])
def test_denoiser_other_geometries_take_the_generic_kernels(cuda, shape, precision):
    """The layout-resident kernels are built for the reference's one backbone (medium shrunk by 29/32: d_model 464,
    8 heads, d_ff 1856) at 96 < S <= 128 tokens.  Any other geometry must still give the reference's numbers through
    the generic tiled GEMM / attention kernels (ldm_create picks them; nothing is silently approximated)."""
    import dataclasses

    from layout_dm_amd.binding import Engine

    base = SP.SPECS["rico25"]
    spec = {"short_sequence": dataclasses.replace(base, name="short", max_elem=10),            # S = 50
            # S = 105: still the layout-resident kernels (96 < S <= 128), with 23 padded rows / keys instead of 3
            "mid_sequence": dataclasses.replace(base, name="mid", max_elem=21),
            "long_sequence": dataclasses.replace(base, name="long", max_elem=30, n_layer=2),   # S = 150 (exact only)
            "small_backbone": dataclasses.replace(base, name="small", d_model=256, n_head=4,   # head dim 64
                                                  d_ff=1024, n_layer=2, n_step=20)}[shape]
    sd = synth.synth_state_dict(spec, seed=WEIGHT_SEED, perturb=True)
    W = R.as_torch_weights(sd)
    kw = dict(n_category=spec.n_category, n_bin=spec.n_bin, max_elem=spec.max_elem, d_model=spec.d_model,
              n_head=spec.n_head, d_ff=spec.d_ff, n_layer=spec.n_layer, n_step=spec.n_step, precision=precision,
              max_batch=8)
    if shape == "long_sequence" and precision == "fast":
        with pytest.raises(RuntimeError, match="at most 128 tokens"):  # one 128 x 128 score tile per (layout, head)
            Engine(**kw)
        return
    e = Engine(**kw)
    e.load_state_dict(sd)
    g = torch.Generator().manual_seed(5)
    tokens = torch.randint(0, spec.n_class, (5, spec.seq_len), generator=g)
    t = spec.n_step // 3
    out = e.denoise_logits(tokens.int(), t).cpu()
    e.close()
    assert _rel(out, R.denoiser_logits(W, spec, tokens, t)) <= LOGIT_REL_TOL[precision]


@pytest.mark.parametrize("precision", ["exact", "split", "fast"])
def test_denoiser_ragged_batch_and_chunks(cuda, precision):
    """B not a multiple of the chunk / of the 128-row GEMM tile; rows must not interact."""
    spec, W = weights("rico25")
    e = engine("rico25", precision, max_batch=11, chunk=4)
    g = torch.Generator().manual_seed(3)
    tokens = torch.randint(0, spec.n_class, (11, spec.seq_len), generator=g)
    out = e.denoise_logits(tokens.int(), 37).cpu()
    ref = R.denoiser_logits(W, spec, tokens, 37)
    assert _rel(out, ref) <= LOGIT_REL_TOL[precision]
    one = e.denoise_logits(tokens[7:8].int(), 37).cpu()
    assert torch.equal(one[0], out[7])  # batch-composition independent, bit for bit


# ----------------------------------------------------------------------------- posterior
@pytest.mark.parametrize("ds", ["rico25", "publaynet"])
def test_posterior_vs_reference_golden(cuda, golden_dir, ds):
    """ldm_posterior == predict_start tail + q_posterior of the reference (golden)."""
    e = engine(ds, "exact")
    g = np.load(os.path.join(golden_dir, f"{ds}_step_cases.npz"))
    for t in g["ts"]:
        t = int(t)
        tokens = torch.from_numpy(g[f"tokens_{t}"].astype(np.int32))
        logits = torch.from_numpy(g[f"logits_{t}"])
        ref = torch.from_numpy(g[f"post_{t}"])
        out = e.posterior(logits, tokens, t).cpu()
        assert (out - ref).abs().max().item() <= 2e-4, t
        assert torch.equal(out.argmax(1), ref.argmax(1))
        dead = ref == ref.min()
        assert torch.equal(out[dead], ref[dead])  # log(1e-30) fill is exact


def test_posterior_cond_overrides(cuda, golden_dir):
    """strong mask / refinement prior / PAD disable (base.py:243-284) vs oracle.apply_cond."""
    spec, W = weights("rico25")
    e = engine("rico25", "exact")
    g = np.load(os.path.join(golden_dir, "rico25_refinement_trajectory.npz"))
    seq, mask = g["cond_seq"].astype(np.int64), g["cond_mask"]
    table = torch.from_numpy(g["weak_table"])
    seq_orig = torch.from_numpy(g["seq_orig"].astype(np.int64))
    wl = table[seq_orig].permute(0, 2, 1).contiguous()
    wm = torch.from_numpy(~mask)[:, None, :].expand(-1, spec.n_class, -1)
    for ctype in ("refinement", "c", "partial"):
        cond = {"seq": seq, "mask": mask, "type": ctype, "weak_mask": wm, "weak_logits": wl}
        for i in (0, 40, 99):
            toks = torch.from_numpy(g["states_before"][i].astype(np.int64))
            t = int(g["steps"][i])
            logits = R.denoiser_logits(W, spec, toks, t)
            ref = R.apply_cond(spec, R.q_posterior(W, spec, R.predict_start_from_logits(logits), toks, t), cond)
            out = e.posterior(logits, toks.int(), t, cond).cpu()
            assert (out - ref).abs().max().item() <= 2e-4, (ctype, i)


# ----------------------------------------------------------------------------- sampler
def test_sampler_deterministic_and_inverse_cdf(cuda):
    spec = SP.RICO25
    e = engine("rico25", "exact", max_batch=64)
    g = torch.Generator().manual_seed(0)
    B = 64
    logp = torch.log_softmax(3.0 * torch.randn(B, spec.n_class, spec.seq_len, generator=g), dim=1)
    det = e.sample_tokens(logp, {"name": "deterministic"}).cpu().long()
    assert torch.equal(det, logp.argmax(1))
    # ties resolve to the first maximum like torch.argmax
    tie = logp.clone()
    tie[:, 5, :] = 1.0
    tie[:, 90, :] = 1.0
    assert (e.sample_tokens(tie, {"name": "deterministic"}).cpu() == 5).all()
    for cfg in ({"name": "random", "temperature": 1.0}, {"name": "random", "temperature": 0.7},
                {"name": "top_p", "top_p": 0.9, "temperature": 1.0}, {"name": "top_k", "top_k": 5, "temperature": 1.0}):
        out = e.sample_tokens(logp, cfg, seed=11, first_layout=5, step=42).cpu().long()
        u = R.token_uniforms(11, 5, B, spec.seq_len, 42)[..., 0]
        ref = R.sample_tokens(logp, cfg, uniforms=u)
        frac = (out != ref).float().mean().item()
        assert frac <= 2e-3, (cfg, frac)  # only draws whose u sits within fp32 rounding of a CDF edge
        # never a filtered-out class
        probs = R.sample_probs(logp, cfg)
        assert (probs.gather(1, out[:, None, :]) > 0).all()


def test_sampler_statistics_match_reference_distribution(cuda):
    """chi-square of our Philox draws against the probabilities the reference hands to
    torch.multinomial (oracle.sample_probs == golden top_p_probs, see test_oracle_golden)."""
    spec = SP.RICO25
    B = 512
    e = engine("rico25", "exact", max_batch=B)
    g = torch.Generator().manual_seed(1)
    row = torch.full((spec.n_class,), SP.LOG_EPS)
    live = torch.as_tensor(spec.full_ids(1))
    row[live] = torch.log_softmax(torch.randn(len(live), generator=g) * 1.5, 0)
    logp = row.view(1, -1, 1).repeat(B, 1, spec.seq_len).contiguous()
    for cfg in ({"name": "random", "temperature": 1.0}, {"name": "top_p", "top_p": 0.8, "temperature": 1.0},
                {"name": "gumbel", "temperature": 1.0}):
        out = e.sample_tokens(logp, cfg, seed=3, step=7).cpu().long().ravel().numpy()
        if cfg["name"] == "gumbel":
            # gumbel noise THEN multinomial (sampling.py:112-127): E[softmax(l+g)] has no closed form;
            # check support + that it is more spread than the plain distribution's mode
            assert np.isin(out, live.numpy()).all()
            continue
        p = R.sample_probs(logp[:1, :, :1], cfg)[0, :, 0].numpy().astype(np.float64)
        cnt = np.bincount(out, minlength=spec.n_class).astype(np.float64)
        n = cnt.sum()
        keep = p * n >= 5
        chi2 = (((cnt - n * p) ** 2)[keep] / (n * p)[keep]).sum()
        dof = keep.sum() - 1
        assert cnt[~keep].sum() <= 5 * (~keep).sum() + 10
        assert chi2 < dof + 6 * np.sqrt(2 * dof) + 10, (cfg, chi2, dof)


# ----------------------------------------------------------------------------- fused step / loop
# Greedy-token criterion per numerics mode (north star: "token indices bit-exact under greedy/argmax decoding"):
#   exact : 0 mismatches against the reference's own argmax tokens;
#   split / fast : a token may differ from the reference ONLY where the reference's own top-2 log-probability gap
#   (`greedy_margin`, captured at the reference's sample() call site by oracle/make_golden.py) is below the bound
#   the mode's logits tolerance allows: the fast mode's measured logits error is <= 4.6e-4 of max |logit| (~2 on these
#   weights), i.e. <= ~1e-3 absolute per class, so only two classes closer than 2e-3 are a legitimate tie for it
#   (largest margin ever observed among its mismatches: 3.4e-4), AND at most MISMATCH_COUNT_BOUND of the tokens may be
#   such ties (observed: 1 / 162 500 and 4 / 112 500).  `fast_verified` (tests/test_fast_verified.py) removes even those.
MARGIN_BOUND = {"exact": 0.0, "split": 1e-4, "fast": 2e-3, "mixed": 2e-3, "hybrid": 2e-3}
MISMATCH_COUNT_BOUND = {"exact": 0.0, "split": 2e-5, "fast": 2e-4, "mixed": 2e-4, "hybrid": 2e-4}


def _traj_check(e, g, cfg, cond=None, prefix=""):
    steps = g[prefix + "steps"]
    before = torch.from_numpy(g[prefix + "states_before"].astype(np.int32))
    ref_next = torch.from_numpy(g[prefix + "greedy_next"].astype(np.int32))
    margin = torch.from_numpy(g[prefix + "greedy_margin"])
    bad, worst = 0, 0.0
    for i, t in enumerate(steps):
        out = e.sample_step(before[i], int(t), cfg, cond=cond, step=i).cpu()
        mism = out != ref_next[i]
        if mism.any():
            bad += int(mism.sum())
            worst = max(worst, margin[i][mism].max().item())
    return bad, ref_next.numel(), worst


def _assert_traj(name, precision, bad, n, worst):
    print(f"[{name}/{precision}] greedy tokens differing from the reference: {bad}/{n}"
          + (f" (largest reference top-2 margin among them {worst:.3e})" if bad else ""))
    if precision == "exact":
        assert bad == 0, f"{bad}/{n} tokens differ"
    else:
        assert bad == 0 or worst < MARGIN_BOUND[precision], (bad, worst)
        assert bad <= MISMATCH_COUNT_BOUND[precision] * n, (bad, n)


@pytest.mark.parametrize("precision", ["exact", "split", "mixed", "hybrid", "fast"])
def test_step_teacher_forced_uncond_all_t(cuda, golden_dir, precision):
    """Every t in 99..0: fused HIP step (greedy) vs the reference's own argmax tokens on the states visited by a
    stochastic reference trajectory: bit-exact in `exact`, margin-bounded in the fp16 modes (see MARGIN_BOUND)."""
    e = engine("rico25", precision)
    g = np.load(os.path.join(golden_dir, "rico25_uncond_trajectory.npz"))
    _assert_traj("uncond", precision, *_traj_check(e, g, {"name": "deterministic"}))


@pytest.mark.parametrize("precision", ["exact", "split", "mixed", "hybrid", "fast"])
def test_step_teacher_forced_cond_c(cuda, golden_dir, precision):
    e = engine("publaynet", precision)
    g = np.load(os.path.join(golden_dir, "publaynet_cond_c_trajectory.npz"))
    cond = {"seq": g["cond_seq"].astype(np.int64), "mask": g["cond_mask"], "type": "c"}
    _assert_traj("cond=c", precision, *_traj_check(e, g, {"name": "deterministic"}, cond))


@pytest.mark.parametrize("precision", ["exact", "split", "mixed", "hybrid", "fast"])
def test_step_teacher_forced_refinement(cuda, golden_dir, precision):
    e = engine("rico25", precision)
    g = np.load(os.path.join(golden_dir, "rico25_refinement_trajectory.npz"))
    table = torch.from_numpy(g["weak_table"])
    seq_orig = torch.from_numpy(g["seq_orig"].astype(np.int64))
    cond = {"seq": g["cond_seq"].astype(np.int64), "mask": g["cond_mask"], "type": "refinement",
            "weak_logits": table[seq_orig].permute(0, 2, 1).contiguous()}
    _assert_traj("refinement", precision, *_traj_check(e, g, {"name": "deterministic"}, cond))


@pytest.mark.parametrize("ds", ["rico25", "publaynet"])
def test_posterior_fast_mode_f32_lse_vs_reference_golden(cuda, golden_dir, ds):
    """The fast mode replaces the reference's float64 log-softmax (base.py:137) by an fp32 one inside the fused step
    (kernels_post.hip, f32_lse).  ldm_posterior on a fast-mode handle runs that variant: it must reproduce the
    reference's post_* golden (reference logits in) to the same tolerance as the f64 path, argmax included."""
    e = engine(ds, "fast")
    g = np.load(os.path.join(golden_dir, f"{ds}_step_cases.npz"))
    for t in g["ts"]:
        t = int(t)
        tokens = torch.from_numpy(g[f"tokens_{t}"].astype(np.int32))
        ref = torch.from_numpy(g[f"post_{t}"])
        out = e.posterior(torch.from_numpy(g[f"logits_{t}"]), tokens, t).cpu()
        assert (out - ref).abs().max().item() <= 5e-4, t
        assert torch.equal(out.argmax(1), ref.argmax(1))
        dead = ref == ref.min()
        assert torch.equal(out[dead], ref[dead])


@pytest.mark.parametrize("use_graph", [False, True])
def test_loop_greedy_matches_reference(cuda, golden_dir, use_graph):
    """Full T=100 greedy loop (and the strided T=25 schedule) == reference sample()."""
    spec = SP.RICO25
    e = engine("rico25", "exact")
    g = np.load(os.path.join(golden_dir, "rico25_uncond_greedy_loop.npz"))
    ref = torch.from_numpy(g["states_after"].astype(np.int32))
    B = ref.shape[1]
    steps = R.timestep_list(spec.n_step, 100)
    tok = torch.full((B, spec.seq_len), spec.mask_id, dtype=torch.int32, device=cuda)
    out, inter = e.sample_loop(tok, steps, steps, {"name": "deterministic"}, intermediates=True, use_graph=use_graph)
    assert torch.equal(inter.cpu(), ref)
    assert torch.equal(out.cpu(), ref[-1])
    # strided: t_post = t - skip_step when t > skip_step (base.py:227-235)
    g = np.load(os.path.join(golden_dir, "rico25_uncond_greedy_T25.npz"))
    ref = torch.from_numpy(g["states_after"].astype(np.int32))
    steps = R.timestep_list(spec.n_step, 25)
    tpost, prev = [], spec.n_step
    for t in steps:
        skip = prev - t - 1
        tpost.append(t - skip if (skip > 0 and t > skip) else t)
        prev = t
    tok = torch.full((ref.shape[1], spec.seq_len), spec.mask_id, dtype=torch.int32, device=cuda)
    out, inter = e.sample_loop(tok, steps, tpost, {"name": "deterministic"}, intermediates=True, use_graph=use_graph)
    assert torch.equal(inter.cpu(), ref)


def test_loop_random_vs_oracle_same_uniforms(cuda):
    """Stochastic loop: HIP and oracle consume identical Philox uniforms, so a free-running T=100
    `random` run agrees token-for-token except where fp32 rounding moves a CDF edge across u."""
    spec, W = weights("rico25")
    e = engine("rico25", "exact")
    B = 4
    steps = R.timestep_list(spec.n_step, 100)
    tok = torch.full((B, spec.seq_len), spec.mask_id, dtype=torch.int32, device=cuda)
    out, inter = e.sample_loop(tok, steps, steps, {"name": "random", "temperature": 1.0}, seed=123, first_layout=1000,
                               intermediates=True, use_graph=False)
    ref = R.sample_loop(W, spec, B, {"name": "random", "temperature": 1.0}, seed=123, first_layout=1000,
                        get_intermediate_results=True)
    ref = torch.stack(ref).int()
    diff = inter.cpu() != ref
    frac = diff.float().mean().item()
    # a draw differs only where fp32 rounding moves a CDF edge across u (~1e-4 of the tokens); a layout that took such a
    # draw then follows its own trajectory, so what is bounded is (a) the first-divergence events and (b) the total
    print(f"[loop random / exact, B={B}] tokens differing from the oracle on identical uniforms: {int(diff.sum())}/"
          f"{diff.numel()} = {frac:.2e}; layouts that diverged at some step: {int(diff.any(dim=2).any(dim=0).sum())}/{B}")
    assert frac <= 1e-3, frac
    final = out.cpu()
    assert (final != spec.mask_id).all()


@pytest.mark.parametrize("shape", ["short_sequence", "long_sequence", "small_backbone"])
def test_loop_other_geometries_vs_oracle_same_uniforms(cuda, shape):
    """The whole reverse loop (denoiser, posterior, draw; hipGraph) at geometries other than the reference's S = 125 /
    d_model 464: free-running `random` sampling against the oracle on identical Philox uniforms (exact mode)."""
    import dataclasses

    from layout_dm_amd.binding import Engine

    base = dataclasses.replace(SP.SPECS["publaynet"], n_layer=2, n_step=20)
    spec = {"short_sequence": dataclasses.replace(base, name="short", max_elem=7),            # S = 35
            "long_sequence": dataclasses.replace(base, name="long", max_elem=30),             # S = 150
            "small_backbone": dataclasses.replace(base, name="small", d_model=192, n_head=6,  # head dim 32
                                                  d_ff=768)}[shape]
    sd = synth.synth_state_dict(spec, seed=WEIGHT_SEED, perturb=True)
    W = R.as_torch_weights(sd)
    e = Engine(n_category=spec.n_category, n_bin=spec.n_bin, max_elem=spec.max_elem, d_model=spec.d_model,
               n_head=spec.n_head, d_ff=spec.d_ff, n_layer=spec.n_layer, n_step=spec.n_step, precision="exact",
               max_batch=8)
    e.load_state_dict(sd)
    B = 6
    steps = R.timestep_list(spec.n_step, spec.n_step)
    tok = torch.full((B, spec.seq_len), spec.mask_id, dtype=torch.int32, device=cuda)
    cfg = {"name": "random", "temperature": 1.0}
    out, inter = e.sample_loop(tok, steps, steps, cfg, seed=7, first_layout=40, intermediates=True, use_graph=True)
    ref = torch.stack(R.sample_loop(W, spec, B, cfg, seed=7, first_layout=40, get_intermediate_results=True)).int()
    e.close()
    frac = (inter.cpu() != ref).float().mean().item()
    assert frac <= 5e-3, frac
    assert (out.cpu() != spec.mask_id).all()


# ----------------------------------------------------------------------------- full size (config 2)
def test_full_batch_512_properties(cuda):
    """BASELINE config 2 size (Rico25, B=512, T=100): size-independent properties.
    * shard invariance: one B=512 call == two B=256 calls with first_layout offsets (what the
      multi-GPU path relies on) — bit-exact tokens;
    * hipGraph replay == eager launches;
    * determinism across repeated replays; no [MASK] survives."""
    spec = SP.RICO25
    B = 512
    e = engine("rico25", "exact", max_batch=B)
    steps = R.timestep_list(spec.n_step, 100)
    cfg = {"name": "random", "temperature": 1.0}
    mk = lambda n: torch.full((n, spec.seq_len), spec.mask_id, dtype=torch.int32, device=cuda)
    full, _ = e.sample_loop(mk(B), steps, steps, cfg, seed=7, first_layout=0, use_graph=True)
    full = full.clone()
    again, _ = e.sample_loop(mk(B), steps, steps, cfg, seed=7, first_layout=0, use_graph=True)
    assert torch.equal(full, again)
    eager, _ = e.sample_loop(mk(B), steps, steps, cfg, seed=7, first_layout=0, use_graph=False)
    assert torch.equal(full, eager)
    lo, _ = e.sample_loop(mk(256), steps, steps, cfg, seed=7, first_layout=0, use_graph=False)
    hi, _ = e.sample_loop(mk(256), steps, steps, cfg, seed=7, first_layout=256, use_graph=False)
    assert torch.equal(full, torch.cat([lo, hi]))
    other, _ = e.sample_loop(mk(B), steps, steps, cfg, seed=8, first_layout=0, use_graph=True)
    assert not torch.equal(full, other)
    assert (full != spec.mask_id).all()
    # every token belongs to its attribute's sub-vocabulary (or PAD)
    f = full.cpu().long()
    for a in range(spec.n_attr):
        ids = torch.as_tensor(spec.full_ids(a))[:-1]
        assert torch.isin(f[:, a::spec.n_attr], ids).all()


def test_split_mode_loop_properties(cuda):
    """The split mode (fp16 x 3 GEMMs on 256 x 256 tiles, r04) on a batch that does not fill its last chunk or its last
    row tile (B = 300: chunks of 256 + 44 layouts = 32 000 + 5 500 rows): hipGraph replay == eager launches, the tokens of
    layout i do not depend on how the batch is cut (rows past M feed products that are never stored), determinism, and a
    teacher-forced greedy step == the exact engine's at B = 300."""
    spec = SP.RICO25
    B = 300
    e = engine("rico25", "split", max_batch=512)
    x = engine("rico25", "exact", max_batch=512)
    steps = R.timestep_list(spec.n_step, 20)
    tpost, prev = [], spec.n_step
    for t in steps:  # base.py:227-235
        skip = prev - t - 1
        tpost.append(t - skip if (skip > 0 and t > skip) else t)
        prev = t
    cfg = {"name": "random", "temperature": 1.0}
    mk = lambda n: torch.full((n, spec.seq_len), spec.mask_id, dtype=torch.int32, device=cuda)
    full, inter = e.sample_loop(mk(B), steps, tpost, cfg, seed=21, first_layout=0, use_graph=True, intermediates=True)
    full, inter = full.clone(), inter.clone()
    eager = e.sample_loop(mk(B), steps, tpost, cfg, seed=21, first_layout=0, use_graph=False)[0]
    assert torch.equal(full, eager)
    again = e.sample_loop(mk(B), steps, tpost, cfg, seed=21, first_layout=0, use_graph=True)[0]
    assert torch.equal(full, again)
    lo = e.sample_loop(mk(100), steps, tpost, cfg, seed=21, first_layout=0, use_graph=False)[0]
    hi = e.sample_loop(mk(200), steps, tpost, cfg, seed=21, first_layout=100, use_graph=False)[0]
    assert torch.equal(full, torch.cat([lo, hi]))
    assert (full != spec.mask_id).all()
    state = inter[9]
    assert torch.equal(e.sample_step(state, steps[10], {"name": "deterministic"}, t_post=tpost[10], step=10),
                       x.sample_step(state, steps[10], {"name": "deterministic"}, t_post=tpost[10], step=10))


@pytest.mark.parametrize("B,sampler", [(1, "random"), (257, "random"), (300, "top_p"), (511, "gumbel")])
def test_loop_ragged_batches_fast_mode(cuda, B, sampler):
    """The shipping configuration (fast mode, 256-layout chunks on two lanes, hipGraph, stack kernel with the fused
    head and the embedding written by the posterior kernel) on batches that do not fill their last chunk: graph ==
    eager, and the tokens of layout i do not depend on how the batch is cut (Philox keyed by the global layout index;
    one workgroup per layout, so no arithmetic depends on the chunk's composition) — bit-exact."""
    spec = SP.RICO25
    e = engine("rico25", "fast", max_batch=512)
    steps = R.timestep_list(spec.n_step, 100)
    cfg = {"name": sampler, "temperature": 1.0, "top_p": 0.9}
    mk = lambda n: torch.full((n, spec.seq_len), spec.mask_id, dtype=torch.int32, device=cuda)
    full = e.sample_loop(mk(B), steps, steps, cfg, seed=11, first_layout=0, use_graph=True)[0].clone()
    eager = e.sample_loop(mk(B), steps, steps, cfg, seed=11, first_layout=0, use_graph=False)[0].clone()
    assert torch.equal(full, eager)
    assert (full != spec.mask_id).all()
    cut = max(1, min(256, B - 1)) if B > 1 else 1
    parts = [e.sample_loop(mk(cut), steps, steps, cfg, seed=11, first_layout=0, use_graph=True)[0].clone()]
    if B > cut:
        parts.append(e.sample_loop(mk(B - cut), steps, steps, cfg, seed=11, first_layout=cut, use_graph=True)[0].clone())
    assert torch.equal(full, torch.cat(parts))


def test_fused_loop_shorter_sequence_vs_oracle(cuda):
    """The one-launch loop at S = 105 tokens (21 elements: 96 < S <= 128 keeps the layout-resident kernels, with 23 padded
    rows / keys per layout instead of 3): free-running `random` and greedy loops against the oracle on identical Philox
    uniforms, and cond=c with its strong mask."""
    import dataclasses

    from layout_dm_amd.binding import Engine

    spec = dataclasses.replace(SP.RICO25, name="rico25_e21", max_elem=21, n_step=30)
    assert spec.seq_len == 105
    sd = synth.synth_state_dict(spec, seed=WEIGHT_SEED, perturb=True)
    W = R.as_torch_weights(sd)
    B = 5
    steps = R.timestep_list(spec.n_step, spec.n_step)
    e = Engine(n_category=spec.n_category, n_bin=spec.n_bin, max_elem=spec.max_elem, d_model=spec.d_model,
               n_head=spec.n_head, d_ff=spec.d_ff, n_layer=spec.n_layer, n_step=spec.n_step, precision="fast", max_batch=8)
    e.load_state_dict(sd)
    for cfg, bound in (({"name": "random", "temperature": 1.0}, 5e-3), ({"name": "deterministic"}, 5e-3)):
        tok = torch.full((B, spec.seq_len), spec.mask_id, dtype=torch.int32, device=cuda)
        out, inter = e.sample_loop(tok, steps, steps, cfg, seed=13, first_layout=21, intermediates=True, use_graph=True)
        ref = torch.stack(R.sample_loop(W, spec, B, cfg, seed=13, first_layout=21, get_intermediate_results=True)).int()
        frac = (inter.cpu() != ref).float().mean().item()
        print(f"[S=105 loop / fast / {cfg['name']}] tokens differing from the oracle: {frac:.2e}")
        assert frac <= bound, (cfg, frac)
        assert (out.cpu() != spec.mask_id).all()
    c = synth.synth_cond_c(spec, B, seed=4)
    cond = {"seq": c["seq"], "mask": c["mask"], "type": "c"}
    tok = torch.from_numpy(c["seq"]).int().to(cuda)
    got = e.sample_loop(tok, steps, steps, {"name": "random", "temperature": 1.0}, cond=cond, seed=13, first_layout=21,
                        use_graph=True)[0].cpu()
    e.close()
    m = torch.from_numpy(c["mask"])
    assert torch.equal(got.long()[m], torch.from_numpy(c["seq"])[m]), "strong-masked tokens changed"
    assert (got != spec.mask_id).all()


def test_fused_loop_t200_spans_two_launches_vs_oracle(cuda):
    """BASELINE config 5's step count: a T = 200 schedule (alpha_schedule over 200 steps, AdaLN table of 200 timesteps).
    The loop kernel takes its timesteps in the kernel arguments, 128 per launch (kStackLoopMaxSteps): 200 steps = two
    launches, the second starting from the tokens of the first with the RNG counter word at 128.  Checked: the loop ==
    the same steps one ldm_sample_step at a time (same kernel, same state: bit-exact, across the launch seam); the
    fast loop and the exact-mode loop against the oracle's free-running loop on identical Philox uniforms."""
    import dataclasses

    from layout_dm_amd.binding import Engine

    spec = dataclasses.replace(SP.RICO25, name="rico25_t200", n_step=200)
    sd = synth.synth_state_dict(spec, seed=WEIGHT_SEED, perturb=True)
    W = R.as_torch_weights(sd)
    B = 3
    steps = R.timestep_list(spec.n_step, 200)
    assert len(steps) == 200
    cfg = {"name": "random", "temperature": 1.0}
    ref = torch.stack(R.sample_loop(W, spec, B, cfg, seed=31, first_layout=70, get_intermediate_results=True)).int()
    assert ref.shape[0] == 200
    for precision, bound in (("fast", 5e-3), ("exact", 1e-3)):
        e = Engine(n_category=spec.n_category, n_bin=spec.n_bin, max_elem=spec.max_elem, d_model=spec.d_model,
                   n_head=spec.n_head, d_ff=spec.d_ff, n_layer=spec.n_layer, n_step=spec.n_step, precision=precision,
                   max_batch=8)
        e.load_state_dict(sd)
        tok = torch.full((B, spec.seq_len), spec.mask_id, dtype=torch.int32, device=cuda)
        out, inter = e.sample_loop(tok, steps, steps, cfg, seed=31, first_layout=70, intermediates=True, use_graph=True)
        inter = inter.cpu()
        assert torch.equal(inter[-1], out.cpu())
        if precision == "fast":
            cur = torch.full((B, spec.seq_len), spec.mask_id, dtype=torch.int32, device=cuda)
            for i, t in enumerate(steps):
                cur = e.sample_step(cur, t, cfg, seed=31, first_layout=70, step=i)
                if i in (0, 1, 126, 127, 128, 129, 198, 199):
                    assert torch.equal(cur.cpu(), inter[i]), i
        e.close()
        diff = inter != ref
        frac = diff.float().mean().item()
        print(f"[T=200 loop / {precision}] tokens differing from the oracle on identical uniforms: {int(diff.sum())}/"
              f"{diff.numel()} = {frac:.2e}; first 130 steps: {int(diff[:130].sum())}")
        assert frac <= bound, frac
        assert (out.cpu() != spec.mask_id).all()


@pytest.mark.parametrize("sampler", ["deterministic", "random", "top_p", "top_k", "gumbel"])
def test_fused_loop_equals_per_step_path(cuda, monkeypatch, sampler):
    """The shipping fast path — the whole reverse loop of a layout in ONE launch (kernels_stack.hip HEAD == 2: tokens in
    LDS, embedding gathered into the accumulators, the step's tail on 16-lane groups behind the vocabulary head) —
    against the per-step path it replaced (LDM_STACK_LOOP=0: stack kernel -> logits in HBM -> posterior_sample_k, captured
    in hipGraphs), which the reference-pinned tests above cover stage by stage.  Same Philox uniforms, same arithmetic
    per class (csrc/ldm_post_token.h is the one source of both tails); what differs is the summation order of the
    log-softmax, so tokens may differ only where that moves an argmax tie or a CDF edge.  Free and cond=c, with the
    intermediates of every step; also: a loop == the same steps one call at a time (ldm_sample_step)."""
    from layout_dm_amd.binding import Engine

    spec = SP.RICO25
    sd = synth.synth_state_dict(spec, seed=WEIGHT_SEED, perturb=True)
    steps = R.timestep_list(spec.n_step, 100)
    cfg = {"name": sampler, "temperature": 1.0, "top_p": 0.9, "top_k": 5}
    c = synth.synth_cond_c(spec, 300, seed=3)
    outs = {}
    for loop in ("0", "1"):
        monkeypatch.setenv("LDM_DEV", "1")
        monkeypatch.setenv("LDM_STACK_LOOP", loop)
        e = Engine(n_category=spec.n_category, n_bin=spec.n_bin, max_elem=spec.max_elem, d_model=spec.d_model,
                   n_head=spec.n_head, d_ff=spec.d_ff, n_layer=spec.n_layer, n_step=spec.n_step, precision="fast",
                   max_batch=512)
        e.load_state_dict(sd)
        tok = torch.full((300, spec.seq_len), spec.mask_id, dtype=torch.int32, device=cuda)
        free, inter = e.sample_loop(tok, steps, steps, cfg, seed=5, first_layout=9, intermediates=True, use_graph=True)
        free, inter = free.cpu(), inter.cpu()
        assert torch.equal(inter[-1], free)
        cond = {"seq": c["seq"], "mask": c["mask"], "type": "c"}
        tok = torch.from_numpy(c["seq"]).int().to(cuda)
        cnd = e.sample_loop(tok, steps, steps, cfg, cond=cond, seed=5, first_layout=9, use_graph=True)[0].cpu()
        if loop == "1":  # one launch per step == one launch for the loop (bit-exact: same kernel, same state)
            cur = torch.full((300, spec.seq_len), spec.mask_id, dtype=torch.int32, device=cuda)
            for i, t in enumerate(steps[:12]):
                cur = e.sample_step(cur, t, cfg, seed=5, first_layout=9, step=i)
                assert torch.equal(cur.cpu(), inter[i]), i
        e.close()
        outs[loop] = (free, cnd, inter)
    for a, b in zip(outs["0"][:2], outs["1"][:2]):
        frac = (a != b).float().mean().item()
        print(f"[fused loop vs per-step path / {sampler}] final tokens differing: {frac:.2e}")
        assert frac <= 2e-3
        assert (b != spec.mask_id).all()
    # the first steps, before a moved draw can compound: the two paths agree to the draw
    early = (outs["0"][2][:10] != outs["1"][2][:10]).float().mean().item()
    assert early <= 2e-4, early
    m = torch.from_numpy(c["mask"])
    assert torch.equal(outs["1"][1].long()[m], torch.from_numpy(c["seq"])[m]), "strong-masked tokens changed"


@pytest.mark.parametrize("sampler", ["deterministic", "random", "top_p", "top_k", "gumbel"])
def test_strong_mask_shortcut_is_an_identity(cuda, monkeypatch, sampler):
    """Strong-masked positions skip the posterior and the draw and take their conditioned token (csrc/ldm_post_token.h
    strong_shortcut): the 16-lane-group kernel (with the shortcut) against the wavefront-per-token kernel (LDM_POST_WAVE=1:
    full vocabulary, no shortcut, every position goes through posterior + draw) — bit-exact tokens over a cond=c loop,
    every sampler, exact numerics (the per-step path, where both kernels are selectable)."""
    spec = SP.PUBLAYNET
    e = engine("publaynet", "exact", max_batch=64)
    c = synth.synth_cond_c(spec, 64, seed=11)
    cond = {"seq": c["seq"], "mask": c["mask"], "type": "c"}
    steps = R.timestep_list(spec.n_step, 20)
    cfg = {"name": sampler, "temperature": 1.0, "top_p": 0.9, "top_k": 5}
    outs = []
    for wave in ("0", "1"):
        monkeypatch.setenv("LDM_DEV", "1")
        monkeypatch.setenv("LDM_POST_WAVE", wave)
        tok = torch.from_numpy(c["seq"]).int().to(cuda)
        outs.append(e.sample_loop(tok, steps, steps, cfg, cond=cond, seed=4, first_layout=2, intermediates=True,
                                  use_graph=False)[1].cpu().clone())
    assert torch.equal(outs[0], outs[1])
    m = torch.from_numpy(c["mask"])
    assert torch.equal(outs[0][-1].long()[m], torch.from_numpy(c["seq"])[m])


@pytest.mark.parametrize("precision", ["exact", "split", "fast"])
def test_full_batch_512_one_step_vs_oracle(cuda, precision):
    """Teacher-forced single step at B=512 (M=64000 rows) against the oracle on CPU."""
    spec, W = weights("rico25")
    B = 512
    e = engine("rico25", precision, max_batch=B)
    g = torch.Generator().manual_seed(5)
    t = 30
    tokens = torch.empty(B, spec.seq_len, dtype=torch.long)
    for a in range(spec.n_attr):
        ids = torch.as_tensor(spec.full_ids(a))
        tokens[:, a::spec.n_attr] = ids[torch.randint(0, len(ids) - 1, (B, spec.max_elem), generator=g)]
    tokens[torch.rand(B, spec.seq_len, generator=g) < 0.3] = spec.mask_id
    ref_next, ref_logits, ref_logp = R.single_step(W, spec, tokens, t, {"name": "deterministic"}, return_all=True)
    logits = e.denoise_logits(tokens.int(), t).cpu()
    rel = _rel(logits, ref_logits)
    nxt = e.sample_step(tokens.int(), t, {"name": "deterministic"}).cpu().long()
    mism = (nxt != ref_next)
    # a token may differ only where the oracle's own top-2 margin is inside the logits tolerance
    top2 = ref_logp.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1])
    print(f"[B=512/{precision}] logits rel err {rel:.3e}; greedy token mismatches {int(mism.sum())}/{mism.numel()}")
    assert rel <= LOGIT_REL_TOL[precision]
    if precision == "exact":
        assert int(mism.sum()) == 0 or margin[mism].max().item() < 1e-4
    else:
        assert mism.float().mean().item() < 5e-3
        assert (margin[mism] < 0.05).all()


@pytest.mark.parametrize("precision", ["split", "fast"])
def test_logits_bitwise_repeatable(cuda, precision):
    """The same denoiser pass, 25 times, on a full chunk (256 layouts) and on a ragged one (37): bit-identical logits every time.
    The hand-pipelined kernels order their LDS reads, DMA pieces and MFMAs with counted waits the compiler cannot check — a
    wait that is one operation short shows up as a timing-dependent low-order difference long before it shows up as a parity
    failure (r05: a discarded build of kernels_lngemm.hip was 6e-5 off and not repeatable while the tolerance tests passed)."""
    spec, _ = weights("rico25")
    for B in (256, 37):
        e = engine("rico25", precision, max_batch=256)
        g = torch.Generator().manual_seed(11 + B)
        tokens = torch.empty(B, spec.seq_len, dtype=torch.long)
        for a in range(spec.n_attr):
            ids = torch.as_tensor(spec.full_ids(a))
            tokens[:, a::spec.n_attr] = ids[torch.randint(0, len(ids) - 1, (B, spec.max_elem), generator=g)]
        tokens[torch.rand(B, spec.seq_len, generator=g) < 0.5] = spec.mask_id
        tok = tokens.int().to(cuda)
        first = e.denoise_logits(tok, 47).clone()
        assert bool(torch.isfinite(first).all())
        for i in range(24):
            again = e.denoise_logits(tok, 47)
            assert torch.equal(again, first), f"{precision} B={B}: repetition {i + 1} differs by {(again - first).abs().max().item():.3e}"


@pytest.mark.parametrize("level", ["0", "1", "2", "3"])
def test_split_mode_alternative_structures_agree(cuda, monkeypatch, level):
    """The split mode's other launch structures behind LDM_X3_LNGEMM (dev knob) against the default one and the oracle: 0 = the r04
    structure (LayerNorm launches + tiled GEMMs), 1 = the row-resident kernels without a GEMM prologue, 2 / 3 = out_proj as a GEMM
    prologue too / only (default, level 4: linear2 only).  Same numerics class (fp16 x 3, fp32 accumulation): logits within the split mode's tolerance of the oracle and within
    2e-6 of the default structure; greedy step identical.  Full chunk and ragged batch (the last workgroup's rows are partly masked)."""
    from oracle import restatement as R

    spec, W = weights("rico25")
    for B in (256, 37):
        g = torch.Generator().manual_seed(3 + B)
        tokens = torch.empty(B, spec.seq_len, dtype=torch.long)
        for a in range(spec.n_attr):
            ids = torch.as_tensor(spec.full_ids(a))
            tokens[:, a::spec.n_attr] = ids[torch.randint(0, len(ids) - 1, (B, spec.max_elem), generator=g)]
        tokens[torch.rand(B, spec.seq_len, generator=g) < 0.5] = spec.mask_id
        tok = tokens.int().to(cuda)
        monkeypatch.delenv("LDM_X3_LNGEMM", raising=False)
        monkeypatch.delenv("LDM_DEV", raising=False)
        e0 = engine("rico25", "split", max_batch=256)
        base = e0.denoise_logits(tok, 33).cpu()
        nxt0 = e0.sample_step(tok, 33, {"name": "deterministic"}).cpu()
        monkeypatch.setenv("LDM_DEV", "1")
        monkeypatch.setenv("LDM_X3_LNGEMM", level)
        from layout_dm_amd.binding import Engine

        e1 = Engine(n_category=spec.n_category, n_bin=spec.n_bin, max_elem=spec.max_elem, d_model=spec.d_model, n_head=spec.n_head,
                    d_ff=spec.d_ff, n_layer=spec.n_layer, n_step=spec.n_step, precision="split", max_batch=256)   # (not cached: built under the knob)
        e1.load_state_dict(synth.synth_state_dict(spec, seed=WEIGHT_SEED, perturb=True))
        assert f"LDM_X3_LNGEMM={level}" in e1.describe()["knobs"]
        got = e1.denoise_logits(tok, 33).cpu()
        nxt1 = e1.sample_step(tok, 33, {"name": "deterministic"}).cpu()
        e1.close()
        ref = R.denoiser_logits(W, spec, tokens[:8], 33)
        scale = ref.abs().max().item()
        err_ref = (got[:8] - ref).abs().max().item() / scale
        err_base = (got - base).abs().max().item() / scale
        print(f"[split structure {level}, B={B}] vs oracle {err_ref:.2e}, vs the default structure {err_base:.2e}")
        assert err_ref <= 5e-5 and err_base <= 2e-6
        assert torch.equal(nxt1, nxt0)


# ----------------------------------------------------------------------------- drop-in layer
class _MockTokenizer:
    """Duck-types the properties of the reference's LayoutSequenceTokenizer that LayoutDM reads
    (helpers/layout_tokenizer.py:123-186); decode() is the caller's own host code."""

    def __init__(self, spec):
        self.spec = spec
        self.N_category, self.N_bbox_per_var = spec.n_category, spec.n_bin
        self.max_seq_length, self.N_var_per_element = spec.max_elem, spec.n_attr
        self.N_total, self.max_token_length = spec.n_class, spec.seq_len
        self.var_names = ["c", "x", "y", "w", "h"]
        self.special_tokens = ["pad", "mask"]

    def id_to_name(self, i):
        return {self.spec.pad_id: "pad", self.spec.mask_id: "mask"}[i]

    def name_to_id(self, n):
        return {"pad": self.spec.pad_id, "mask": self.spec.mask_id}[n]

    def decode(self, ids):
        return {"ids": ids, "mask": ids[:, ::5] != self.spec.pad_id}


BACKBONE_CFG = {"_target_": "trainer.models.transformer_utils.TransformerEncoder",
                "encoder_layer": {"_target_": "trainer.models.transformer_utils.Block", "d_model": 512, "nhead": 8,
                                  "dim_feedforward": 2048, "dropout": 0.0, "batch_first": True, "norm_first": True,
                                  "timestep_type": "adalayernorm", "diffusion_step": 100},
                "num_layers": 4}


def test_dropin_layoutdm_class(cuda, golden_dir):
    """layout_dm_amd.layoutdm.LayoutDM honours the contract test.py relies on (SURVEY §8b Seam 1)."""
    from layout_dm_amd.layoutdm import LayoutDM

    spec, W = weights("rico25")
    m = LayoutDM(backbone_cfg=BACKBONE_CFG, tokenizer=_MockTokenizer(spec), q_type="constrained",
                 precision="exact", max_batch=8).to("cuda")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(spec, seed=WEIGHT_SEED, perturb=True).items()})
    m.eval()
    cfg = {"name": "deterministic", "num_timesteps": 100}
    out = m.sample(batch_size=4, cond=None, sampling_cfg=cfg, cond_type="unconditional")
    g = np.load(os.path.join(golden_dir, "rico25_uncond_greedy_loop.npz"))
    assert torch.equal(out["ids"], torch.from_numpy(g["states_after"][-1].astype(np.int64)))
    inter = m.model.sample(batch_size=4, sampling_cfg=cfg, get_intermediate_results=True)  # notebook usage
    assert len(inter) == 100 and torch.equal(inter[-1], out["ids"])
    # stochastic runs are reproducible under torch.manual_seed like the reference's set_seed()
    rc = {"name": "random", "temperature": 1.0, "num_timesteps": 20}
    torch.manual_seed(5)
    a = m.sample(batch_size=3, sampling_cfg=rc)["ids"]
    torch.manual_seed(5)
    b = m.sample(batch_size=3, sampling_cfg=rc)["ids"]
    assert torch.equal(a, b)
    with pytest.raises(NotImplementedError):
        m.train()


# ----------------------------------------------------------------------------- cond = relation
def _random_graph(spec, B, gen, edge_ratio=0.6):
    """Random relation graphs in the reference's format (data/util.py:128-177): node 0 of every layout = canvas."""
    ys, eis, eas, bts, seqs, off = [], [], [], [], [], 0
    for b in range(B):
        n = int(torch.randint(0 if b == 1 else 2, 12, (1,), generator=gen))   # layout 1 may be empty
        lab = torch.randint(0, spec.n_category, (n,), generator=gen)
        ys.append(torch.cat([torch.zeros(1, dtype=torch.long), lab + 1]))
        for i in range(n + 1):
            for j in range(i + 1, n + 1):
                if torch.rand(1, generator=gen).item() < edge_ratio:
                    size = int(torch.randint(0, 4, (1,), generator=gen))
                    loc = int(torch.randint(4, 10, (1,), generator=gen))
                    eis.append((off + i, off + j))
                    eas.append((1 << size) | (1 << loc))
        bts.append(torch.full((n + 1,), b, dtype=torch.long))
        off += n + 1
        seq = torch.full((spec.seq_len,), spec.pad_id, dtype=torch.long)
        seq[: n * spec.n_attr] = spec.mask_id
        seq[0: n * spec.n_attr: spec.n_attr] = lab
        seqs.append(seq)
    ei = torch.tensor(eis, dtype=torch.long).t().contiguous() if eis else torch.zeros((2, 0), dtype=torch.long)
    return {"y": torch.cat(ys), "edge_index": ei, "edge_attr": torch.tensor(eas, dtype=torch.long),
            "batch": torch.cat(bts)}, torch.stack(seqs)


def _rel_close(a, ref, start, rtol):
    """fp32 agreement of two SGD results: error relative to the size of the update (the steps are O(10..1e4) while the
    results may cancel back to O(1e-2)); analytic vs autograd gradients measured 3e-7 .. 1.5e-6 on the CPU."""
    scale = max(1.0, (ref - start).abs().max().item())
    return (a - ref).abs().max().item() <= rtol * scale


def test_relation_update_vs_reference_and_oracle(cuda, golden_dir):
    """kernels_relation.hip (analytic gradient) == logit_adjustment.update (autograd): reference-produced golden at the
    default hyper-parameters, then random graphs / moderate step sizes against the oracle restatement."""
    spec = SP.RICO25
    e = engine("rico25", "exact")
    g = np.load(os.path.join(golden_dir, "rico25_relation.npz"))
    graph = {k: torch.from_numpy(g[k]) for k in ("y", "edge_index", "edge_attr", "batch")}
    seq = torch.from_numpy(g["cond_seq"].astype(np.int64))
    B = seq.shape[0]
    rel = e.make_relation(graph, g["centres"], g["canvas_bins"], float(g["lr"]), int(g["num_update"]), B)
    for t in (50, 5):
        out = e.relation_update(torch.from_numpy(g["logp_in"]).to(cuda).contiguous(), seq, rel, t).cpu()
        assert _rel_close(out, torch.from_numpy(g[f"logp_out_t{t}"]), torch.from_numpy(g["logp_in"]), 1e-5), t
    # random graphs (one layout without elements), several step sizes, incl. a graph-free batch
    gen = torch.Generator().manual_seed(17)
    centres = np.stack([np.linspace(0, 1 - 1 / 32, 32), np.linspace(0, 1 - 1 / 32, 32), np.linspace(1 / 32, 1, 32),
                        np.linspace(1 / 32, 1, 32)])
    bins = [16, 16, 31, 31]
    for B, lr, nup in ((5, 3e4, 3), (3, 5e3, 1), (4, 2e5, 4)):
        graph, seq = _random_graph(spec, B, gen)
        logp = torch.log_softmax(1.5 * torch.randn(B, spec.n_class, spec.seq_len, generator=gen), dim=1).clamp(-70, 0)
        ref = R.relation_update(spec, logp, seq, graph, centres, bins, lr, nup, 40)
        rel = e.make_relation(graph, centres, bins, lr, nup, B)
        out = e.relation_update(logp.to(cuda).contiguous(), seq, rel, 40).cpu()
        assert (ref != logp).any()
        assert _rel_close(out, ref, logp, 1e-5), (B, lr, nup, (out - ref).abs().max().item())
    empty = {"y": graph["y"], "edge_index": torch.zeros((2, 0), dtype=torch.long), "edge_attr": torch.zeros(0, dtype=torch.long),
             "batch": graph["batch"]}
    rel = e.make_relation(empty, centres, bins, 3e6, 3, B)
    out = e.relation_update(logp.to(cuda).contiguous(), seq, rel, 40).cpu()
    assert torch.equal(out, logp)


class _RelBboxTokenizer:
    """BboxTokenizer attributes used by the relation plan (bbox_tokenizer.py:28-115): linear bins."""
    shared_bbox_vocab, bbox_quantization = "x-y-w-h", "linear"
    var_names = ["x", "y", "w", "h"]

    def __init__(self, n_bin):
        d = 1.0 / n_bin
        mk = lambda a: type("M", (), {"cluster_centers_": a.reshape(-1, 1)})()
        self.clustering_models = {f"x-{n_bin}": mk(np.linspace(0, 1 - d, n_bin)), f"y-{n_bin}": mk(np.linspace(0, 1 - d, n_bin)),
                                  f"w-{n_bin}": mk(np.linspace(d, 1, n_bin)), f"h-{n_bin}": mk(np.linspace(d, 1, n_bin))}
        self.n_bin = n_bin

    def encode(self, bbox):  # bbox_tokenizer.py:84-115, linear
        d = 1.0 / self.n_bin
        q = torch.zeros_like(bbox)
        q[..., :2] = torch.clamp(bbox[..., :2], 0.0, 1.0 - d)
        q[..., 2:] = torch.clamp(bbox[..., 2:], d, 1.0) - d
        idx = (self.n_bin * q).round().long()
        return idx + torch.arange(4) * self.n_bin


def test_relation_sampling_loop_hip_update_equals_oracle_update(cuda):
    """sample_with_relation end to end: the HIP logit adjustment inside the split-step loop gives the same greedy
    tokens as the same loop driven by the oracle's autograd update."""
    from layout_dm_amd.diffusion import HipMaskAndReplaceDiffusion
    from layout_dm_amd.relation import sample_with_relation

    spec = SP.RICO25
    sd = synth.synth_state_dict(spec, seed=WEIGHT_SEED, perturb=True)
    m = HipMaskAndReplaceDiffusion(n_category=spec.n_category, precision="exact", max_batch=4)
    m.load_state_dict(sd)
    tok = _MockTokenizer(spec)
    tok.bbox_tokenizer = _RelBboxTokenizer(spec.n_bin)
    gen = torch.Generator().manual_seed(23)
    B = 3
    graph, seq = _random_graph(spec, B, gen)
    gb = type("G", (), dict(graph, to=lambda self, *_a, **_k: self))()
    cond = {"seq": seq, "mask": seq != spec.mask_id, "type": "relation", "batch_w_canvas": gb}
    cfg = {"name": "deterministic", "num_timesteps": 12, "relation_lambda": 2e4, "relation_mode": "average",
           "relation_tau": 1.0, "relation_num_update": 2}
    centres = np.stack([tok.bbox_tokenizer.clustering_models[f"{k}-32"].cluster_centers_.reshape(-1) for k in "xywh"])

    def oracle_update(t, cond, model_log_prob, tokenizer, sampling_cfg):
        out = R.relation_update(spec, model_log_prob.detach().cpu(), seq, graph, centres, [16, 16, 31, 31],
                                cfg["relation_lambda"], cfg["relation_num_update"], t)
        return out.to(model_log_prob.device)

    a = sample_with_relation(m, B, cond, cfg, tok, seed=1)                            # fused: inside the hipGraph
    b = sample_with_relation(m, B, cond, cfg, tok, update_fn=oracle_update, seed=1)    # split-step, oracle autograd
    assert (a != b).float().mean().item() <= 0.01
    assert (a[:, 0::5] == seq[:, 0::5]).all()  # conditioned categories survive

    # graph path == eager path == split-step path (every stage launched separately through the parity hooks, with the
    # HIP logit adjustment as the update function): bit-exact tokens, deterministic AND stochastic sampler
    from layout_dm_amd.relation import hip_relation_plan

    eng = m.engine
    plan = hip_relation_plan(eng, {"batch_w_canvas": gb}, cfg, tok, B)

    def hip_update(t, cond, model_log_prob, tokenizer, sampling_cfg):
        return eng.relation_update(model_log_prob.contiguous(), seq, plan, t)

    for scfg in (cfg, dict(cfg, name="random", temperature=1.0), dict(cfg, name="top_p", top_p=0.9, temperature=1.0)):
        fused = sample_with_relation(m, B, cond, scfg, tok, seed=5)
        m.use_graph = False
        eager = sample_with_relation(m, B, cond, scfg, tok, seed=5)
        m.use_graph = True
        split = sample_with_relation(m, B, cond, scfg, tok, update_fn=hip_update, seed=5)
        assert torch.equal(fused, eager), scfg["name"]
        assert torch.equal(fused, split), scfg["name"]
    inter = sample_with_relation(m, B, cond, cfg, tok, seed=5, get_intermediate_results=True)
    assert len(inter) == 12 and torch.equal(inter[-1], sample_with_relation(m, B, cond, cfg, tok, seed=5))
    # relation_mode="gumbel" has no implementation (and no reference fallback)
    with pytest.raises(NotImplementedError):
        sample_with_relation(m, B, cond, dict(cfg, relation_mode="gumbel"), tok, seed=1)
    # more layouts than max_batch: windows of the same graph (max_batch = 4 here)
    B2 = 6
    graph2, seq2 = _random_graph(spec, B2, gen)
    gb2 = type("G", (), dict(graph2, to=lambda self, *_a, **_k: self))()
    cond2 = {"seq": seq2, "mask": seq2 != spec.mask_id, "type": "relation", "batch_w_canvas": gb2}
    big = sample_with_relation(m, B2, cond2, cfg, tok, seed=9)
    m6 = HipMaskAndReplaceDiffusion(n_category=spec.n_category, precision="exact", max_batch=8)
    m6.load_state_dict(sd)
    assert torch.equal(big, sample_with_relation(m6, B2, cond2, cfg, tok, seed=9))


# ----------------------------------------------------------------------------- q_type = vanilla
@pytest.mark.parametrize("precision", ["exact", "fast"])
def test_vanilla_q_type_vs_reference_golden(cuda, golden_dir, precision):
    """VanillaMaskAndReplaceDiffusion (categorical_diffusion/vanilla.py) through the same kernels: posterior vs the
    reference's q_posterior on unrestricted token states, greedy step on every state of a reference trajectory, and a
    seeded stochastic loop vs the oracle with the same Philox uniforms."""
    from layout_dm_amd.binding import Engine

    spec = SP.RICO25
    sd = synth.synth_state_dict(spec, seed=WEIGHT_SEED, perturb=True, q_type="vanilla")
    W = R.as_torch_weights(sd)
    e = Engine(n_category=spec.n_category, precision=precision, max_batch=8, q_type="vanilla")
    e.load_state_dict(sd)
    g = np.load(os.path.join(golden_dir, "rico25_vanilla.npz"))
    for t in g["ts"]:
        t = int(t)
        tokens = torch.from_numpy(g[f"tokens_{t}"].astype(np.int64))
        ref_post = torch.from_numpy(g[f"post_{t}"])
        logits = e.denoise_logits(tokens.int(), t)
        post = e.posterior(logits, tokens.int(), t).cpu()
        tol = 2e-4 if precision == "exact" else 5e-2  # log-probabilities; logits themselves carry <=1e-3 rel in fast
        assert (post - ref_post).abs().max().item() <= tol
    bad, n, worst = _traj_check(e, g, {"name": "deterministic"}, prefix="traj_")
    _assert_traj("vanilla", precision, bad, n, worst)
    if precision == "exact":
        cfg = {"name": "random", "temperature": 1.0, "num_timesteps": 20}
        tk = torch.full((3, spec.seq_len), spec.mask_id, dtype=torch.int32)
        from layout_dm_amd.diffusion import timestep_schedule

        tm, tp = timestep_schedule(spec.n_step, 20)
        out, _ = e.sample_loop(tk, tm, tp, cfg, seed=77, first_layout=5)
        ref = R.sample_loop(W, spec, 3, cfg, seed=77, first_layout=5, q_type="vanilla")
        frac = (out.cpu().long() != ref).float().mean().item()
        assert frac <= 0.02, frac  # inverse-CDF draws can flip only where a uniform lands within fp32 noise of a bin edge
    e.close()
    with pytest.raises(NotImplementedError):
        Engine(n_category=spec.n_category, q_type="single")


# ----------------------------------------------------------------------------- result packaging (decode)
@pytest.mark.parametrize("ds", ["rico25", "publaynet"])
def test_decode_vs_reference_tokenizer_golden(cuda, golden_dir, ds):
    """kernels_decode.hip == the reference's LayoutSequenceTokenizer.decode (bit-exact boxes, labels, masks)."""
    spec = SP.SPECS[ds]
    e = engine(ds, "exact")
    g = np.load(os.path.join(golden_dir, f"{ds}_decode.npz"))
    tokens = torch.from_numpy(g["tokens"]).int()
    lin = e.decode(tokens)
    assert lin["bbox"].dtype == torch.float32 and lin["label"].dtype == torch.int64 and lin["mask"].dtype == torch.bool
    assert np.array_equal(lin["bbox"].cpu().numpy(), g["linear_bbox"])
    assert np.array_equal(lin["label"].cpu().numpy(), g["linear_label"])
    assert np.array_equal(lin["mask"].cpu().numpy(), g["linear_mask"])
    km = e.decode(tokens, centres=torch.from_numpy(g["centres"]))
    assert km["bbox"].dtype == torch.float64
    assert np.array_equal(km["bbox"].cpu().numpy(), g["kmeans_bbox"])
    assert np.array_equal(km["label"].cpu().numpy(), g["kmeans_label"])
    assert np.array_equal(km["mask"].cpu().numpy(), g["kmeans_mask"])
    # larger ragged batch vs the oracle restatement, incl. out-of-vocabulary / negative ids and an empty batch
    gen = torch.Generator().manual_seed(3)
    big = torch.randint(-3, spec.n_class + 3, (777, spec.seq_len), generator=gen).int()
    big[::3, 0::5] = torch.randint(0, spec.n_category, big[::3, 0::5].shape, generator=gen).int()
    for a in range(1, 5):
        ids = torch.as_tensor(spec.full_ids(a)[:-2])
        big[::3, a::5] = ids[torch.randint(0, len(ids), big[::3, a::5].shape, generator=gen)].int()
    ref = R.decode_layouts(spec, big)
    out = e.decode(big)
    assert torch.equal(out["bbox"].cpu(), ref["bbox"]) and torch.equal(out["label"].cpu(), ref["label"])
    assert torch.equal(out["mask"].cpu(), ref["mask"]) and 0.2 < ref["mask"].float().mean() < 0.5
    empty = e.decode(torch.empty((0, spec.seq_len), dtype=torch.int32))
    assert empty["bbox"].shape == (0, spec.max_elem, 4) and empty["mask"].shape == (0, spec.max_elem)


class _LinearBboxTokenizer:
    """The attributes of the reference's BboxTokenizer (helpers/bbox_tokenizer.py:28-83) that the drop-in inspects."""
    shared_bbox_vocab, bbox_quantization = "x-y-w-h", "linear"
    var_names = ["x", "y", "w", "h"]
    _var_order = ["x", "y", "w", "h"]


def test_dropin_sample_decodes_on_device(cuda, golden_dir):
    """LayoutDM.sample -> {"bbox","label","mask"} CPU tensors (layoutdm.py:77-88) with the decode on the GPU."""
    from layout_dm_amd.layoutdm import LayoutDM

    spec, _ = weights("rico25")
    tok = _MockTokenizer(spec)
    tok.bbox_tokenizer = _LinearBboxTokenizer()
    m = LayoutDM(backbone_cfg=BACKBONE_CFG, tokenizer=tok, q_type="constrained", precision="exact", max_batch=8)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(spec, seed=WEIGHT_SEED, perturb=True).items()})
    out = m.sample(batch_size=4, cond=None, sampling_cfg={"name": "deterministic", "num_timesteps": 100})
    g = np.load(os.path.join(golden_dir, "rico25_uncond_greedy_loop.npz"))
    ref = R.decode_layouts(spec, g["states_after"][-1])
    assert set(out) == {"bbox", "label", "mask"} and all(not v.is_cuda for v in out.values())
    assert out["bbox"].shape == (4, 25, 4) and out["bbox"].dtype == torch.float32
    assert torch.equal(out["bbox"], ref["bbox"]) and torch.equal(out["label"], ref["label"])
    assert torch.equal(out["mask"], ref["mask"])


def test_relation_split_step_equals_fused_with_identity_update(cuda):
    """cond=relation split-step path (denoise -> posterior -> update_fn -> draw) with an identity
    update_fn must reproduce the fused kernel path of cond=c (same strong mask + PAD disable)."""
    from layout_dm_amd.diffusion import HipMaskAndReplaceDiffusion
    from layout_dm_amd.relation import sample_with_relation

    spec = SP.PUBLAYNET
    m = HipMaskAndReplaceDiffusion(n_category=spec.n_category, precision="exact", max_batch=4)
    m.load_state_dict(synth.synth_state_dict(spec, seed=WEIGHT_SEED, perturb=True))
    c = synth.synth_cond_c(spec, 4, seed=0)
    cond = {"seq": torch.from_numpy(c["seq"]), "mask": torch.from_numpy(c["mask"]), "type": "c"}
    cfg = {"name": "random", "temperature": 1.0, "num_timesteps": 25}
    fused = m.sample(batch_size=4, cond=cond, sampling_cfg=cfg, seed=77)
    ident = lambda t, cond, model_log_prob, tokenizer, sampling_cfg: model_log_prob
    rel = dict(cond, type="relation")
    split = sample_with_relation(m, 4, rel, cfg, _MockTokenizer(spec), update_fn=ident, seed=77)
    assert torch.equal(fused, split)
    assert (split[cond["mask"]] == cond["seq"][cond["mask"]]).all()


def test_loop_time_difference_and_cond_graph_reuse(cuda):
    """(a) time_difference > 0 (base.py:218-225) through the host schedule == oracle; (b) conditional
    loops replayed from ONE captured graph stay correct when the caller's cond tensors move."""
    from layout_dm_amd.diffusion import HipMaskAndReplaceDiffusion

    spec, W = weights("publaynet")
    m = HipMaskAndReplaceDiffusion(n_category=spec.n_category, precision="exact", max_batch=4)
    m.load_state_dict(synth.synth_state_dict(spec, seed=WEIGHT_SEED, perturb=True))
    cfg = {"name": "deterministic", "num_timesteps": 20, "time_difference": 0.1}
    out = m.sample(batch_size=3, sampling_cfg=cfg)
    ref = R.sample_loop(W, spec, 3, cfg)
    assert torch.equal(out, ref)
    cfg = {"name": "deterministic", "num_timesteps": 10}
    outs = []
    for seed in (0, 1, 2):
        c = synth.synth_cond_c(spec, 4, seed=seed)
        cond = {"seq": torch.from_numpy(c["seq"]), "mask": torch.from_numpy(c["mask"]), "type": "c"}
        junk = torch.empty(1000 * (seed + 1), device="cuda")  # perturb the allocator between calls
        got = m.sample(batch_size=4, cond=cond, sampling_cfg=cfg)
        want = R.sample_loop(W, spec, 4, cfg, cond={"seq": c["seq"], "mask": c["mask"], "type": "c"})
        assert torch.equal(got, want), seed
        del junk
        outs.append(got)
    assert not torch.equal(outs[0], outs[1])
