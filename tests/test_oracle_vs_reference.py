"""CPU, build container only (skipped where /root/reference is absent, e.g. on the GPU box): the committed golden
fixtures are exactly what the REAL reference produces today, and the oracle restatement agrees with the live
reference on fresh inputs.  This is the pin of oracle/restatement.py (the reference ships no tests or
golden vectors of its own)."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_harness as rh

pytestmark = pytest.mark.skipif(not rh.reference_available(), reason="reference tree not present")


def test_golden_fixtures_regenerate_bit_identically(tmp_path, golden_dir):
    """python -m oracle.make_golden into a scratch directory == tests/golden/*.npz, array for array."""
    from oracle import make_golden

    out = str(tmp_path / "golden")
    make_golden.main(out_dir=out)
    # (rico25_mid_reference_samples.npz, rico25_fitted.npz and rico25_b512_reference_loops.npz have their own generators and their own
    #  regeneration tests below)
    names = sorted(f for f in os.listdir(golden_dir) if f.endswith(".npz") and f not in ("rico25_mid_reference_samples.npz", "rico25_fitted.npz",
                                                                                           "rico25_b512_reference_loops.npz"))
    assert names == sorted(f for f in os.listdir(out) if f.endswith(".npz"))
    for n in names:
        a, b = np.load(os.path.join(golden_dir, n)), np.load(os.path.join(out, n))
        assert sorted(a.keys()) == sorted(b.keys()), n
        for k in a.keys():
            assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), (n, k)
    for n in (f for f in os.listdir(golden_dir) if f.endswith(".txt")):
        assert open(os.path.join(golden_dir, n)).read() == open(os.path.join(out, n)).read(), n


def test_fitted_fixture_regenerates_from_the_fitted_weights(golden_dir):
    """tests/golden/rico25_fitted.npz (oracle/make_trained_fixture.py): the 30-minute training run is not repeated here, but everything the
    fixture says ABOUT the fitted checkpoint — the reference's logits / posterior at three timesteps, its f32 noise floor, its largest attention
    score, the 100-state trajectory with greedy answers and margins — is recomputed by the live reference from oracle/_fit/rico25_fitted.npz and
    must equal the committed arrays bit for bit (skipped where the 50-MB weight file, a git-ignored build output, is absent)."""
    import hashlib

    import torch

    from oracle import make_trained_fixture as MF
    from oracle import spec as SP

    if not os.path.exists(MF.WEIGHTS):
        pytest.skip("oracle/_fit/rico25_fitted.npz absent (python -m oracle.make_trained_fixture)")
    g = np.load(os.path.join(golden_dir, "rico25_fitted.npz"))
    with open(MF.WEIGHTS, "rb") as fh:
        assert hashlib.sha256(fh.read()).hexdigest() == str(g["weights_sha256"])
    spec = SP.SPECS["rico25"]
    m, tok = rh.build_reference_model("rico25", seed=int(g["train_args"][3]))
    w = np.load(MF.WEIGHTS)
    m.load_state_dict({k.split("model.module.")[-1]: torch.from_numpy(w[k]) for k in w.files})
    m.eval()
    label, bbox, mask = MF.structured_layouts(int(g["train_args"][2]), spec.n_category, int(g["train_args"][3]) + 1)
    seq = tok.encode({"label": label, "bbox": bbox, "mask": mask})["seq"]
    out = MF.goldens(m, spec, seq)
    for k, v in out.items():
        assert np.array_equal(np.asarray(v), g[k]), k


def test_restatement_matches_live_reference_on_fresh_states():
    """Fresh random states (not the committed ones), both datasets: denoiser logits, x0, posterior, greedy step of the
    oracle == the reference's own modules (transformer forward nn_lib.py:191-237, predict_start base.py:127-146,
    q_posterior constrained.py:135-206, _sample_single_step base.py:205-291)."""
    from oracle import restatement as R
    from oracle import spec as SP
    from oracle import synth

    for ds in ("rico25", "publaynet"):
        spec = SP.SPECS[ds]
        m, _tok = rh.build_reference_model(ds, seed=0)
        from trainer.models.categorical_diffusion.util import index_to_log_onehot

        ssd = synth.synth_state_dict(spec, seed=7, perturb=True, prefix="")
        m.load_state_dict({k: torch.from_numpy(v) for k, v in ssd.items()})
        W = R.as_torch_weights(synth.synth_state_dict(spec, seed=7, perturb=True))
        g = torch.Generator().manual_seed(99)
        for t in (97, 55, 12, 0):
            tokens = torch.empty(3, spec.seq_len, dtype=torch.long)
            for a in range(spec.n_attr):
                ids = torch.as_tensor(spec.full_ids(a))
                tokens[:, a::spec.n_attr] = ids[torch.randint(0, len(ids) - 1, (3, spec.max_elem), generator=g)]
            tokens[torch.rand(3, spec.seq_len, generator=g) < t / 99] = spec.mask_id
            tt = torch.full((3,), t, dtype=torch.long)
            with torch.no_grad():
                ref_logits = m.transformer(tokens, timestep=tt)["logits"]
                lz = index_to_log_onehot(tokens, spec.n_class)
                ref_x0 = m.predict_start(lz, tt)
                ref_post = m.q_posterior(ref_x0, lz, tt)
                ref_next = m._sample_single_step(lz, tt, skip_step=0, sampling_cfg=rh.sampling_cfg("deterministic"),
                                                 cond=None).argmax(1)
            logits = R.denoiser_logits(W, spec, tokens, t)
            assert ((logits - ref_logits).abs().max() / ref_logits.abs().max()).item() < 2e-5
            x0 = R.predict_start_from_logits(ref_logits)
            assert torch.equal(x0, ref_x0)
            post = R.q_posterior(W, spec, ref_x0, tokens, t)
            assert (post - ref_post).abs().max().item() == 0.0
            nxt = R.single_step(W, spec, tokens, t, {"name": "deterministic"})
            assert torch.equal(nxt, ref_next)


def test_reference_sample_sets_regenerate(golden_dir):
    """tests/golden/rico25_mid_reference_samples.npz (the two 1 024-layout sample sets of the reference's own sample() behind
    tests/test_fid_vs_reference.py) is what the real reference draws today: the first chunk of 64 layouts of each seed,
    re-generated here, equals the committed tokens bit for bit (the whole file takes ~40 min of CPU: python -m
    oracle.make_reference_samples)."""
    from oracle import make_reference_samples as mrs

    g = np.load(os.path.join(golden_dir, "rico25_mid_reference_samples.npz"))
    new = mrs.generate(n_chunk=1)
    assert g["tokens"].shape == (2, mrs.N_CHUNK * mrs.CHUNK, 125) and tuple(g["seeds"]) == mrs.SEEDS
    assert np.array_equal(new["tokens"], g["tokens"][:, :mrs.CHUNK])
    # two independent draws of a non-degenerate distribution: no [MASK] left, the sets differ
    assert (g["tokens"] != 154).all() and not np.array_equal(g["tokens"][0], g["tokens"][1])


def test_b512_reference_loops_fixture_slice_regenerates(golden_dir):
    """tests/golden/rico25_b512_reference_loops.npz (oracle/make_b512_golden.py: the reference's own full loops at BASELINE config 2's batch, 20 min of
    CPU) is not regenerated whole here; a slice is: the reference's greedy continuation of four of its mid-trajectory states, run live at batch 4,
    must end in the stored tokens — and so must the oracle restatement's.  (Layouts whose smallest top-2 margin is below 1e-3 are left out: the
    reference's own fp32 GEMMs may order their sums differently at batch 4 and at batch 512.)"""
    from oracle import make_b512_golden as MB
    from oracle import make_golden as MG
    from oracle import restatement as R
    from oracle import spec as SP
    from oracle import synth

    path = os.path.join(golden_dir, "rico25_b512_reference_loops.npz")
    g = np.load(path)
    spec = SP.SPECS["rico25"]
    assert g["final_greedy"].shape == (512, spec.seq_len) and g["final_from_mid"].shape == (512, spec.seq_len)
    assert int(g["weight_seed"]) == MG.WEIGHT_SEED and not (g["final_greedy"] == spec.mask_id).any() and not (g["final_from_mid"] == spec.mask_id).any()
    assert (g["mid_state"] == spec.mask_id).mean() > 0.2          # a mid-trajectory state: a good part still masked
    idx = np.nonzero(g["min_margin_from_mid"] >= 1e-3)[0][:4]
    assert len(idx) == 4
    mid = torch.from_numpy(g["mid_state"][idx].astype(np.int64))
    want = torch.from_numpy(g["final_from_mid"][idx].astype(np.int64))
    m, _ = rh.build_reference_model("rico25")
    MG.load_synth(m, spec)
    with torch.no_grad():
        assert torch.equal(MB.greedy_from(m, spec, mid, 49), want)
    W = R.as_torch_weights(synth.synth_state_dict(spec, seed=MG.WEIGHT_SEED, perturb=True))
    tok = mid.clone()
    for t in range(49, -1, -1):
        tok = R.single_step(W, spec, tok, t, {"name": "deterministic"})
    assert torch.equal(tok, want)
