"""GPU: `fast_verified` — greedy decoding on the fp16 engine with the near-tie layouts re-decided by the exact engine
(layout_dm_amd/verified.py, ldm_set_tie_report / ldm_get_tie_flags) — is BIT-EXACT against the reference's own argmax
tokens on every reference-produced trajectory the plain fast mode is only margin-bounded on (north star: "token indices
bit-exact under greedy/argmax decoding"; helpers/sampling.py:88-90, base.py:205-291).  Also: the report is sound at the
benchmark's batch size (every layout whose fast tokens differ from the exact mode's was marked), and the marked fraction
and the share of work the exact engine redoes are printed."""
import os

import numpy as np
import pytest
import torch

from oracle import restatement as R
from oracle import spec as SP
from oracle import synth

from test_hip_parity import WEIGHT_SEED, engine  # noqa: F401  (shared engine cache)
from test_hip_parity import cuda  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

_VG = {}


def verified(ds):
    from layout_dm_amd.verified import VerifiedGreedy

    if ds not in _VG:
        _VG[ds] = VerifiedGreedy(engine(ds, "fast"), engine(ds, "exact"))
        cal = _VG[ds].calibrate()
        print(f"[fast_verified/{ds}] calibration: {cal}")
    return _VG[ds]


def _cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "rico25_uncond_trajectory.npz"))
    yield "uncond", "rico25", g, "", None, None
    g = np.load(os.path.join(golden_dir, "publaynet_cond_c_trajectory.npz"))
    yield "cond=c", "publaynet", g, "", {"seq": g["cond_seq"].astype(np.int64), "mask": g["cond_mask"], "type": "c"}, None
    g = np.load(os.path.join(golden_dir, "rico25_refinement_trajectory.npz"))
    table = torch.from_numpy(g["weak_table"])
    seq_orig = torch.from_numpy(g["seq_orig"].astype(np.int64))
    yield "refinement", "rico25", g, "", {"seq": g["cond_seq"].astype(np.int64), "mask": g["cond_mask"],
                                          "type": "refinement",
                                          "weak_logits": table[seq_orig].permute(0, 2, 1).contiguous()}, None
    g = np.load(os.path.join(golden_dir, "rico25_cond_variants.npz"))
    for v in ("cwh", "partial", "td"):
        cond = None if v == "td" else {"seq": g[v + "_cond_seq"].astype(np.int64), "mask": g[v + "_cond_mask"], "type": v}
        yield v, "rico25", g, v + "_", cond, (0.15 if v == "td" else None)


def test_verified_greedy_steps_bit_exact_on_reference_trajectories(cuda, golden_dir):  # noqa: F811
    total = marked = 0
    for name, ds, g, prefix, cond, td in _cases(golden_dir):
        vg = verified(ds)
        T = vg.fast.T
        before = torch.from_numpy(g[prefix + "states_before"].astype(np.int32))
        ref_next = torch.from_numpy(g[prefix + "greedy_next"].astype(np.int32))
        bad = flagged = 0
        for i, tm in enumerate(g[prefix + "steps"]):
            tm = int(tm)
            tp = min(max(tm - int(T * td), 0), T - 1) if td else tm  # base.py:218-226
            out = vg.sample_step(before[i], tm, t_post=tp, cond=cond, step=i).cpu()
            bad += int((out != ref_next[i]).sum())
            flagged += vg.last_stats["marked_layout_steps"]
        n_ls = before.shape[0] * before.shape[1]
        print(f"[{name}/fast_verified] greedy tokens differing from the reference: {bad}/{ref_next.numel()}; "
              f"layout-steps re-decided in the exact mode: {flagged}/{n_ls}")
        assert bad == 0, (name, bad)
        total += n_ls
        marked += flagged
    print(f"[fast_verified] re-decided {marked}/{total} layout-steps = {marked / total:.3%}")
    assert marked < 0.25 * total  # the verification must stay the exception, or the mode has no point


def test_verified_greedy_loop_matches_reference(cuda, golden_dir):  # noqa: F811
    """Full T=100 greedy loop and the strided T=25 schedule == the reference's sample() (states after every step)."""
    spec = SP.RICO25
    vg = verified("rico25")
    for fixture, T_eval in (("rico25_uncond_greedy_loop.npz", 100), ("rico25_uncond_greedy_T25.npz", 25)):
        ref = torch.from_numpy(np.load(os.path.join(golden_dir, fixture))["states_after"].astype(np.int32))
        steps = R.timestep_list(spec.n_step, T_eval)
        tpost, prev = [], spec.n_step
        for t in steps:  # base.py:227-235
            skip = prev - t - 1
            tpost.append(t - skip if (skip > 0 and t > skip) else t)
            prev = t
        tok = torch.full((ref.shape[1], spec.seq_len), spec.mask_id, dtype=torch.int32, device=cuda)
        out, inter = vg.sample_loop(tok, steps, tpost, intermediates=True)
        print(f"[fast_verified loop T={T_eval}] {vg.last_stats}")
        assert torch.equal(inter.cpu(), ref)
        assert torch.equal(out.cpu(), ref[-1])


def test_tie_report_is_sound_at_benchmark_batch(cuda):  # noqa: F811
    """B = 512 on states a stochastic run visits (the greedy trajectory of random-init weights keeps everything [MASK]
    until the last two steps): every layout in which the fast mode's greedy tokens differ from the exact
    mode's must have been marked; the verified step equals the exact step on ALL 512 layouts."""
    spec = SP.RICO25
    B = 512
    from layout_dm_amd.verified import VerifiedGreedy

    fa, ex = engine("rico25", "fast", max_batch=B), engine("rico25", "exact", max_batch=B)
    vg = VerifiedGreedy(fa, ex)
    vg.calibrate()
    steps = R.timestep_list(spec.n_step, 100)
    tok = torch.full((B, spec.seq_len), spec.mask_id, dtype=torch.int32, device=cuda)
    _, inter = ex.sample_loop(tok, steps, steps, {"name": "random", "temperature": 1.0}, seed=3, intermediates=True)
    inter = inter.clone()
    n_marked = n_diff = 0
    for i in (9, 29, 49, 69, 89, 97, 98):
        before = inter[i - 1]
        t = steps[i]
        e_out = ex.sample_step(before, t, {"name": "deterministic"}, step=i)
        fa.set_tie_report(vg.tie_rel, vg.tie_abs)
        f_out = fa.sample_step(before, t, {"name": "deterministic"}, step=i)
        flags = fa.tie_flags(1, B)[0].bool()
        diff = (f_out != e_out).any(dim=1)
        assert not (diff & ~flags).any(), f"step {i}: a layout with differing tokens was not marked"
        v_out = vg.sample_step(before, t, step=i)
        assert torch.equal(v_out, e_out), i
        n_marked += int(flags.sum())
        n_diff += int(diff.sum())
    print(f"[tie report, B=512, 7 steps] marked layout-steps {n_marked}/{7 * B}; layouts whose fast tokens differ from "
          f"the exact mode's {n_diff} (all marked)")
    fa.set_tie_report(0.0, 0.0)
