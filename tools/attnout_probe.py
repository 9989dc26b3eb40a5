"""Dev probe (GPU box): the split mode's fused attention + out_proj launch (kernels_attnout.hip) against the r05 structure
(attn16x3_k + gemm16x3_k, LDM_DEV=1 LDM_X3_ATTNOUT=0) and against the fp32-MFMA engine: logits error on init / mid / wide weights at a
few batch shapes, run-to-run bitwise repeatability, per-kernel launch times of one denoiser pass.

    python tools/attnout_probe.py            # numerics + timing of the default build
    LDM_DEV=1 LDM_X3_ATTNOUT=0 python tools/attnout_probe.py timing     # the r05 structure
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from layout_dm_amd import synthetic as SY  # noqa: E402
from layout_dm_amd.binding import Engine  # noqa: E402


def tokens_for(spec, B, seed, mask_frac=0.5):
    g = torch.Generator().manual_seed(seed)
    t = torch.empty(B, spec.seq_len, dtype=torch.long)
    for a in range(spec.n_attr):
        ids = torch.as_tensor(spec.full_ids(a))
        t[:, a::spec.n_attr] = ids[torch.randint(0, len(ids) - 1, (B, spec.max_elem), generator=g)]
    t[torch.rand(B, spec.seq_len, generator=g) < mask_frac] = spec.mask_id
    return t.int()


def vs_float64():
    """B = 3: both engines against the float64 oracle restatement (what the reference computes, without fp32 noise)."""
    from oracle import restatement as R
    from oracle import spec as SP
    from oracle import synth

    spec = SP.SPECS["rico25"]
    for point in ("init", "mid", "wide"):
        sd = synth.trained_like_state_dict(spec, point, seed=2)
        W = R.as_torch_weights(sd, dtype=torch.float64)
        tok = tokens_for(spec, 3, 8)
        for t in (90, 40, 3):
            ref = R.denoiser_logits(W, spec, tok.long(), t, dtype=torch.float64)
            res = {}
            for prec in ("exact", "split"):
                e = Engine(n_category=spec.n_category, precision=prec, max_batch=3)
                e.load_state_dict(sd)
                got = e.denoise_logits(tok, t).double().cpu()
                res[prec] = ((got - ref).abs().max() / ref.abs().max()).item()
            print(f"F64 {point} t={t}: exact {res['exact']:.2e}  split {res['split']:.2e}", flush=True)


def numerics():
    spec = SY.SPECS["rico25"]
    out = {}
    for point in ("init", "mid", "wide"):
        sd = SY.trained_like_state_dict(spec, point, seed=2) if hasattr(SY, "trained_like_state_dict") else SY.synth_state_dict(spec, seed=0)
        for B in (3, 256, 300):
            ex = Engine(n_category=spec.n_category, precision="exact", max_batch=B)
            ex.load_state_dict(sd)
            sp = Engine(n_category=spec.n_category, precision="split", max_batch=B)
            sp.load_state_dict(sd)
            tok = tokens_for(spec, B, 5 + B)
            for t in (90, 40, 3):
                a = ex.denoise_logits(tok, t).float().cpu()
                b = sp.denoise_logits(tok, t).float().cpu()
                b2 = sp.denoise_logits(tok, t).float().cpu()
                rel = ((a - b).abs().max() / a.abs().max()).item()
                out[f"{point}_B{B}_t{t}"] = {"rel_err_vs_fp32_mfma": rel, "bitwise_repeatable": bool(torch.equal(b, b2)),
                                           "finite": bool(torch.isfinite(b).all())}
                print(point, B, t, out[f"{point}_B{B}_t{t}"], flush=True)
            del ex, sp
        if not hasattr(SY, "trained_like_state_dict"):
            break
    return out


def timing(B=512, reps=10):
    spec = SY.SPECS["rico25"]
    e = Engine(n_category=spec.n_category, precision="split", max_batch=B)
    e.load_state_dict(SY.synth_state_dict(spec, seed=0))
    tok = tokens_for(spec, B, 0)
    for _ in range(2):
        o = e.denoise_logits(tok, 50)
    torch.cuda.synchronize()
    e.set_profiling(True)
    for _ in range(reps):
        o = e.denoise_logits(tok, 50)
    torch.cuda.synchronize()
    rows = e.profile(reset=True)
    e.set_profiling(False)
    res = {r["name"]: round(1e3 * r["ms"] / r["launches"], 2) for r in rows}
    res["_sum_us_per_pass"] = round(sum(1e3 * r["ms"] for r in rows) / reps, 1)
    res["_logits_sum"] = float(o.double().sum())
    res["_describe"] = e.describe() if hasattr(e, "describe") else ""
    print("TIMING " + json.dumps(res), flush=True)
    # the whole loop, as bench.py's modes.split measures it
    import time

    from layout_dm_amd.diffusion import timestep_schedule

    tm, tp = timestep_schedule(100, 100)
    start = torch.full((B, spec.seq_len), spec.mask_id, dtype=torch.int32).cuda()
    for i in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.sample_loop(start.clone(), tm, tp, {"name": "random", "temperature": 1.0}, seed=i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"LOOP {B} layouts x 100 steps: {dt * 1e3:.1f} ms = {B / dt:.0f} layouts/s", flush=True)
    return res


def phases(B=256, reps=5):
    """LDM_DEV=1 LDM_ATTNOUT_TM=1: s_memtime sums of the fused attention + out_proj kernel, cycles per workgroup and launch."""
    import ctypes

    from layout_dm_amd import binding

    lib = binding.load_library()
    spec = SY.SPECS["rico25"]
    e = Engine(n_category=spec.n_category, precision="split", max_batch=B)
    e.load_state_dict(SY.synth_state_dict(spec, seed=0))
    tok = tokens_for(spec, B, 0)
    e.denoise_logits(tok, 50)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 24)()
    lib.ldm_dev_attnout_phases(buf)
    for _ in range(reps):
        e.denoise_logits(tok, 50)
    torch.cuda.synchronize()
    lib.ldm_dev_attnout_phases(buf)
    n = max(1, buf[15])
    names = ["Ba", "Bb", "Bc", "Bd1", "Bd2", "Bd3"]
    per = lambda v: v / n   # noqa: E731
    print(f"PHASES workgroups {buf[15]} (x 100 MHz s_memtime ticks per workgroup)")
    print("  vmcnt waits  :", "  ".join(f"{nm} {per(buf[i]):8.0f}" for i, nm in enumerate(names)))
    print("  barrier waits:", "  ".join(f"{nm} {per(buf[6 + i]):8.0f}" for i, nm in enumerate(names)))
    print(f"  head loop {per(buf[12]):.0f}   epilogue {per(buf[13]):.0f}   kernel {per(buf[14]):.0f}")
    print("  in the loop  :", "  ".join(f"{nm} {per(buf[16 + i]):8.0f}" for i, nm in enumerate(["softmax", "PV", "O-split", "out_proj", "scores"])), flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["numerics", "timing"]
    if "f64" in what:
        vs_float64()
    if "numerics" in what:
        numerics()
    if "timing" in what:
        timing()
    if "phases" in what:
        phases()
