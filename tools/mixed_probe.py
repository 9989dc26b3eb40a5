"""Dev probe (GPU box): the two-product form of the split mode's weight GEMMs (PROBE_PREC=mixed: activations hi + lo, weights fp16
only) against the three-product form (PROBE_PREC=split) — logits error against the float64 oracle on the init / mid / wide points and
against the reference's own logits on the fitted checkpoint, the per-kernel launch times and the 100-step loop.  (Its first run,
profiles/r06_mixed_mode.txt part 1, selected the form with a dev knob of the split handle before the mode had a name.)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from layout_dm_amd import synthetic as SY  # noqa: E402
from layout_dm_amd.binding import Engine  # noqa: E402
from layout_dm_amd.diffusion import timestep_schedule  # noqa: E402
from oracle import restatement as R, spec as SP, synth  # noqa: E402

PREC = os.environ.get("PROBE_PREC", "mixed")   # split | mixed | hybrid
tag = PREC
spec = SP.RICO25
g = torch.Generator().manual_seed(0)


def rel(a, ref):
    return ((a.double() - ref.double()).abs().max() / ref.abs().max()).item()


fit = os.path.join(ROOT, "oracle", "_fit", "rico25_fitted.npz")
if os.path.exists(fit):
    w = np.load(fit)
    sd = {k: w[k] for k in w.files}
    gold = np.load(os.path.join(ROOT, "tests", "golden", "rico25_fitted.npz"))
    e = Engine(n_category=spec.n_category, precision=PREC, max_batch=8)
    e.load_state_dict(sd)
    errs = [rel(e.denoise_logits(torch.from_numpy(gold[f"tokens_{int(t)}"].astype(np.int32)), int(t)).cpu(), torch.from_numpy(gold[f"logits_{int(t)}"]))
            for t in gold["ts"]]
    bad = 0
    for i, t in enumerate(gold["steps"]):
        out = e.sample_step(torch.from_numpy(gold["states_before"][i].astype(np.int32)), int(t), {"name": "deterministic"}, step=i).cpu()
        bad += int((out.numpy() != gold["greedy_next"][i]).sum())
    print(f"[{tag}] fitted: logits error vs the reference " + " ".join(f"{x:.2e}" for x in errs) + f"; greedy tokens differing {bad}/25000", flush=True)
    e.close()
for point in ("init", "mid", "wide"):
    sd = synth.synth_state_dict(spec, seed=0) if point == "init" else synth.trained_like_state_dict(spec, point, seed=3)
    W64 = R.as_torch_weights(sd, torch.float64)
    e = Engine(n_category=spec.n_category, precision=PREC, max_batch=8)
    e.load_state_dict(sd)
    errs = []
    for t in (50, 90, 5):
        tokens = torch.empty(4, spec.seq_len, dtype=torch.long)
        for a in range(spec.n_attr):
            ids = torch.as_tensor(spec.full_ids(a))
            tokens[:, a::spec.n_attr] = ids[torch.randint(0, len(ids) - 1, (4, spec.max_elem), generator=g)]
        tokens[torch.rand(4, spec.seq_len, generator=g) < t / 99] = spec.mask_id
        errs.append(rel(e.denoise_logits(tokens.int(), t).cpu()[..., :spec.n_class], R.denoiser_logits(W64, spec, tokens, t, dtype=torch.float64)))
    print(f"[{tag}] {point}: logits error vs the float64 oracle " + " ".join(f"{x:.2e}" for x in errs), flush=True)
    e.close()

B = 512
sy = SY.SPECS["rico25"]
e = Engine(n_category=sy.n_category, precision=PREC, max_batch=B)
e.load_state_dict(SY.synth_state_dict(sy, seed=0))
tok = torch.full((B, sy.seq_len), sy.mask_id, dtype=torch.int32).cuda()
cfg = {"name": "random", "temperature": 1.0}
for _ in range(2):
    e.sample_step(tok, 50, cfg, seed=1)
torch.cuda.synchronize()
e.set_profiling(True)
for i in range(5):
    e.sample_step(tok, 50, cfg, seed=i)
torch.cuda.synchronize()
rows = e.profile(reset=True)
e.set_profiling(False)
res = {r["name"]: round(1e3 * r["ms"] / r["launches"], 2) for r in rows}
res["_sum_us_per_step"] = round(sum(1e3 * r["ms"] for r in rows) / 5, 1)
print(f"[{tag}] STEP " + json.dumps(res), flush=True)
tm, tp = timestep_schedule(100, 100)
for i in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out, _ = e.sample_loop(tok.clone(), tm, tp, cfg, seed=i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"[{tag}] LOOP {B} layouts x 100 steps: {dt * 1e3:.1f} ms = {B / dt:.0f} layouts/s  tokens sum {int(out.sum())}", flush=True)
