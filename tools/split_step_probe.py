"""Dev probe (GPU box): per-kernel launch times of ONE split-mode sampling step (denoiser pass + posterior), sequential
(profiled eagerly: no lane overlap), and the whole 100-step loop.  PROBE_B layouts (default 512)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from layout_dm_amd import synthetic as SY  # noqa: E402
from layout_dm_amd.binding import Engine  # noqa: E402
from layout_dm_amd.diffusion import timestep_schedule  # noqa: E402

B = int(os.environ.get("PROBE_B", "512"))
prec = os.environ.get("PROBE_PREC", "split")
spec = SY.SPECS["rico25"]
e = Engine(n_category=spec.n_category, precision=prec, max_batch=B)
e.load_state_dict(SY.synth_state_dict(spec, seed=0))
tok = torch.full((B, spec.seq_len), spec.mask_id, dtype=torch.int32).cuda()
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
print("STEP " + json.dumps(res), flush=True)
tm, tp = timestep_schedule(100, 100)
for i in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out, _ = e.sample_loop(tok.clone(), tm, tp, cfg, seed=i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"LOOP {B} layouts x 100 steps: {dt * 1e3:.1f} ms = {B / dt:.0f} layouts/s  tokens sum {int(out.sum())}", flush=True)
