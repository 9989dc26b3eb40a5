"""Dev tool (GPU box): per-kernel times of one split-mode denoiser pass over 256 layouts (the chunk of the per-step path), eager,
HIP events around every launch — used with LDM_DEV=1 LDM_LNGEMM_ABL=<mask> to see what bounds kernels_lngemm.hip.
    python tools/lngemm_probe.py [reps]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layout_dm_amd import synthetic as SP
from layout_dm_amd.binding import Engine

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
spec = SP.SPECS["rico25"]
B = 256
e = Engine(n_category=spec.n_category, precision="split", max_batch=B)
e.load_state_dict(SP.synth_state_dict(spec, seed=0))
tok = torch.randint(0, spec.n_class - 2, (B, spec.seq_len), dtype=torch.int32, device="cuda")
for _ in range(2):
    e.denoise_logits(tok, 50)
torch.cuda.synchronize()
e.set_profiling(True)
for _ in range(reps):
    e.denoise_logits(tok, 50)
torch.cuda.synchronize()
rows = e.profile(reset=True)
if os.environ.get("LDM_LNGEMM_TM") == "1":
    import ctypes as C
    ph = (C.c_ulonglong * 8)()
    e.lib.ldm_dev_lngemm_phases.argtypes = [C.POINTER(C.c_ulonglong)]
    e.lib.ldm_dev_lngemm_phases.restype = None
    e.lib.ldm_dev_lngemm_phases(ph)
    n = max(ph[0], 1)
    print(json.dumps({"lngemm_phase_cycles_per_workgroup_mean_over_all_launches": {
        "workgroups": ph[0], "total": ph[1] // n, "prologue": ph[2] // n, "sync_waits(vmcnt+barrier)": ph[3] // n,
        "lgkm_waits": ph[4] // n, "tail_epilogue": ph[5] // n}, "note": "s_memtime ticks (100 MHz constant clock x ?): compare ratios"}))
print(json.dumps({"abl": os.environ.get("LDM_LNGEMM_ABL", "0"), "knobs": e.describe().get("knobs"),
                  "us_per_launch": {r["name"]: round(1e3 * r["ms"] / max(r["launches"], 1), 1) for r in rows}}))
