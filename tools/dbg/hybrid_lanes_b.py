"""Dev probe (GPU): hybrid loop throughput at several batch sizes for the (chunk, lanes) setting in the environment (LDM_DEV=1 LDM_CHUNK / LDM_LANES)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from layout_dm_amd import synthetic as SY
from layout_dm_amd.binding import Engine
from layout_dm_amd.diffusion import timestep_schedule
sy = SY.SPECS["rico25"]
tm, tp = timestep_schedule(100, 100)
PREC = os.environ.get("PROBE_PREC", "hybrid")
for B in [int(x) for x in os.environ.get("PROBE_BS", "256,300,512,1024").split(",")]:
    e = Engine(n_category=sy.n_category, precision=PREC, max_batch=B)
    e.load_state_dict(SY.synth_state_dict(sy, seed=0))
    tok = torch.full((B, sy.seq_len), sy.mask_id, dtype=torch.int32).cuda()
    cfg = {"name": "random", "temperature": 1.0}
    best = 0
    for i in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e.sample_loop(tok.clone(), tm, tp, cfg, seed=i)
        torch.cuda.synchronize(); best = max(best, B / (time.perf_counter() - t0))
    print(f"[{PREC} {os.environ.get('LDM_CHUNK', 'default')} x {os.environ.get('LDM_LANES', 'default')}] B = {B}: {best:.0f} layouts/s", flush=True)
    e.close()
