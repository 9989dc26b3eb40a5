"""Dev probe (GPU box): cond=relation vs cond=c on the one-launch loop, 512 layouts, T = 100, same engine, interleaved."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from layout_dm_amd import synthetic as SP
from layout_dm_amd.binding import Engine
from layout_dm_amd.diffusion import timestep_schedule
spec = SP.SPECS["rico25"]; B = 512
e = Engine(n_category=spec.n_category, precision="fast", max_batch=B)
e.load_state_dict(SP.synth_state_dict(spec, seed=0))
cond, graph = SP.synth_cond_relation(spec, B, seed=0)
plan = e.make_relation(graph, SP.linear_bin_centres(spec.n_bin), [16, 16, 31, 31], 3e6, 3, B)
tm, tp = timestep_schedule(100, 100)
c = {"seq": cond["seq"], "mask": cond["mask"], "type": "relation"}
cc = {"seq": cond["seq"], "mask": cond["mask"], "type": "c"}
def run(rel):
    ts = []
    for i in range(6):
        tok = torch.from_numpy(cond["seq"]).int().cuda()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e.sample_loop(tok, tm, tp, {"name": "random", "temperature": 1.0}, cond=c if rel else cc, seed=i, relation=plan if rel else None)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sum(ts[1:]) / 5
for k in range(3):
    a, b = run(False), run(True)
    print(f"cond=c {a*1e3:.2f} ms  cond=relation {b*1e3:.2f} ms  ratio {a/b:.4f}  ({B/a:.0f} vs {B/b:.0f} layouts/s)", flush=True)
