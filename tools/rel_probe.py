"""Dev probe (GPU box): three cond=relation sampling calls (rico25, 512 layouts, T = 100, random) for `rocprofv3 --stats`
(tools/gpu_call.sh relstats): ONE launch of stack_stream_k<., 2, true> per call is what the table must show."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from layout_dm_amd import synthetic as SP  # noqa: E402
from layout_dm_amd.binding import Engine  # noqa: E402
from layout_dm_amd.diffusion import timestep_schedule  # noqa: E402

spec = SP.SPECS["rico25"]
B = 512
e = Engine(n_category=spec.n_category, precision="fast", max_batch=B)
e.load_state_dict(SP.synth_state_dict(spec, seed=0))
cond, graph = SP.synth_cond_relation(spec, B, seed=0)
plan = e.make_relation(graph, SP.linear_bin_centres(spec.n_bin), [16, 16, 31, 31], 3e6, 3, B)
tm, tp = timestep_schedule(100, 100)
c = {"seq": cond["seq"], "mask": cond["mask"], "type": "relation"}
for i in range(3):
    tok = torch.from_numpy(cond["seq"]).int().cuda()
    e.sample_loop(tok, tm, tp, {"name": "random", "temperature": 1.0}, cond=c, seed=i, relation=plan)
torch.cuda.synchronize()
