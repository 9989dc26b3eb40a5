"""Dev tool / bench.py helper: one sampling call at the bench's launch shape (config 2: 512 layouts, T = PMC_STEPS = 100
reverse steps — in the fast mode ONE launch of the loop-resident stack kernel), to be run under `rocprofv3 --pmc <counter>`
(one counter pass per run).  PMC_DATASET / PMC_PRECISION select the model shape and numerics mode (default rico25 / fast)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layout_dm_amd.diffusion import HipMaskAndReplaceDiffusion
from layout_dm_amd import synthetic as SP
spec = SP.SPECS[os.environ.get("PMC_DATASET", "rico25")]
m = HipMaskAndReplaceDiffusion(n_category=spec.n_category, precision=os.environ.get("PMC_PRECISION", "fast"),
                               max_batch=512, use_graph=False)
m.load_state_dict(SP.synth_state_dict(spec, seed=0))
out = m.sample(batch_size=512, sampling_cfg={"name": "random", "temperature": 1.0, "num_timesteps": int(os.environ.get("PMC_STEPS", "100"))}, seed=1)
torch.cuda.synchronize()
print(out.shape)
