"""Dev tool / bench.py helper: one short loop of the sampling hot path at the bench's launch shapes (B=512 => M=64000-row
kernels), to be run under `rocprofv3 --pmc <counter>` (one counter pass per run).  PMC_DATASET / PMC_PRECISION select
the model shape and numerics mode (default rico25 / fast)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layout_dm_amd.diffusion import HipMaskAndReplaceDiffusion
from layout_dm_amd import synthetic as SP
spec = SP.SPECS[os.environ.get("PMC_DATASET", "rico25")]
m = HipMaskAndReplaceDiffusion(n_category=spec.n_category, precision=os.environ.get("PMC_PRECISION", "fast"),
                               max_batch=512, use_graph=False)
m.load_state_dict(SP.synth_state_dict(spec, seed=0))
out = m.sample(batch_size=512, sampling_cfg={"name": "random", "temperature": 1.0, "num_timesteps": 4}, seed=1)
torch.cuda.synchronize()
print(out.shape)
