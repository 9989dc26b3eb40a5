"""Dev tool: one short fast-mode loop (for rocprofv3 --pmc passes over the production kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layout_dm_amd.diffusion import HipMaskAndReplaceDiffusion
from layout_dm_amd import synthetic as SP
spec = SP.RICO25
m = HipMaskAndReplaceDiffusion(n_category=25, precision="fast", max_batch=512, use_graph=False)
m.load_state_dict(SP.synth_state_dict(spec, seed=0))
out = m.sample(batch_size=512, sampling_cfg={"name": "random", "temperature": 1.0, "num_timesteps": 4}, seed=1)
torch.cuda.synchronize()
print(out.shape)
