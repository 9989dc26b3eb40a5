#!/bin/bash
# Dev tool (GPU box): same-box A/B of two kernels_lngemm.hip builds (LDM_LG_DIRECT = 0 / 1: epilogue through an LDS transpose /
# straight from the accumulator layout): split logits error, per-launch times, split bench line.
set -u
O=gpurun_out/${1:-r05_call8}; mkdir -p $O
export TMPDIR=/tmp
cat > /tmp/err_probe.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from layout_dm_amd.binding import Engine
from oracle import restatement as R, spec as SP, synth
spec = SP.RICO25
sd = synth.synth_state_dict(spec, seed=1, perturb=True)
W = R.as_torch_weights(sd)
e = Engine(n_category=spec.n_category, precision="split", max_batch=16)
e.load_state_dict(sd)
g = torch.Generator().manual_seed(5)
worst = 0.0
for t in (95, 30):
    tokens = torch.empty(16, spec.seq_len, dtype=torch.long)
    for a in range(spec.n_attr):
        ids = torch.as_tensor(spec.full_ids(a))
        tokens[:, a::spec.n_attr] = ids[torch.randint(0, len(ids) - 1, (16, spec.max_elem), generator=g)]
    tokens[torch.rand(16, spec.seq_len, generator=g) < t / 99] = spec.mask_id
    ref = R.denoiser_logits(W, spec, tokens, t)
    lg = e.denoise_logits(tokens.int(), t).cpu()
    assert bool(torch.isfinite(lg).all()), "non-finite logits"
    worst = max(worst, ((lg - ref).abs().max() / ref.abs().max()).item())
print(os.environ.get("LDM_HIP_LIB", "default").split("/")[-1], f"split logits rel err {worst:.3e}")
PY
Q="--precision split --steps 5 --warmup 1 --no-extras --no-cpu-baseline --no-traffic --modes none"
for v in direct0 direct1 direct0 direct1; do
  export LDM_HIP_LIB=$(pwd)/layout_dm_amd/build/variants/libldm_$v.so
  timeout 200 python /tmp/err_probe.py 2>&1 | tail -1 | tee -a $O/lngemm_epilogue_ab.txt
  LDM_DEV=1 timeout 120 python tools/lngemm_probe.py 10 2>/dev/null | tail -1 | sed "s/^/$v /" | tee -a $O/lngemm_epilogue_ab.txt
  timeout 300 python bench.py $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bench split', d['value'], 'layouts/s')" | tee -a $O/lngemm_epilogue_ab.txt
done
