#!/bin/bash
# Dev tool (GPU box): out_proj is the one GEMM left on gemm16x3_k — its tile configurations (LDM_X3_CFG) in the split bench and per launch, same box.
set -u
O=gpurun_out/r05_call32; mkdir -p $O
export TMPDIR=/tmp
Q="--precision split --steps 4 --warmup 1 --no-extras --no-cpu-baseline --no-traffic --modes none"
for cfg in 8 0 1 2 3 5 8; do
  LDM_DEV=1 LDM_X3_CFG=$cfg timeout 200 python bench.py $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('x3_cfg $cfg split', d['value'], 'layouts/s', 'gemm_attn_out', d['kernel_breakdown_ms'].get('gemm_attn_out'))" | tee -a $O/x3cfg.txt
done
