#!/bin/bash
# GPU call 19: stream layer kernel with the 3-stage ring + hand-pipelined attention core (P cast one step ahead of its MFMA).
set -u
OUT=gpurun_out/r02_call19
mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -x -k "denoiser or fast or full_batch_512_one" 2>&1 | tail -4 | tee $OUT/pytest.txt
timeout 300 python tools/kernel_ab.py "LDM_FUSED_ATTN=3" "LDM_FUSED_ATTN=5 LDM_LAYER_DBG=1" "LDM_FUSED_ATTN=5" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
LDM_ATTN_TM=1 timeout 200 python tools/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/phase.txt
for v in "LDM_FUSED_ATTN=3" "LDM_LAYER_DBG=1" "LDM_LAYER_DBG=0"; do
env $v timeout 300 python bench.py --modes none --no-cpu-baseline --no-traffic --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['kernel_breakdown_ms'], d['roofline']['frac'], d['gemm_mfma_utilisation'])" | tee -a $OUT/bench.txt
done
