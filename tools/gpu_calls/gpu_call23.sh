#!/bin/bash
# GPU call 23: stack kernel (LDM_FUSED_ATTN=6: all layers per launch, rows resident in the out-projection accumulators).
set -u
OUT=gpurun_out/r02_call23
mkdir -p $OUT
timeout 300 python tools/kernel_ab.py "LDM_FUSED_ATTN=5" "LDM_FUSED_ATTN=6" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
LDM_ATTN_TM=1 timeout 200 python tools/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/phase.txt
LDM_FUSED_ATTN=6 timeout 600 python -m pytest tests -m gpu -q -x -k "denoiser or fast_mode or full_batch_512_one" 2>&1 | tail -3 | tee $OUT/pytest.txt
for v in 5 6; do
LDM_FUSED_ATTN=$v timeout 300 python bench.py --modes none --no-cpu-baseline --no-traffic --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused=$v', d['value'], d['ms_per_step'], d['kernel_breakdown_ms'], d['roofline']['frac'], d['gemm_mfma_utilisation'])" | tee -a $OUT/bench.txt
done
