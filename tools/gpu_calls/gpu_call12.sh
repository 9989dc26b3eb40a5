#!/bin/bash
# GPU call 12: multi-layer fused kernel (all 4 layers per launch) — parity, A/B vs one launch per layer, bench.
set -u
OUT=gpurun_out/r02_call12
mkdir -p $OUT
echo "== pytest (fast-mode)"
timeout 900 python -m pytest tests -m gpu -q -rA -k "fast or denoiser or full_batch" 2>&1 | tail -45 > $OUT/pytest.log; tail -8 $OUT/pytest.log
echo "== A/B" | tee $OUT/ab.txt
timeout 400 python tools/kernel_ab.py "LDM_FUSED_ATTN=3" "LDM_FUSED_ATTN=4" 2>&1 | tee -a $OUT/ab.txt
echo "== bench" | tee $OUT/bench.txt
for fa in 3 4; do
LDM_FUSED_ATTN=$fa timeout 300 python bench.py --modes none --no-cpu-baseline --no-traffic --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused_attn=$fa', d['value'], d['ms_per_step'], d['kernel_breakdown_ms'], d['roofline']['frac'], d['gemm_mfma_utilisation'])" | tee -a $OUT/bench.txt
done
echo "== phase probe (multi-layer)" | tee $OUT/phase.txt
LDM_ATTN_TM=1 timeout 200 python tools/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/phase.txt
