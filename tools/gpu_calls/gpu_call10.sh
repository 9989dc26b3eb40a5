#!/bin/bash
# GPU call 10: fused layer kernel (attention block + FFN in one per-layout launch) — parity first, then A/B and bench.
set -u
OUT=gpurun_out/r02_call10
mkdir -p $OUT
echo "== pytest (fast-mode)"
timeout 900 python -m pytest tests -m gpu -q -rA -k "fast or denoiser or full_batch" 2>&1 | tail -45 > $OUT/pytest.log; tail -8 $OUT/pytest.log
echo "== A/B" | tee $OUT/ab.txt
timeout 400 python tools/kernel_ab.py "LDM_FUSED_ATTN=2" "LDM_FUSED_ATTN=3" 2>&1 | tee -a $OUT/ab.txt
echo "== bench" | tee $OUT/bench.txt
for fa in 2 3; do
LDM_FUSED_ATTN=$fa timeout 300 python bench.py --modes none --no-cpu-baseline --no-traffic --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused_attn=$fa', d['value'], d['ms_per_step'], d['kernel_breakdown_ms'], d['roofline']['frac'], d['gemm_mfma_utilisation'])" | tee -a $OUT/bench.txt
done
LDM_FUSED_ATTN=3 timeout 300 python bench.py --lanes 1 --modes none --no-cpu-baseline --no-traffic --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused_attn=3 lanes=1', d['value'], d['ms_per_step'])" | tee -a $OUT/bench.txt
