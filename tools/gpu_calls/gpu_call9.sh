#!/bin/bash
# GPU call 9: FFN continuous pipeline (FfnStream) — A/B, phase probe, parity tests.
set -u
OUT=gpurun_out/r02_call9
mkdir -p $OUT
echo "== A/B" | tee $OUT/ab.txt
timeout 400 python tools/kernel_ab.py "LDM_FFN_STREAM=0" "LDM_FFN_STREAM=1" 2>&1 | tee -a $OUT/ab.txt
echo "== phase probe (stream)" | tee $OUT/phase.txt
LDM_FFN_DBG=3 timeout 150 python tools/phase_probe.py 2>&1 | grep -v amdgpu.ids | grep -A9 "^ffn" | tee -a $OUT/phase.txt
echo "== pytest"
timeout 900 python -m pytest tests -m gpu -q -rA -k "fast or fid or denoiser or full_batch" 2>&1 | tail -40 > $OUT/pytest.log; tail -5 $OUT/pytest.log
echo "== bench"
for st in 0 1; do
LDM_FFN_STREAM=$st timeout 300 python bench.py --modes none --no-cpu-baseline --no-traffic --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stream=$st', d['value'], d['ms_per_step'], d['kernel_breakdown_ms'], d['roofline']['frac'], d['gemm_mfma_utilisation'])" | tee -a $OUT/bench.txt
done
