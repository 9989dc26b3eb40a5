#!/bin/bash
# GPU call 6: two accumulator chains (FFN CH=2, attention V=7) — A/B, phase probe, parity tests, bench.
set -u
OUT=gpurun_out/r02_call6
mkdir -p $OUT
echo "== A/B (256-layout launches)" | tee $OUT/ab.txt
timeout 500 python tools/kernel_ab.py "LDM_FFN_CH=1 LDM_ATTN_V=3" "LDM_FFN_CH=2 LDM_ATTN_V=3" "LDM_FFN_CH=1 LDM_ATTN_V=7" "LDM_FFN_CH=2 LDM_ATTN_V=7" 2>&1 | tee -a $OUT/ab.txt
echo "== phase probe" | tee $OUT/phase.txt
LDM_FFN_DBG=3 LDM_ATTN_TM=1 timeout 200 python tools/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/phase.txt
echo "== pytest (fast-mode relevant + fid)"
timeout 900 python -m pytest tests -m gpu -q -rA -k "fast or fid or denoiser or full_batch or entry" 2>&1 | tail -40 > $OUT/pytest.log; tail -6 $OUT/pytest.log
echo "== bench"
timeout 300 python bench.py --modes none --no-cpu-baseline --no-traffic --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_breakdown_ms'], d['roofline']['frac'], d['gemm_mfma_utilisation'])" | tee $OUT/bench.txt
