#!/bin/bash
# GPU call 20: in-place AGPR parking of the attention outputs (no spills); full GPU test suite with the stream kernel as default.
set -u
OUT=gpurun_out/r02_call20
mkdir -p $OUT
timeout 300 python tools/kernel_ab.py "LDM_FUSED_ATTN=5 LDM_LAYER_DBG=1" "LDM_FUSED_ATTN=5" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
LDM_ATTN_TM=1 timeout 200 python tools/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/phase.txt
timeout 300 python bench.py --modes none --no-cpu-baseline --no-traffic --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_breakdown_ms'], d['roofline']['frac'], d['gemm_mfma_utilisation'])" | tee -a $OUT/bench.txt
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $OUT/pytest.txt
