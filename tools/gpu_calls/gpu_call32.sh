#!/bin/bash
# GPU call 32: vocabulary head fused into the stack kernel.
set -u
OUT=gpurun_out/r02_call32
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $OUT/pytest.txt
timeout 300 python bench.py --modes none --no-cpu-baseline --no-traffic --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_breakdown_ms'], d['roofline']['frac'], d['gemm_mfma_utilisation'])" | tee -a $OUT/bench.txt
timeout 300 python bench.py --config 3 --modes none --no-cpu-baseline --no-traffic --steps 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config3', d['value'], d['ms_per_step'])" | tee -a $OUT/bench.txt
LDM_ATTN_TM=1 timeout 200 python tools/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/phase.txt
