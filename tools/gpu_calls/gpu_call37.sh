#!/bin/bash
# GPU call 37: exact mode, 128x160 fp32 GEMM tile at 3 workgroups per CU without scratch vs 128x128; exact-mode tests.
set -u
OUT=gpurun_out/r02_call37
mkdir -p $OUT
for w in 0 1; do
LDM_GEMM32_WIDE=$w timeout 300 python bench.py --precision exact --modes none --no-cpu-baseline --no-traffic --steps 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wide=$w', d['value'], d['ms_per_step'], d['kernel_breakdown_ms'])" | tee -a $OUT/exact_ab.txt
done
timeout 600 python -m pytest tests -m gpu -q -x -k "exact or golden or greedy or teacher" 2>&1 | tail -3 | tee $OUT/pytest.txt
