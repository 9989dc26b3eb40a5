#!/bin/bash
# GPU call 39: HEAD re-verification after the schedule refactor / LEAN wait fix: full GPU suite, smoke, bench.
set -u
OUT=gpurun_out/r02_call39
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $OUT/pytest.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $OUT/smoke.txt
timeout 300 python bench.py --modes none --no-cpu-baseline --no-traffic --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_breakdown_ms'], d['roofline']['frac'], d['gemm_mfma_utilisation'])" | tee -a $OUT/bench.txt
