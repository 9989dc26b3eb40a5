#!/bin/bash
# Dev tool (GPU box): lngemm level 2 (out_proj / linear2 as the GEMM prologue of the row-resident kernels) — parity subset, per-launch
# times at levels 2 / 1 / 0 (LDM_X3_LNGEMM), split bench at the three levels on the same box.
set -u
O=gpurun_out/${1:-r05_call24}; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "repeatable or (split and not b1024) or precision_report" > $O/pytest_split.log 2>&1; tail -6 $O/pytest_split.log
for lv in 2 1 0; do
  LDM_DEV=1 LDM_X3_LNGEMM=$lv timeout 120 python tools/lngemm_probe.py 10 2>/dev/null | tail -1 | sed "s/^/level=$lv /" | tee -a $O/lngemm_probe.txt
done
Q="--precision split --steps 5 --warmup 1 --no-extras --no-cpu-baseline --no-traffic --modes none"
for lv in 2 1 0 2; do
  LDM_DEV=1 LDM_X3_LNGEMM=$lv timeout 300 python bench.py $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split level=$lv', d['value'], 'layouts/s', json.dumps(d.get('kernel_breakdown_ms')))" | tee -a $O/lngemm_probe.txt
done
