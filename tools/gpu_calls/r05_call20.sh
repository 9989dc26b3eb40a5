#!/bin/bash
# Dev tool (GPU box): phase offset between the lngemm workgroups (LDM_LNGEMM_PHASE, 64-cycle units per phase step): per-launch times + split bench.
set -u
O=gpurun_out/${1:-r05_call20}; mkdir -p $O
export TMPDIR=/tmp
for ph in ${PHASES:-0 6 12 20 32 0}; do
  LDM_DEV=1 LDM_LNGEMM_PHASE=$ph timeout 120 python tools/lngemm_probe.py 10 2>/dev/null | tail -1 | sed "s/^/phase=$ph /" | tee -a $O/lngemm_phase.txt
done
Q="--precision split --steps 5 --warmup 1 --no-extras --no-cpu-baseline --no-traffic --modes none"
for ph in ${BENCH_PHASES:-0 12}; do
LDM_DEV=1 LDM_LNGEMM_PHASE=$ph timeout 300 python bench.py $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split phase=$ph', d['value'], 'layouts/s', json.dumps(d.get('kernel_breakdown_ms')))" | tee -a $O/lngemm_phase.txt
done
timeout 600 python -m pytest tests -m gpu -q -x -k "repeatable or (split and not b1024) or precision_report" > $O/pytest_split.log 2>&1; tail -2 $O/pytest_split.log
