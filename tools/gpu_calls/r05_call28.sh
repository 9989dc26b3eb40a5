#!/bin/bash
# Dev tool (GPU box): the split mode with linear2 as the GEMM prologue by default (level 4): parity subset, alternative structures, bench A/B
# against level 1 (alternating), probe, SQ counters.
set -u
O=gpurun_out/${1:-r05_call28}; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "repeatable or (split and not b1024) or precision_report or alternative_structures or default_path or getcond" > $O/pytest_split.log 2>&1; tail -4 $O/pytest_split.log
LDM_DEV=1 timeout 120 python tools/lngemm_probe.py 10 2>/dev/null | tail -1 | tee -a $O/probe.txt
Q="--precision split --steps 5 --warmup 1 --no-extras --no-cpu-baseline --no-traffic --modes none"
for i in 1 2; do
  timeout 300 python bench.py $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default (level 4)', d['value'], 'layouts/s', d['config']['library'].get('kernels'), json.dumps(d.get('kernel_breakdown_ms')))" | tee -a $O/probe.txt
  LDM_DEV=1 LDM_X3_LNGEMM=1 timeout 300 python bench.py $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('level 1', d['value'], 'layouts/s')" | tee -a $O/probe.txt
done
bash tools/pmc_sq.sh $O/sq_counters_split.txt split 4 > /dev/null 2>&1; grep -E "^void ldm::lngemm|matrix pipes|effective clock|mean duration" $O/sq_counters_split.txt | head -16
