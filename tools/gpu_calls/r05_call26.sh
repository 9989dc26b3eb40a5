#!/bin/bash
# Dev tool (GPU box): (lanes, chunk) sweep of the split mode at lngemm level 2 (GEMM prologues) and level 1, same box.
set -u
O=gpurun_out/${1:-r05_call26}; mkdir -p $O
export TMPDIR=/tmp
Q="--precision split --steps 4 --warmup 1 --no-extras --no-cpu-baseline --no-traffic --modes none"
for lv in 2 1; do
  for cfg in "2 256" "1 256" "2 128" "3 128" "4 128" "1 512"; do
    set -- $cfg
    LDM_DEV=1 LDM_X3_LNGEMM=$lv timeout 300 python bench.py $Q --lanes $1 --chunk $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('level $lv lanes $1 chunk $2 split', d['value'], 'layouts/s')" | tee -a $O/sweep.txt
  done
done
