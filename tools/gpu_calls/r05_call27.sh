#!/bin/bash
# Dev tool (GPU box): the alternative split structures under test + levels 3 / 4 (only out_proj / only linear2 as a GEMM prologue) in the bench.
set -u
O=gpurun_out/${1:-r05_call27}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "alternative_structures or describe or knobs" -s > $O/pytest_alt.log 2>&1; grep -E "split structure|passed|failed" $O/pytest_alt.log | tail -12
Q="--precision split --steps 4 --warmup 1 --no-extras --no-cpu-baseline --no-traffic --modes none"
for lv in 1 4 3 2 1; do
  for cfg in "2 256" "4 128"; do
    set -- $cfg
    LDM_DEV=1 LDM_X3_LNGEMM=$lv timeout 300 python bench.py $Q --lanes $1 --chunk $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('level $lv lanes $1 chunk $2 split', d['value'], 'layouts/s')" | tee -a $O/sweep.txt
  done
done
