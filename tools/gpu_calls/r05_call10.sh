#!/bin/bash
# Dev tool (GPU box): split mode at HEAD — repeatability + parity tests, lanes sweep, SQ counters.
set -u
O=gpurun_out/${1:-r05_call10}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "repeatable or (split and not b1024) or precision_report" > $O/pytest_split.log 2>&1; tail -4 $O/pytest_split.log
Q="--precision split --steps 5 --warmup 1 --no-extras --no-cpu-baseline --no-traffic --modes none"
for lanes in 2 1 3; do
  timeout 300 python bench.py $Q --lanes $lanes 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes $lanes split', d['value'], 'layouts/s', json.dumps(d.get('kernel_breakdown_ms')))" | tee -a $O/split_lanes.txt
done
for ck in 128 512; do
  timeout 300 python bench.py $Q --chunk $ck 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk $ck split', d['value'], 'layouts/s', d['config']['lanes'])" | tee -a $O/split_lanes.txt
done
bash tools/pmc_sq.sh $O/sq_counters_split.txt split 4 > /dev/null 2>&1; head -60 $O/sq_counters_split.txt | grep -E "^void|matrix pipes|MFMA busy|effective clock|mean duration"
