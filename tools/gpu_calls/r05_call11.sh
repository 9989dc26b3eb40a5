#!/bin/bash
# Dev tool (GPU box): kernels_lngemm.hip round — parity / repeatability, per-launch times next to the r04 structure, split bench.
set -u
O=gpurun_out/${1:-r05_call11}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "repeatable or (split and not b1024) or precision_report" > $O/pytest_split.log 2>&1; tail -4 $O/pytest_split.log
LDM_DEV=1 timeout 120 python tools/lngemm_probe.py 10 2>/dev/null | tail -1 | tee -a $O/lngemm_probe.txt
LDM_DEV=1 LDM_X3_LNGEMM=0 timeout 120 python tools/lngemm_probe.py 10 2>/dev/null | tail -1 | tee -a $O/lngemm_probe.txt
Q="--precision split --steps 5 --warmup 1 --no-extras --no-cpu-baseline --no-traffic --modes none"
timeout 300 python bench.py $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split', d['value'], 'layouts/s', json.dumps(d.get('kernel_breakdown_ms')))" | tee -a $O/lngemm_probe.txt
LDM_DEV=1 LDM_X3_LNGEMM=0 timeout 300 python bench.py $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split r04 structure', d['value'], 'layouts/s')" | tee -a $O/lngemm_probe.txt
bash tools/pmc_sq.sh $O/sq_counters_split.txt split 4 > /dev/null 2>&1; grep -E "^void ldm::lngemm|matrix pipes|wave-cycle|effective clock" $O/sq_counters_split.txt | head -8
