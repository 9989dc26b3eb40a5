#!/bin/bash
# Dev tool (GPU box): the GEMM prologue with three A register sets (A rows requested two stages ahead, counted vmcnt at the stage barrier, 3-deep
# fragment queue): parity / repeatability subset, per-launch times next to the measurement variant 128 (L2-resident A), split bench.
set -u
O=gpurun_out/${1:-r05_call31}; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "repeatable or (split and not b1024) or precision_report or alternative_structures or default_path" > $O/pytest_split.log 2>&1; tail -4 $O/pytest_split.log
for m in 0 128; do
  LDM_DEV=1 LDM_HIP_LIB=$PWD/layout_dm_amd/libldm_hip_abl_lngemm.so LDM_LNGEMM_ABL=$m timeout 120 python tools/lngemm_probe.py 10 2>/dev/null | tail -1 | sed "s/^/abl=$m /" | tee -a $O/probe.txt
done
Q="--precision split --steps 5 --warmup 1 --no-extras --no-cpu-baseline --no-traffic --modes none"
for i in 1 2; do
  timeout 300 python bench.py $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default (level 4)', d['value'], 'layouts/s', json.dumps(d.get('kernel_breakdown_ms')))" | tee -a $O/probe.txt
  LDM_DEV=1 LDM_X3_LNGEMM=1 timeout 300 python bench.py $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('level 1', d['value'], 'layouts/s')" | tee -a $O/probe.txt
done
