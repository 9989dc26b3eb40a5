#!/bin/bash
# Dev tool (GPU box): kernels_lngemm.hip with the step as one asm statement (fillers between the MFMAs): parity / repeatability, per-launch
# times, the compile-time variants of the measurement build (2 no fragment reads, 4 no DMA, 8 no epilogue), split bench A/B, SQ counters.
set -u
O=gpurun_out/${1:-r05_call17}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "repeatable or (split and not b1024) or precision_report" > $O/pytest_split.log 2>&1; tail -4 $O/pytest_split.log
LDM_DEV=1 timeout 120 python tools/lngemm_probe.py 10 2>/dev/null | tail -1 | tee -a $O/lngemm_probe.txt
LDM_DEV=1 LDM_X3_LNGEMM=0 timeout 120 python tools/lngemm_probe.py 10 2>/dev/null | tail -1 | tee -a $O/lngemm_probe.txt
for m in ${VARIANTS:-0 8 4 2 14 12 10 6}; do
  LDM_DEV=1 LDM_HIP_LIB=$PWD/layout_dm_amd/libldm_hip_abl_lngemm.so LDM_LNGEMM_ABL=$m timeout 120 python tools/lngemm_probe.py 10 2>/dev/null | tail -1 | sed "s/^/abl=$m /" | tee -a $O/lngemm_variants.txt
done
Q="--precision split --steps 5 --warmup 1 --no-extras --no-cpu-baseline --no-traffic --modes none"
timeout 300 python bench.py $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split', d['value'], 'layouts/s', json.dumps(d.get('kernel_breakdown_ms')))" | tee -a $O/lngemm_probe.txt
LDM_DEV=1 LDM_X3_LNGEMM=0 timeout 300 python bench.py $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split r04 structure', d['value'], 'layouts/s')" | tee -a $O/lngemm_probe.txt
if [ "${SQ:-1}" = 1 ]; then bash tools/pmc_sq.sh $O/sq_counters_split.txt split 4 > /dev/null 2>&1; grep -E "^void ldm::lngemm|matrix pipes|wave-cycle|effective clock" $O/sq_counters_split.txt | head -8; fi
