#!/bin/bash
# Dev tool (GPU box): what bounds kernels_lngemm.hip — the kernel next to ablations of itself (LDM_LNGEMM_ABL bit mask).
set -u
O=gpurun_out/${1:-r05_call5}; mkdir -p $O
for abl in 0 1 2 4 3 6 7; do
  LDM_DEV=1 LDM_LNGEMM_ABL=$abl timeout 120 python tools/lngemm_probe.py 10 2>/dev/null | tail -1 | tee -a $O/lngemm_ablations.txt
done
LDM_DEV=1 LDM_X3_LNGEMM=0 timeout 120 python tools/lngemm_probe.py 10 2>/dev/null | tail -1 | tee -a $O/lngemm_ablations.txt
