#!/bin/bash
# Dev tool (GPU box): phase timers of kernels_lngemm.hip (LDM_LNGEMM_TM=1) next to the plain per-launch times.
set -u
O=gpurun_out/${1:-r05_call12}; mkdir -p $O
LDM_DEV=1 timeout 120 python tools/lngemm_probe.py 10 2>/dev/null | tail -1 | tee -a $O/lngemm_phases.txt
LDM_DEV=1 LDM_LNGEMM_TM=1 timeout 120 python tools/lngemm_probe.py 10 2>/dev/null | tail -2 | tee -a $O/lngemm_phases.txt
