#!/bin/bash
# Dev tool (GPU box): what the lngemm epilogue's remaining ~44 us per linear1 launch are: variants 16 (no global stores) / 32 (stores aimed at
# an L2-resident target) / 8 (no epilogue) of the measurement build.
set -u
O=gpurun_out/${1:-r05_call19}; mkdir -p $O
export TMPDIR=/tmp
for m in 0 16 32 8 0; do
  LDM_DEV=1 LDM_HIP_LIB=$PWD/layout_dm_amd/libldm_hip_abl_lngemm.so LDM_LNGEMM_ABL=$m timeout 120 python tools/lngemm_probe.py 10 2>/dev/null | tail -1 | sed "s/^/abl=$m /" | tee -a $O/lngemm_variants.txt
done
