#!/bin/bash
# Dev tool (GPU box): lngemm measurement variants, list in $1 (default: 0 64 16 32 0).
set -u
O=gpurun_out/r05_call21; mkdir -p $O
export TMPDIR=/tmp
for m in ${1:-0 64 16 32 0}; do
  LDM_DEV=1 LDM_HIP_LIB=$PWD/layout_dm_amd/libldm_hip_abl_lngemm.so LDM_LNGEMM_ABL=$m timeout 120 python tools/lngemm_probe.py 10 2>/dev/null | tail -1 | sed "s/^/abl=$m /" | tee -a $O/lngemm_variants.txt
done
