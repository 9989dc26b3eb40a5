#!/bin/bash
# Dev tool (GPU box): compile-time timing variants of kernels_lngemm.hip's loop (measurement build, tools/build_measurement_variants.py lngemm).
# LDM_LNGEMM_ABL mask: 1 no MFMAs, 2 no fragment reads / counted waits, 4 no weight DMA, 8 no epilogue (sum, transpose, stores).
set -u
O=gpurun_out/${1:-r05_call16}; mkdir -p $O
export TMPDIR=/tmp LDM_DEV=1 LDM_HIP_LIB=$PWD/layout_dm_amd/libldm_hip_abl_lngemm.so
for m in 0 14 12 10 6 8 4 2 1 15 0; do
  LDM_LNGEMM_ABL=$m timeout 120 python tools/lngemm_probe.py 10 2>/dev/null | tail -1 | sed "s/^/abl=$m /" | tee -a $O/lngemm_variants.txt
done
