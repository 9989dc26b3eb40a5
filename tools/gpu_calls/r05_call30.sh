#!/bin/bash
# Dev tool (GPU box): is the GEMM prologue of the linear2 + AdaLN + in_proj launch waiting for its A rows (HBM)?  Measurement build,
# LDM_LNGEMM_ABL=128: every stage re-reads the A fragments of stage 0 (L2-resident; wrong numerics) — per-launch times next to the product's.
set -u
O=gpurun_out/r05_call30; mkdir -p $O
export TMPDIR=/tmp
for m in 0 128 0 128; do
  LDM_DEV=1 LDM_HIP_LIB=$PWD/layout_dm_amd/libldm_hip_abl_lngemm.so LDM_LNGEMM_ABL=$m timeout 120 python tools/lngemm_probe.py 10 2>/dev/null | tail -1 | sed "s/^/abl=$m /" | tee -a $O/variants.txt
done
