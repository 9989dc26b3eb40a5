#!/bin/bash
# GPU call 28: FFN GEMM1 as one AGPR accumulator chain (LDM_STACK_DBG=1) vs two VGPR chains.
set -u
OUT=gpurun_out/r02_call28
mkdir -p $OUT
timeout 300 python tools/kernel_ab.py "LDM_STACK_DBG=0" "LDM_STACK_DBG=1" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
