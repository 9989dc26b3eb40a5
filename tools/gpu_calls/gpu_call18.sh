#!/bin/bash
# GPU call 18d: VALU write -> MFMA operand read wait states (LDM_LAYER_DBG=128: s_nop 3 between the P cast and the MFMA).
set -u
OUT=gpurun_out/r02_call18
mkdir -p $OUT
PROBE_REPS=3 timeout 300 python tools/kernel_ab.py "LDM_FUSED_ATTN=5 LDM_LAYER_DBG=1" "LDM_FUSED_ATTN=5 LDM_LAYER_DBG=128" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab4.txt
