#!/bin/bash
# GPU call 15: what paces an LDS-fed MFMA stream (micro-benchmark); per-tile sync probes of the stream layer kernel;
# tile accumulators in AGPRs (LDM_LAYER_V=1).
set -u
OUT=gpurun_out/r02_call15
mkdir -p $OUT
timeout 120 tools/microbench/mfma_feed 2>&1 | tee $OUT/mfma_feed.txt
LDM_FUSED_ATTN=5 LDM_ATTN_TM=1 timeout 200 python tools/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/phase.txt
LDM_FUSED_ATTN=5 LDM_LAYER_V=1 LDM_ATTN_TM=1 timeout 200 python tools/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/phase_acc_agpr.txt
timeout 300 python tools/kernel_ab.py "LDM_FUSED_ATTN=5" "LDM_FUSED_ATTN=5 LDM_LAYER_V=1" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
