#!/bin/bash
# GPU call 16: sync-wait probes of the stream layer kernel (per tile / slab / FFN chunk).
set -u
OUT=gpurun_out/r02_call16
mkdir -p $OUT
LDM_ATTN_TM=1 timeout 200 python tools/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/phase.txt
