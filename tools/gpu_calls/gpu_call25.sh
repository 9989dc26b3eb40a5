#!/bin/bash
# GPU call 25: stack kernel — sub-phase probes of the layer entry and the sync waits; full GPU suite with LDM_FUSED_ATTN=6.
set -u
OUT=gpurun_out/r02_call25
mkdir -p $OUT
LDM_FUSED_ATTN=6 LDM_ATTN_TM=1 timeout 200 python tools/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/phase.txt
LDM_FUSED_ATTN=6 timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee $OUT/pytest.txt
