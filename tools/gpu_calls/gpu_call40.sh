#!/bin/bash
# GPU call 40: per-layer stream kernel (LDM_FUSED_ATTN=5) now uses the all-VGPR attention core: generation test + parity subset.
set -u
OUT=gpurun_out/r02_call40
mkdir -p $OUT
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -k "generations" 2>&1 | tail -2 | tee $OUT/pytest.txt
LDM_FUSED_ATTN=5 timeout 600 python -m pytest tests -m gpu -q -x -k "denoiser or fast_mode" 2>&1 | tail -2 | tee -a $OUT/pytest.txt
