#!/bin/bash
# GPU call 44: reverse loop at other geometries (S = 35 / 150, d_model 192) against the oracle on identical uniforms.
set -u
OUT=gpurun_out/r02_call44
mkdir -p $OUT
timeout 60 python -m pytest tests/test_hip_parity.py -m gpu -q -k "loop_other_geometries" 2>&1 | tail -25 | tee $OUT/pytest.txt
