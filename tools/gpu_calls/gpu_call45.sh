#!/bin/bash
# GPU call 45: ragged batches (1 / 257 / 300 / 511 layouts) in the shipping configuration: graph == eager, cut invariance.
set -u
OUT=gpurun_out/r02_call45
mkdir -p $OUT
timeout 60 python -m pytest tests/test_hip_parity.py -m gpu -q -k "ragged_batches_fast" 2>&1 | tail -25 | tee $OUT/pytest.txt
