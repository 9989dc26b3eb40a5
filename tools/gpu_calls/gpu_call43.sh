#!/bin/bash
# GPU call 43: attn_mfma_k masks padded keys in every key tile (sequences shorter than 97 tokens in the fast mode);
# the other-geometry tests and the generation test (generation 0 runs this kernel at S = 125).
set -u
OUT=gpurun_out/r02_call43
mkdir -p $OUT
timeout 85 python -m pytest tests/test_hip_parity.py -m gpu -q -k "other_geometries or generations or ragged" 2>&1 | tail -8 | tee $OUT/pytest.txt
