#!/bin/bash
# GPU call 46 (the round's last seconds of GPU budget): fused step vs the reference's cond=cwh / partial / time_difference answers.
OUT=gpurun_out/r02_call46
mkdir -p $OUT
timeout 16 python -m pytest tests/test_reference_cond_variants.py -m gpu -q -s 2>&1 | tail -12 | tee $OUT/pytest.txt
