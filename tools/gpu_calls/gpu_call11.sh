#!/bin/bash
# GPU call 11: fused layer — phase probe, full GPU test suite, full default bench line.
set -u
OUT=gpurun_out/r02_call11
mkdir -p $OUT
echo "== phase probe (fused layer)" | tee $OUT/phase.txt
LDM_ATTN_TM=1 timeout 200 python tools/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/phase.txt
echo "== pytest"
timeout 900 python -m pytest tests -m gpu -q -rA 2>&1 | tail -150 > $OUT/pytest.log; tail -5 $OUT/pytest.log
echo "== bench"
timeout 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json; tail -3 $OUT/bench.err
cp gpurun_out/fast_mode_parity.json $OUT/ 2>/dev/null
