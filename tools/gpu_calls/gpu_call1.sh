#!/bin/bash
# GPU call 1 (round 2): A/B of the new prologue/epilogue/register-exchange variants + wave skew, phase probes,
# full GPU test suite, default bench line.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r02_call1
mkdir -p $OUT
export TMPDIR=/tmp
echo "== A/B" | tee $OUT/ab.txt
timeout 600 python tools/kernel_ab.py \
  "LDM_ATTN_V=0 LDM_FFN_V=0" \
  "LDM_ATTN_V=0 LDM_FFN_V=1" \
  "LDM_ATTN_V=1 LDM_FFN_V=1" \
  "LDM_ATTN_V=3 LDM_FFN_V=1" \
  "LDM_ATTN_V=3 LDM_FFN_V=1 LDM_ATTN_SKEW=1 LDM_FFN_SKEW=1" \
  "LDM_ATTN_V=3 LDM_FFN_V=1 LDM_ATTN_SKEW=2 LDM_FFN_SKEW=2" \
  "LDM_ATTN_V=3 LDM_FFN_V=1 LDM_ATTN_SKEW=4 LDM_FFN_SKEW=4" \
  "LDM_ATTN_V=3 LDM_FFN_V=1 LDM_FFN_VAR=3" 2>&1 | tee -a $OUT/ab.txt
echo "== phase probes (new default)" | tee $OUT/phase.txt
LDM_FFN_DBG=3 LDM_ATTN_TM=1 timeout 200 python tools/phase_probe.py 2>&1 | tee -a $OUT/phase.txt
echo "== phase probes (r01 variants)" | tee -a $OUT/phase.txt
LDM_FFN_V=0 LDM_ATTN_V=0 LDM_FFN_DBG=3 LDM_ATTN_TM=1 timeout 200 python tools/phase_probe.py 2>&1 | tee -a $OUT/phase.txt
echo "== pytest"
timeout 900 python -m pytest tests -m gpu -q -rA 2>&1 | tail -150 > $OUT/pytest.log; tail -40 $OUT/pytest.log
echo "== bench"
timeout 420 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json; tail -5 $OUT/bench.err
cp gpurun_out/fast_mode_parity.json $OUT/ 2>/dev/null
