#!/bin/bash
# GPU call 4: FFN version 2 (single read of the residual row) + lanes defaults: A/B, phase probe, full tests, bench.
set -u
OUT=gpurun_out/r02_call4
mkdir -p $OUT
echo "== A/B (256-layout launches)" | tee $OUT/ab.txt
timeout 400 python tools/kernel_ab.py "LDM_FFN_V=1" "LDM_FFN_V=2" 2>&1 | tee -a $OUT/ab.txt
echo "== phase probe (defaults, 256-layout launches)" | tee $OUT/phase.txt
LDM_FFN_DBG=3 LDM_ATTN_TM=1 timeout 200 python tools/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/phase.txt
echo "== pytest"
timeout 900 python -m pytest tests -m gpu -q -rA 2>&1 | tail -150 > $OUT/pytest.log; tail -5 $OUT/pytest.log
echo "== bench"
timeout 420 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json; tail -3 $OUT/bench.err
echo "== bench FFN_V=1 / lanes 1 for comparison"
LDM_FFN_V=1 timeout 200 python bench.py --modes none --no-cpu-baseline --no-traffic --no-roofline --steps 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ffn_v1', d['value'], d['ms_per_step'])" | tee -a $OUT/ab.txt
timeout 200 python bench.py --lanes 1 --modes none --no-cpu-baseline --no-traffic --no-roofline --steps 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lanes1', d['value'], d['ms_per_step'])" | tee -a $OUT/ab.txt
cp gpurun_out/fast_mode_parity.json $OUT/ 2>/dev/null
