#!/bin/bash
# r03 call 1: first hardware run of the fused step tail (LDM_STACK_POST=1) + A/B bench
set -u
OUT=gpurun_out/r03_call1; mkdir -p $OUT
export TMPDIR=/tmp
LDM_TEST_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -k experimental_fused_step_tail > $OUT/exp_test.log 2>&1; tail -15 $OUT/exp_test.log
for cfg in "LDM_STACK_POST=0" "LDM_STACK_POST=1" "LDM_STACK_POST=1 LDM_LANES=1" "LDM_STACK_POST=0 LDM_LANES=1"; do
  echo "== $cfg"; env $cfg timeout 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-traffic --modes none > $OUT/bench_$(echo $cfg | tr ' =' '__').json 2> $OUT/err.log; python - <<PY
import json
d=json.load(open("$OUT/bench_$(echo $cfg | tr ' =' '__').json"))
print(d["value"], d["ms_per_step"], d.get("roofline",{}).get("avg_launch_ms"), d.get("kernel_breakdown_ms"))
PY
done
echo "== config 3"; for cfg in "LDM_STACK_POST=0" "LDM_STACK_POST=1"; do env $cfg timeout 120 python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --modes none --no-roofline > $OUT/bench_c3_$(echo $cfg | tr ' =' '__').json 2>> $OUT/err.log; python -c "
import json; d=json.load(open('$OUT/bench_c3_$(echo $cfg | tr ' =' '__').json')); print('$cfg', d['value'], d['ms_per_step'])"; done
