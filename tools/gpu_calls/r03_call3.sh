#!/bin/bash
set -u
OUT=gpurun_out/r03_call3; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -s -x -k "fused_loop or teacher_forced or ragged or loop_greedy or sampler or posterior" > $OUT/tests1.log 2>&1; tail -25 $OUT/tests1.log
for cfg in "LDM_STACK_LOOP=1" "LDM_STACK_LOOP=0"; do
  echo "== $cfg"; env $cfg timeout 150 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-traffic --modes none > $OUT/bench_$(echo $cfg | tr ' =' '__').json 2>> $OUT/err.log; python - <<PY
import json
d=json.load(open("$OUT/bench_$(echo $cfg | tr ' =' '__').json"))
print(d["value"], d["ms_per_step"], d.get("roofline",{}), d.get("kernel_breakdown_ms"))
PY
done
echo "== config 3"; for cfg in "LDM_STACK_LOOP=1" "LDM_STACK_LOOP=0"; do env $cfg timeout 150 python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --modes none --no-roofline > $OUT/bench_c3_$(echo $cfg | tr ' =' '__').json 2>> $OUT/err.log; python -c "
import json; d=json.load(open('$OUT/bench_c3_$(echo $cfg | tr ' =' '__').json')); print('$cfg', d['value'], d['ms_per_step'])"; done
tail -5 $OUT/err.log
