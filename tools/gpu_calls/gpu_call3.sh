#!/bin/bash
# GPU call 3: concurrent lanes (per-lane hipGraphs, phase offset) vs chunk size; correctness of the laned loop.
set -u
OUT=gpurun_out/r02_call3
mkdir -p $OUT
run() {  # lanes chunk offset
  r=$(LDM_LANE_OFFSET_US=$3 timeout 200 python bench.py --lanes $1 --chunk $2 --modes none --no-cpu-baseline --no-traffic --no-roofline --steps 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "lanes=$1 chunk=$2 offset_us=$3 : $r" | tee -a $OUT/lanes.txt
}
run 1 512 0
run 1 256 0
run 2 256 0
run 2 256 50
run 2 128 0
run 2 128 40
run 2 128 80
run 4 128 25
run 4 64 20
run 3 128 30
echo "== B=1024 (config 4 shape)" | tee -a $OUT/lanes.txt
for cfg in "1 512 0" "2 256 50" "2 128 40" "4 128 25"; do
  set -- $cfg
  r=$(LDM_LANE_OFFSET_US=$3 timeout 200 python bench.py --config 4 --lanes $1 --chunk $2 --modes none --no-cpu-baseline --no-traffic --no-roofline --steps 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "B=1024 lanes=$1 chunk=$2 offset_us=$3 : $r" | tee -a $OUT/lanes.txt
done
echo "== correctness with lanes" | tee $OUT/pytest_lanes.txt
LDM_LANES=2 LDM_CHUNK=128 timeout 600 python -m pytest tests -m gpu -q -k "full_batch_512 or loop or fast_vs_exact or relation or dropin" 2>&1 | tail -8 | tee -a $OUT/pytest_lanes.txt
LDM_LANES=3 LDM_CHUNK=2 timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -k "loop or relation or dropin or ragged" 2>&1 | tail -8 | tee -a $OUT/pytest_lanes.txt
