#!/bin/bash
# GPU call 24: lane / chunk sweep with the stack kernel (LDM_FUSED_ATTN=6) at B=512.
set -u
OUT=gpurun_out/r02_call24
mkdir -p $OUT
for cfg in "2 256 50" "2 256 100" "2 256 200" "4 128 50" "4 128 100" "3 128 70" "2 128 100"; do
set -- $cfg
LDM_FUSED_ATTN=6 LDM_LANES=$1 LDM_CHUNK=$2 LDM_LANE_OFFSET_US=$3 timeout 300 python bench.py --modes none --no-cpu-baseline --no-traffic --no-roofline --steps 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lanes=$1 chunk=$2 offset_us=$3 :', d['value'], d['ms_per_step'])" | tee -a $OUT/sweep.txt
done
