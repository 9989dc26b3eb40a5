#!/bin/bash
O=gpurun_out/r03_call30; mkdir -p $O
B="python bench.py --no-extras --no-cpu-baseline --no-traffic --no-roofline --precision exact --steps 3 --modes none"
run() { local label=$1; shift
  env "$@" 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['value'], d['ms_per_step'])"
}
for i in 1 2; do
  run new_slots5  LDM_X=1 $B
  run new_slots4  LDM_GEMM32_SLOTS=4 $B
  run prev_slots5 LDM_HIP_LIB=tools/ab/libldm_hip_prev.so $B
  run prev_slots4 LDM_HIP_LIB=tools/ab/libldm_hip_prev.so LDM_GEMM32_SLOTS=4 $B
done
