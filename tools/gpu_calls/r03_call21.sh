#!/bin/bash
# same-box A/B: branch-free top-p / top-k walk (config 3) and the 160-wide fp32 tiles under the 4-waves-per-SIMD GEMM
O=gpurun_out/r03_call21; mkdir -p $O
B="python bench.py --no-extras --no-cpu-baseline --no-roofline --no-traffic"
run() { # label, env..., args
  local label=$1; shift
  env "$@" 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['value'], d['ms_per_step'])"
}
for i in 1 2; do
  run c3_new  LDM_X=1 $B --config 3 --steps 3 --modes none
  run c3_prev LDM_HIP_LIB=tools/ab/libldm_hip_prev.so $B --config 3 --steps 3 --modes none
done
run c3_topk_new  LDM_X=1 $B --config 3 --sampling top_k --steps 3 --modes none
run c3_topk_prev LDM_HIP_LIB=tools/ab/libldm_hip_prev.so $B --config 3 --sampling top_k --steps 3 --modes none
run c2_new  LDM_X=1 $B --steps 3 --modes none
run c2_prev LDM_HIP_LIB=tools/ab/libldm_hip_prev.so $B --steps 3 --modes none
for i in 1 2; do
  run exact_128 LDM_X=1 $B --precision exact --steps 3 --modes none
  run exact_160 LDM_GEMM32_WIDE=1 $B --precision exact --steps 3 --modes none
done
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_fast_mode_parity.py tests/test_config34_shapes.py -m gpu -q -x -k "sampl or top or draw or kind or config or posterior or strong" 2>&1 | tail -3
