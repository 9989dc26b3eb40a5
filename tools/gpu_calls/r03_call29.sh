#!/bin/bash
# same-box A/B: conflict-free swizzle ((row >> 2) & 3) of the fp32 GEMM's LDS-DMA operand images vs the first one ((row >> 1) & 3)
O=gpurun_out/r03_call29; mkdir -p $O
B="python bench.py --no-extras --no-cpu-baseline --no-traffic --precision exact --steps 3 --modes none"
run() { local label=$1; shift
  env "$@" 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['value'], d['ms_per_step'], {k:round(v,1) for k,v in sorted(d.get('kernel_breakdown_ms',{}).items(), key=lambda kv:-kv[1])[:6]})"
}
timeout 200 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "exact and (denoiser or geometr)" 2>&1 | tail -2
for i in 1 2; do
  run new  LDM_X=1 $B
  run prev LDM_HIP_LIB=tools/ab/libldm_hip_prev.so $B
done
timeout 100 tools/microbench/gemm32 2>&1 | grep 'M=32000\|DMA\|pers G'
