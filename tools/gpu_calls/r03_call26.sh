#!/bin/bash
# same-box A/B: 160-wide LDS-DMA tiles for the N = 464 GEMMs (default) vs 128-wide (LDM_GEMM32_WIDE=0 / the previous build)
O=gpurun_out/r03_call26; mkdir -p $O
B="python bench.py --no-extras --no-cpu-baseline --no-traffic --precision exact --steps 3 --modes none"
run() { local label=$1; shift
  env "$@" 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print('$label', d['value'], d['ms_per_step'], {k:round(v,1) for k,v in sorted(d.get('kernel_breakdown_ms',{}).items(), key=lambda kv:-kv[1])[:6]})"
}
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "exact or denoiser or geometr or t200" 2>&1 | tail -4
for i in 1 2; do
  run new  LDM_X=1 $B
  run new_wide0 LDM_GEMM32_WIDE=0 $B
  run prev LDM_HIP_LIB=tools/ab/libldm_hip_prev.so $B
done
