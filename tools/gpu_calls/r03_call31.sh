#!/bin/bash
# same-box A/B: 64-row tiles (two per workgroup) for the one-round fp32 GEMMs (LDM_GEMM32_BM64=1) vs 128-row tiles
O=gpurun_out/r03_call31; mkdir -p $O
B="python bench.py --no-extras --no-cpu-baseline --no-traffic --precision exact --steps 3 --modes none"
run() { local label=$1; shift
  env "$@" 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['value'], d['ms_per_step'], {k:round(v,1) for k,v in sorted(d.get('kernel_breakdown_ms',{}).items(), key=lambda kv:-kv[1])[:5]})"
}
LDM_GEMM32_BM64=1 timeout 200 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "exact and (denoiser or ragged)" 2>&1 | tail -2
for i in 1 2; do
  run bm64 LDM_GEMM32_BM64=1 $B
  run bm128 LDM_X=1 $B
done
