#!/bin/bash
# r04: split GEMM 256x256: interleaved (independent consecutive MFMAs) vs dependent MFMA order, one box
O=gpurun_out/r04_call19; mkdir -p $O
LDM_DEV=1 LDM_X3_CFG=8 timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -k "split" 2>&1 | tail -1
B="python bench.py --no-cpu-baseline --no-traffic --modes none --no-extras --steps 5 --warmup 1 --precision split"
run() { $B 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernel_breakdown_ms'); print('$1', d['value'], {n: round(v) for n, v in k.items() if n.startswith('gemm')})" | tee -a $O/x3_order_ab.txt; }
for i in 1 2; do
  export LDM_DEV=1 LDM_X3_CFG=8; run "256x256_interleaved"; export LDM_X3_CFG=9; run "256x256_dependent"; unset LDM_DEV LDM_X3_CFG
  run "256x128_interleaved"
done
