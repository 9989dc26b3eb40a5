#!/bin/bash
# r04: split GEMM tile ORDER A/B (L2 locality) on the 256x128 tiles
O=gpurun_out/r04_call7; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-traffic --modes none --no-extras --steps 5 --warmup 1 --precision split"
for g in 0 2 4 8 0 4; do
  LDM_DEV=1 LDM_X3_CFG=2 LDM_X3_GRP=$g $B 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernel_breakdown_ms'); print('x3 256x128 grp $g', d['value'], {n: round(v) for n, v in k.items() if n.startswith('gemm')})" | tee -a $O/x3_grp_ab.txt
done
LDM_DEV=1 LDM_X3_CFG=3 LDM_X3_GRP=2 $B 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernel_breakdown_ms'); print('x3 128x256 grp 2', d['value'], {n: round(v) for n, v in k.items() if n.startswith('gemm')})" | tee -a $O/x3_grp_ab.txt
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -k "split and golden" 2>&1 | tail -2
