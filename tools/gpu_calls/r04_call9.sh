#!/bin/bash
# r04: split GEMM, 128-byte operand rows per K tile (BK = 64) vs 64-byte rows (BK = 32)
O=gpurun_out/r04_call9; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-traffic --modes none --no-extras --steps 5 --warmup 1 --precision split"
for c in 2 5 2 5; do
  LDM_DEV=1 LDM_X3_CFG=$c $B 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernel_breakdown_ms'); print('x3cfg $c', d['value'], {n: round(v) for n, v in k.items() if n.startswith('gemm')})" | tee -a $O/x3_bk_ab.txt
done
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -k "split" 2>&1 | tail -2
tail -2 $O/err.log
