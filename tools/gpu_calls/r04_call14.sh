#!/bin/bash
# r04: split GEMM, operands through registers (cfg 6: 256x128, cfg 7: 128x128) vs the LDS-DMA form (cfg 2), one box; parity of cfg 6
O=gpurun_out/r04_call14; mkdir -p $O
LDM_DEV=1 LDM_X3_CFG=6 timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -k "split" 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-traffic --modes none --no-extras --steps 5 --warmup 1 --precision split"
for c in 2 6 7 2 6; do
  LDM_DEV=1 LDM_X3_CFG=$c $B 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernel_breakdown_ms'); print('x3cfg $c', d['value'], {n: round(v) for n, v in k.items() if n.startswith('gemm')})" | tee -a $O/x3_regs_ab.txt
done
tail -2 $O/err.log
