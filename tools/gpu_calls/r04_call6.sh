#!/bin/bash
# r04: split GEMM tile-shape A/B (one box) + the FID-vs-reference harness
O=gpurun_out/r04_call6; mkdir -p $O
timeout 600 python -m pytest tests/test_fid_vs_reference.py -m gpu -q -s 2>&1 | grep -E "^\[|^    |passed|failed|Error|assert" | tee $O/fid_vs_reference.txt | tail -30
B="python bench.py --no-cpu-baseline --no-traffic --modes none --no-extras --steps 5 --warmup 1 --precision split"
for c in 0 1 2 3 4 0; do
  LDM_DEV=1 LDM_X3_CFG=$c $B 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernel_breakdown_ms'); print('x3cfg $c', d['value'], {n: round(v) for n, v in k.items() if n.startswith('gemm')})" | tee -a $O/x3_cfg_ab.txt
done
tail -2 $O/err.log
