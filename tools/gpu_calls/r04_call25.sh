#!/bin/bash
# r04: split GEMM: non-temporal policy on the activation fills (FFN2 / out-proj: operands two column tiles read once each) A/B
O=gpurun_out/r04_call25; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-traffic --modes none --no-extras --steps 4 --warmup 1 --precision split"
run() { $B 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernel_breakdown_ms') or {}; print('$1', d['value'], {n: round(v,1) for n, v in k.items() if n.startswith('gemm')})" | tee -a $O/split_gemm_nt_ab.txt; }
echo "# label layouts/s {GEMM class: ms per 100 steps, eager single-lane profile pass}" > $O/split_gemm_nt_ab.txt
for i in 1 2; do
  run "default"
  LDM_DEV=1 LDM_X3_CFG=10 run "nt_A_ffn2_attnout"
  LDM_DEV=1 LDM_X3_CFG=11 run "nt_A_all"
done
LDM_DEV=1 LDM_X3_CFG=10 timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -k "split" 2>&1 | tail -1 | tee -a $O/split_gemm_nt_ab.txt
