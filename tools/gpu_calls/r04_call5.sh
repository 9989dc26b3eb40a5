#!/bin/bash
# r04: the split mode (fp16 x 3) on the LDS-DMA GEMM (gemm16x3_k): parity + throughput next to the exact mode and the r03 split GEMM
O=gpurun_out/r04_call5; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -s -k "split" 2>&1 | grep -E "^\[|passed|failed|Error|assert" | tail -20
B="python bench.py --no-cpu-baseline --no-traffic --modes none --no-extras --steps 5 --warmup 1"
for v in split exact; do
  $B --precision $v 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['roofline']['kernel'], d['roofline']['frac'], d.get('kernel_breakdown_ms'))" | tee -a $O/split_vs_exact.txt
done
LDM_DEV=1 LDM_SPLIT_GEMM=old $B --precision split 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split_r03_gemm', d['value'], d.get('kernel_breakdown_ms'))" | tee -a $O/split_vs_exact.txt
tail -3 $O/err.log
