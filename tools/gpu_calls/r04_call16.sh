#!/bin/bash
# r04: split GEMM with the LDS-transposed (coalesced) epilogue vs the direct epilogue (tools/ab/libldm_hip_prev.so), one box
O=gpurun_out/r04_call16; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -k "split" 2>&1 | tail -2
LDM_DEV=1 LDM_X3_CFG=6 timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -k "split and golden" 2>&1 | tail -1
B="python bench.py --no-cpu-baseline --no-traffic --modes none --no-extras --steps 5 --warmup 1 --precision split"
run() { $B 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernel_breakdown_ms'); print('$1', d['value'], {n: round(v) for n, v in k.items() if n.startswith('gemm')})" | tee -a $O/x3_epilogue_ab.txt; }
for i in 1 2; do
  unset LDM_HIP_LIB LDM_DEV LDM_X3_CFG; run "new_dma_256x128"
  export LDM_DEV=1 LDM_X3_CFG=6; run "new_regs_256x128"; unset LDM_DEV LDM_X3_CFG
  export LDM_HIP_LIB=tools/ab/libldm_hip_prev.so; run "prev_direct_epilogue"; unset LDM_HIP_LIB
done
tail -2 $O/err.log
