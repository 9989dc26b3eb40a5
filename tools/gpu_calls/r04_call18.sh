#!/bin/bash
# r04: split mode with unscaled lo / ONE accumulator (+ weight pre-scale): parity; 256x256 tiles vs 256x128, one box
O=gpurun_out/r04_call18; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -s -k "split" 2>&1 | grep -E "^\[|passed|failed|assert" | tail -12
for c in 8 9; do LDM_DEV=1 LDM_X3_CFG=$c timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -k "split" 2>&1 | tail -1; done
B="python bench.py --no-cpu-baseline --no-traffic --modes none --no-extras --steps 5 --warmup 1 --precision split"
run() { $B 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernel_breakdown_ms'); print('$1', d['value'], {n: round(v) for n, v in k.items() if n.startswith('gemm')})" | tee -a $O/x3_one_acc_ab.txt; }
for i in 1 2; do
  unset LDM_HIP_LIB LDM_DEV LDM_X3_CFG; run "one_acc_256x128"
  export LDM_DEV=1 LDM_X3_CFG=8; run "one_acc_256x256_2x4"; export LDM_X3_CFG=9; run "one_acc_256x256_4x2"; unset LDM_DEV LDM_X3_CFG
  export LDM_HIP_LIB=tools/ab/libldm_hip_prev.so; run "prev_two_acc_256x128_direct_epilogue"; unset LDM_HIP_LIB
done
python tools/gemm_x3_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_x3_probe.txt
tail -2 $O/err.log
