#!/bin/bash
# r04: relation SGD with LDS-resident centres + per-node incidence lists: parity, then same-box A/B against the first loop-kernel version
O=gpurun_out/r04_call13; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_r04_parity.py -m gpu -q -k "relation" 2>&1 | tail -3
for i in 1 2; do
  for lib in new prev; do
    if [ $lib = prev ]; then export LDM_HIP_LIB=tools/ab/libldm_hip_prev.so; else unset LDM_HIP_LIB; fi
    echo "== $lib"; python /root/repo/tools/gpu_calls/rel_time.py 2>&1 | grep -v amdgpu.ids | tail -2
  done
done | tee $O/relation_sgd_ab.txt
