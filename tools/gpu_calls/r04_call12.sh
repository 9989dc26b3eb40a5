#!/bin/bash
# r04: relation SGD — restructured (this tree) vs the first loop-kernel version (tools/ab/libldm_hip_prev.so), same box, interleaved
O=gpurun_out/r04_call12; mkdir -p $O
for i in 1 2; do
  for lib in new prev; do
    if [ $lib = prev ]; then export LDM_HIP_LIB=tools/ab/libldm_hip_prev.so; else unset LDM_HIP_LIB; fi
    echo "== $lib"; python /root/repo/tools/gpu_calls/rel_time.py 2>&1 | grep -v amdgpu.ids | tail -2
  done
done | tee $O/relation_sgd_ab.txt
