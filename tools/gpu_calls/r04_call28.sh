#!/bin/bash
# (record of an experiment: the prefetch variants of kernels_stack.hip / ldm_pipes.h it timed were NOT kept in the source — results and the
# description of what was built: profiles/r04_call26_27_loop_kernel_weight_stream_window_and_prefetch.txt)
# r04: full-coverage L2 prefetch of the FFN weight stream (two 128-byte-stride LDS-DMAs per wave and chunk = every line of the
# wave's 16 KiB, 3 or 8 chunks ahead) vs the build without it, one box; parity of the new build
O=gpurun_out/r04_call28; mkdir -p $O
B="timeout 120 python bench.py --no-cpu-baseline --no-traffic --modes none --no-extras --steps 5 --warmup 2"
run() { $B 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" | tee -a $O/ffn_prefetch_ab.txt; }
echo "# label layouts/s ms_per_launch (config 2, fast mode, 512 layouts x 100 steps)" > $O/ffn_prefetch_ab.txt
for i in 1 2 3; do
  run "prefetch_dist3"
  LDM_HIP_LIB=$PWD/layout_dm_amd/libldm_hip_abl_pfd8.so run "prefetch_dist8"
  LDM_HIP_LIB=$PWD/layout_dm_amd/libldm_hip_abl_nopf.so run "no_prefetch"
done
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_r04_parity.py -m gpu -q -x -k "fast or loop" 2>&1 | tail -3 | tee $O/pytest_fast.txt
