#!/bin/bash
# r04: what does the weight stream cost the loop kernel?  Measurement builds (tools/build_measurement_variants.py): the FFN weight
# stream (65 % of the 23.4 MB per workgroup-step) re-reads a 64-KiB window (served by the XCD's L2: no fabric traffic) or a
# 16-KiB window (served by the CU's L1: no L2 -> CU traffic either); same instruction stream, same LDS-DMA count
O=gpurun_out/r04_call26; mkdir -p $O
B="timeout 120 python bench.py --no-cpu-baseline --no-traffic --modes none --no-extras --steps 5 --warmup 2"
run() { $B 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" | tee -a $O/ffn_window_ab.txt; }
echo "# label layouts/s ms_per_launch (config 2, fast mode, 512 layouts x 100 steps)" > $O/ffn_window_ab.txt
for i in 1 2; do
  run "shipped_library"
  LDM_HIP_LIB=$PWD/layout_dm_amd/libldm_hip_abl_ffnwin1.so run "ffn_window_64KiB_L2_served"
  LDM_HIP_LIB=$PWD/layout_dm_amd/libldm_hip_abl_ffnwin2.so run "ffn_window_16KiB_L1_served"
done
