#!/bin/bash
# exact mode: chunk size / number of concurrent chunk pipelines (lanes) of the per-step path, same box
O=gpurun_out/r03_call27; mkdir -p $O
B="python bench.py --no-extras --no-cpu-baseline --no-traffic --no-roofline --precision exact --steps 3 --modes none"
run() { local label=$1; shift
  env "$@" 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['value'], d['ms_per_step'])"
}
run chunk256_lanes2 LDM_X=1 $B
run chunk128_lanes4 LDM_X=1 $B --chunk 128 --lanes 4
run chunk128_lanes3 LDM_X=1 $B --chunk 128 --lanes 3
run chunk128_lanes2 LDM_X=1 $B --chunk 128 --lanes 2
run chunk171_lanes3 LDM_X=1 $B --chunk 171 --lanes 3
run chunk256_lanes1 LDM_X=1 $B --lanes 1
run chunk512_lanes1 LDM_X=1 $B --chunk 512 --lanes 1
run chunk256_lanes2_slots4 LDM_GEMM32_SLOTS=4 $B
run chunk256_lanes2_slots6 LDM_GEMM32_SLOTS=6 $B
run chunk256_lanes2 LDM_X=1 $B
