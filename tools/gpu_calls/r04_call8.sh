#!/bin/bash
# r04: SQ counters of the split mode's kernels (what limits gemm16x3_k?)
O=gpurun_out/r04_call8; mkdir -p $O
bash tools/pmc_sq.sh $O/sq_counters_split.txt split 4 > /dev/null 2>&1
head -75 $O/sq_counters_split.txt
