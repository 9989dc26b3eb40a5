#!/bin/bash
# r04: cond=relation in the loop kernel: SGD passes batched (4 rows per 16-lane group in flight, branch-free), the layout's graph
# (packed edges, centres, edge count, canvas box) staged once per launch instead of once per adjusted step; vs the previous build
O=gpurun_out/r04_call29; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "relation or config5" 2>&1 | tail -3 | tee $O/pytest_relation.txt
echo "# cond=c vs cond=relation, 512 layouts x 100 steps, one-launch loop (tools/gpu_calls/rel_time.py)" > $O/relation_ab.txt
for i in 1 2; do
  echo "new build:" | tee -a $O/relation_ab.txt; python tools/gpu_calls/rel_time.py 2>/dev/null | tee -a $O/relation_ab.txt
  echo "previous build:" | tee -a $O/relation_ab.txt; LDM_HIP_LIB=$PWD/tools/ab/libldm_hip_prev.so python tools/gpu_calls/rel_time.py 2>/dev/null | tee -a $O/relation_ab.txt
done
