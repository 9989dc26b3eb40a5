#!/bin/bash
# r04: split mode at HEAD (256x256 tiles): (lanes, chunk) sweep on one box + single-lane rocprofv3 stats (durations not
# inflated by the other lane's kernels)
O=gpurun_out/r04_call22; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-traffic --modes none --no-extras --steps 4 --warmup 1 --precision split"
run() { $B 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" | tee -a $O/split_lanes_chunk_sweep.txt; }
echo "# label layouts/s ms_per_step (config 2: 512 layouts x 100 steps, split mode)" > $O/split_lanes_chunk_sweep.txt
run "default(lanes2,chunk256)"
export LDM_DEV=1
for L in 1 2 3 4; do for C in 128 256 512; do
  export LDM_LANES=$L LDM_CHUNK=$C; run "lanes${L}_chunk${C}"
done; done
unset LDM_CHUNK; export LDM_LANES=1
bash tools/rocprof_stats.sh $O/rocprof_stats_split_one_lane.txt --precision split --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-traffic --modes none > /dev/null 2>&1
unset LDM_DEV LDM_LANES
run "default_again"
cat $O/rocprof_stats_split_one_lane.txt | head -12
