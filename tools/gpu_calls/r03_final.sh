#!/bin/bash
# Dev tool (GPU box, via gpurun): the round's evidence run -> gpurun_out/r03_final/ (copied into profiles/r03_final_*).
set -u
O=gpurun_out/r03_final; mkdir -p $O
export TMPDIR=/tmp
timeout 700 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
Q="--no-extras --no-cpu-baseline --no-traffic --modes none"
bash tools/rocprof_stats.sh $O/rocprof_stats_config2.txt --steps 3 --warmup 1 $Q > /dev/null 2>&1
bash tools/rocprof_stats.sh $O/rocprof_stats_config3.txt --config 3 --steps 3 --warmup 1 $Q > /dev/null 2>&1
bash tools/rocprof_stats.sh $O/rocprof_stats_exact.txt --precision exact --steps 2 --warmup 1 $Q > /dev/null 2>&1
bash tools/pmc_sq.sh $O/sq_counters_fast_loop.txt fast 100 > /dev/null 2>&1
bash tools/pmc_sq.sh $O/sq_counters_exact.txt exact 4 > /dev/null 2>&1
timeout 250 tools/microbench/gemm32 > $O/gemm32_microbench.txt 2>&1
head -12 $O/rocprof_stats_config2.txt; head -8 $O/rocprof_stats_config3.txt; head -30 $O/sq_counters_fast_loop.txt
