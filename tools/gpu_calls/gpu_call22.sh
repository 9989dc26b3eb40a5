#!/bin/bash
# GPU call 22: round-2 measurements of the stream layer kernel: new kernel-generation test, the driver's bench line (all
# numerics modes, live PMC traffic, cpu baseline), config 3, rocprofv3 kernel stats (default = lanes + graphs; sequential).
set -u
OUT=gpurun_out/r02_call22
mkdir -p $OUT
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -k "generations" 2>&1 | tail -3 | tee $OUT/pytest_new.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], d['roofline'], {k:(v.get('value'), (v.get('roofline') or {}).get('frac')) for k,v in d['modes'].items()}, d['cpu_baseline'])"; tail -2 $OUT/bench.err
timeout 400 python bench.py --config 3 --modes none --no-cpu-baseline --steps 3 > $OUT/bench_config3.json 2> $OUT/bench_config3.err; python -c "
import json; d=json.load(open('$OUT/bench_config3.json')); print(d['metric'], d['value'], d['ms_per_step'], d['config'])"
bash tools/rocprof_stats.sh $OUT/kernel_stats_default.txt --steps 2 --warmup 1 --modes none --no-cpu-baseline --no-traffic --no-roofline | head -12
bash tools/rocprof_stats.sh $OUT/kernel_stats_lanes1_nograph.txt --steps 2 --warmup 1 --modes none --no-cpu-baseline --no-traffic --no-roofline --lanes 1 --no-graph | head -12
