#!/bin/bash
# GPU call 29: round-2 measurements at HEAD (spill-free stack kernel as default): the whole GPU suite, the driver's bench line
# (all numerics modes, live PMC traffic, cpu baseline), config 3, rocprofv3 kernel stats (default = lanes + graphs; sequential).
set -u
OUT=gpurun_out/r02_call29
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $OUT/pytest.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], d['roofline'], {k:(v.get('value'), (v.get('roofline') or {}).get('frac')) for k,v in d['modes'].items()}, d['cpu_baseline'], d.get('gemm_mfma_utilisation'))"; tail -2 $OUT/bench.err
timeout 400 python bench.py --config 3 --modes none --no-cpu-baseline --steps 3 > $OUT/bench_config3.json 2> $OUT/bench_config3.err; python -c "
import json; d=json.load(open('$OUT/bench_config3.json')); print(d['metric'], d['value'], d['ms_per_step'], d['config'])"
bash tools/rocprof_stats.sh $OUT/kernel_stats_default.txt --steps 2 --warmup 1 --modes none --no-cpu-baseline --no-traffic --no-roofline | head -8
bash tools/rocprof_stats.sh $OUT/kernel_stats_lanes1_nograph.txt --steps 2 --warmup 1 --modes none --no-cpu-baseline --no-traffic --no-roofline --lanes 1 --no-graph | head -8
LDM_ATTN_TM=1 timeout 200 python tools/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/phase.txt
