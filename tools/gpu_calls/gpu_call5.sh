#!/bin/bash
# GPU call 5: FID + entry point + fp32-MFMA attention (exact mode) validation, config 3 bench, rocprofv3 kernel stats.
set -u
OUT=gpurun_out/r02_call5
mkdir -p $OUT
echo "== new tests first"
timeout 600 python -m pytest tests/test_fid_parity.py tests/test_entry_point.py -m gpu -q -rA 2>&1 | tail -25 | tee $OUT/pytest_new.txt
echo "== exact-mode A/B (VALU vs fp32-MFMA attention)" | tee $OUT/exact_ab.txt
for v in rows mfma; do
  LDM_ATTN32=$v timeout 300 python bench.py --precision exact --modes none --no-cpu-baseline --no-traffic --steps 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['kernel_breakdown_ms'])" | tee -a $OUT/exact_ab.txt
done
echo "== full pytest"
timeout 900 python -m pytest tests -m gpu -q -rA 2>&1 | tail -150 > $OUT/pytest.log; tail -6 $OUT/pytest.log
echo "== bench config 3"
timeout 400 python bench.py --config 3 --modes none --no-cpu-baseline --steps 3 > $OUT/bench_config3.json 2> $OUT/bench_config3.err; cat $OUT/bench_config3.json; tail -3 $OUT/bench_config3.err
echo "== rocprof stats"
bash tools/rocprof_stats.sh $OUT/kernel_stats_default.txt --steps 2 --warmup 1 --modes none --no-cpu-baseline --no-traffic --no-roofline
bash tools/rocprof_stats.sh $OUT/kernel_stats_lanes1_nograph.txt --steps 2 --warmup 1 --modes none --no-cpu-baseline --no-traffic --no-roofline --lanes 1 --no-graph
