#!/bin/bash
# r04: split mode attention on the fp16 pipe (attn16x3_k) vs the fp32-MFMA kernel (LDM_ATTN32=direct), one box
O=gpurun_out/r04_call23; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i "split\|smoke" | tee $O/smoke.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "split or auto or verified" 2>&1 | tail -4 | tee $O/pytest_split.txt
B="python bench.py --no-cpu-baseline --no-traffic --modes none --no-extras --steps 4 --warmup 1 --precision split"
run() { $B 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernel_breakdown_ms') or {}; print('$1', d['value'], {n: round(v,1) for n, v in k.items()})" | tee -a $O/split_attention_ab.txt; }
echo "# label layouts/s {kernel class: ms per 100 steps, eager single-lane profile pass}" > $O/split_attention_ab.txt
for i in 1 2; do
  run "attn16x3"
  LDM_DEV=1 LDM_ATTN32=direct run "attn32_direct"
done
LDM_DEV=1 LDM_LANES=1 bash tools/rocprof_stats.sh $O/rocprof_stats_split_one_lane.txt --precision split --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-traffic --modes none > /dev/null 2>&1
head -12 $O/rocprof_stats_split_one_lane.txt
