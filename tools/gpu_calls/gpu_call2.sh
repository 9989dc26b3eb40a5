#!/bin/bash
# GPU call 2: is the prologue/epilogue an all-CU HBM burst?  Phase probes at B = 512 / 256 / 128 / 64 layouts per launch
# (fewer workgroups bursting at once), and whole-loop throughput vs chunk size.
set -u
OUT=gpurun_out/r02_call2
mkdir -p $OUT
for b in 512 256 128 64; do
  echo "== PROBE_B=$b" | tee -a $OUT/phase_vs_blocks.txt
  PROBE_B=$b LDM_FFN_DBG=3 LDM_ATTN_TM=1 timeout 120 python tools/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/phase_vs_blocks.txt
done
for c in 512 256 128; do
  echo "== chunk $c" | tee -a $OUT/chunk.txt
  timeout 200 python bench.py --chunk $c --modes none --no-cpu-baseline --no-traffic --steps 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_breakdown_ms'])" | tee -a $OUT/chunk.txt
done
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -k "relation" 2>&1 | tail -15 | tee $OUT/pytest_relation.txt
