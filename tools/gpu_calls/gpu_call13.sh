#!/bin/bash
# GPU call 13: attention outputs parked in AGPRs (fused layer) — parity subset, phase probe, bench.
set -u
OUT=gpurun_out/r02_call13
mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -k "fast or denoiser or full_batch_512_one" 2>&1 | tail -4 | tee $OUT/pytest.txt
LDM_ATTN_TM=1 timeout 200 python tools/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/phase.txt
timeout 300 python bench.py --modes none --no-cpu-baseline --no-traffic --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_breakdown_ms'], d['roofline']['frac'], d['gemm_mfma_utilisation'])" | tee $OUT/bench.txt
