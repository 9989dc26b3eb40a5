#!/bin/bash
# GPU call 17: 3-stage weight ring (DMA two tiles ahead), hand-pipelined attention core, V bias folded into b_out;
# row-I/O micro-benchmark.
set -u
OUT=gpurun_out/r02_call17
mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -x -k "denoiser or fast or full_batch_512_one" 2>&1 | tail -6 | tee $OUT/pytest.txt
timeout 300 python tools/kernel_ab.py "LDM_FUSED_ATTN=3" "LDM_FUSED_ATTN=5" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
LDM_ATTN_TM=1 timeout 200 python tools/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/phase.txt
timeout 300 python bench.py --modes none --no-cpu-baseline --no-traffic --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_breakdown_ms'], d['roofline']['frac'], d['gemm_mfma_utilisation'])" | tee -a $OUT/bench.txt
timeout 120 tools/microbench/rowio 2>&1 | tee $OUT/rowio.txt
