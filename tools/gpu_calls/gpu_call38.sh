#!/bin/bash
# GPU call 38: attention core with the next key tile's exponentials in the shadow of the O^T MFMAs; tile-1' DMA three steps
# behind the slab barrier.  Parity subset, phase probe, bench.
set -u
OUT=gpurun_out/r02_call38
mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -x -k "denoiser or fast_mode or full_batch_512_one or generations" 2>&1 | tail -3 | tee $OUT/pytest.txt
LDM_ATTN_TM=1 timeout 200 python tools/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/phase.txt
timeout 300 python bench.py --modes none --no-cpu-baseline --no-traffic --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_breakdown_ms'], d['roofline']['frac'], d['gemm_mfma_utilisation'])" | tee -a $OUT/bench.txt
