#!/bin/bash
# GPU call 21: prologue with the row loads two batches deep; queue depth 10 for the FFN (DBG=2) / slab (DBG=4) streams.
set -u
OUT=gpurun_out/r02_call21
mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -x -k "denoiser or fast_mode or full_batch_512_one" 2>&1 | tail -3 | tee $OUT/pytest.txt
timeout 400 python tools/kernel_ab.py "LDM_LAYER_DBG=1" "LDM_LAYER_DBG=0" "LDM_LAYER_DBG=2" "LDM_LAYER_DBG=4" "LDM_LAYER_DBG=6" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
LDM_ATTN_TM=1 timeout 200 python tools/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/phase.txt
for v in 0 6; do
LDM_LAYER_DBG=$v timeout 300 python bench.py --modes none --no-cpu-baseline --no-traffic --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dbg=$v', d['value'], d['ms_per_step'], d['kernel_breakdown_ms'], d['roofline']['frac'], d['gemm_mfma_utilisation'])" | tee -a $OUT/bench.txt
done
