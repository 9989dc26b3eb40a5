#!/bin/bash
# GPU call 8: what costs GEMM1 45 cycles per MFMA?  phase probes of timing ablations (wrong numerics by design):
# LDM_FFN_DBG=3 normal, 4 no in-loop DMA, 5 constant B operand in GEMM1, 6 both.  + fid golden re-test.
set -u
OUT=gpurun_out/r02_call8
mkdir -p $OUT
for d in 3 4 5 6; do
  echo "== LDM_FFN_DBG=$d" | tee -a $OUT/ffn_ablation.txt
  LDM_FFN_DBG=$d timeout 150 python tools/phase_probe.py 2>&1 | grep -v amdgpu.ids | grep -A8 "^ffn" | tee -a $OUT/ffn_ablation.txt
done
timeout 300 python -m pytest tests/test_fid_parity.py -m gpu -q 2>&1 | tail -4 | tee $OUT/pytest_fid.txt
