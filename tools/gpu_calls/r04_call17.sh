#!/bin/bash
# r04: what bounds gemm16x3_k? the kernel next to its own ablations (fills only / reads + MFMAs only / fills + MFMAs)
O=gpurun_out/r04_call17; mkdir -p $O
python tools/gemm_x3_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_x3_probe.txt
