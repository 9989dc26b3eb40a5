#!/bin/bash
# r04: split GEMM (256x256): is the operand-fill limit per CU or chip-wide?  one round on half / all of the CUs, + full size
O=gpurun_out/r04_call24; mkdir -p $O
LDM_DEV=1 timeout 600 python tools/gemm_x3_probe.py 5376 10752 32000 2>&1 | tee $O/gemm_x3_probe_256x256.txt
