#!/bin/bash
# GPU call 30: MFMA-feed micro-benchmark with the streams' per-tile protocol (barrier, vmcnt(8), epilogue stores).
set -u
OUT=gpurun_out/r02_call30
mkdir -p $OUT
timeout 120 tools/microbench/mfma_feed 2>&1 | tee $OUT/mfma_feed.txt
