#!/bin/bash
# r04: instruction-mix microbenchmark (what the chip sustains for the loop kernel's VALU : MFMA : LDS ratio and duty) +
# the two fixed r04 tests
O=gpurun_out/r04_call2; mkdir -p $O
tools/microbench/mix_feed > $O/mix_feed.txt 2>&1; cat $O/mix_feed.txt
tools/microbench/mfma_feed 2>&1 | head -12 > $O/mfma_feed.txt; head -4 $O/mfma_feed.txt
timeout 600 python -m pytest tests/test_r04_parity.py -m gpu -q -k "fast_mode_and_auto or dev_knobs" 2>&1 | tail -3
