#!/bin/bash
# GPU call 42 (last of round 2): the whole GPU suite at HEAD (after the removal of the tile-by-tile layer generation, the
# dev-hook split and the create-time validation; includes the new other-geometry tests), then the RCCL plumbing of
# bench.py with one rank on the N>1 workload (config 4: 1024 layouts per GPU).
set -u
OUT=gpurun_out/r02_call42
mkdir -p $OUT
timeout 230 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $OUT/pytest.txt
LDM_BENCH_FORCE_DIST=1 timeout 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
  --master-port 29517 bench.py --gpus 1 --config 4 --steps 3 --warmup 1 --modes none --no-traffic --no-cpu-baseline \
  > $OUT/bench_rccl_1rank_config4.json 2> $OUT/bench_rccl.err
echo "bench rc=$?"; tail -c 1500 $OUT/bench_rccl_1rank_config4.json; tail -3 $OUT/bench_rccl.err
