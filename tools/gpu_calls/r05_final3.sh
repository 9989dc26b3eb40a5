#!/bin/bash
# Dev tool (GPU box): last check at HEAD — the GPU suite, the bench line exactly as the driver runs it (no flags), and the N > 1 code path
# (RCCL init / all_gather / barrier / all_reduce) with one rank.
set -u
O=gpurun_out/${1:-r05_final3}; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; tail -2 $O/bench.err
LDM_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --config 4 --steps 5 --warmup 1 --no-cpu-baseline --no-traffic --modes none --no-extras > $O/bench_rccl_1rank_config4.json 2> $O/rccl.err; echo "rccl rc=$?"; tail -c 400 $O/bench_rccl_1rank_config4.json
