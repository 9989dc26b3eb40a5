#!/bin/bash
# r04: split mode at HEAD (256x256, one accumulator): smoke + the tests that touch it; the N > 1 code path of bench.py with ONE rank
# over RCCL (torch.distributed.run --nproc-per-node 1, LDM_BENCH_FORCE_DIST=1): init, all_gather, barrier, per-rank stats
O=gpurun_out/r04_call20; mkdir -p $O
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_r04_parity.py tests/test_fast_verified.py -m gpu -q -k "split or auto or verified or trained_like" 2>&1 | tail -2
LDM_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --config 4 --steps 5 --warmup 1 --no-cpu-baseline --no-traffic --modes none --no-extras > $O/bench_rccl_1rank_config4.json 2> $O/rccl.err; echo "rccl rc=$?"; tail -c 600 $O/bench_rccl_1rank_config4.json
python bench.py --precision split --steps 5 --warmup 1 --no-cpu-baseline --no-traffic --modes none --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split', d['value'], d['roofline'], d['kernel_breakdown_ms'])"
