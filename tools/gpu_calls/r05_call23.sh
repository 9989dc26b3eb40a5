#!/bin/bash
# Dev tool (GPU box): the loop kernel built without SLP vectorisation (no v_pk_*_f32 beside the MFMAs) against the shipped build, same box,
# alternating runs: bench headline (fast, random sampling) at 10 timed steps each.
set -u
O=gpurun_out/${1:-r05_call23}; mkdir -p $O
export TMPDIR=/tmp
Q="--steps 10 --warmup 2 --no-extras --no-cpu-baseline --no-traffic --modes none"
run() { "$@" python bench.py $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG', d['value'], 'layouts/s', d['ms_per_step'], 'ms/step', d['tokens_sha256'][:12] if 'tokens_sha256' in d else '')" | tee -a $O/ab.txt; }
for i in 1 2 3; do
  TAG=shipped run env
  TAG=variant run env LDM_DEV=1 LDM_HIP_LIB=$PWD/layout_dm_amd/libldm_hip_abl_${VARIANT:-noslp}.so
done
