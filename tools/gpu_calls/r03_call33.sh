#!/bin/bash
# cond=relation per-step path: token-major internal log-probability buffer + coalesced relation kernel vs the previous build
O=gpurun_out/r03_call33; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_parity.py tests/test_reference_cond_variants.py -m gpu -q -x -k "relation or cond or posterior or sampler" 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-traffic --no-roofline --modes none --steps 3"
for i in 1 2; do
  for lib in new prev; do
    if [ $lib = prev ]; then export LDM_HIP_LIB=tools/ab/libldm_hip_prev.so; else unset LDM_HIP_LIB; fi
    $B 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['configs']; print('$lib', 'fast', d['value'], 'relation', c['relation']['value'], c['relation'].get('kernel_breakdown_ms'), 'refinement', c['refinement']['value'])"
  done
done
