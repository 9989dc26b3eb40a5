#!/bin/bash
# r04 first GPU pass: new parity tests (trained-like points, config-5 shape T=200, fast_verified rebuild, knobs) + the full
# bench line with the new keys
O=gpurun_out/r04_call1; mkdir -p $O
timeout 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 900 python -m pytest tests/test_r04_parity.py tests/test_fast_verified.py -m gpu -q -s > $O/pytest_r04.log 2>&1; echo "pytest rc=$?"
grep -E "^\[|passed|failed|Error|error" $O/pytest_r04.log | tail -80
timeout 900 python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 3000 $O/bench.err | tail -5
python - <<'PY'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r04_call1/bench.json') if l.startswith('{')][-1])
    print('headline', d['value'], d['roofline']['frac'], d['config']['library'])
    for k,v in d.get('modes',{}).items(): print('mode',k,v['value'], v.get('verification'), {kk:vv for kk,vv in v.get('nondegenerate',{}).items() if kk.startswith('from')})
    print('config2_verbatim', d.get('config2_verbatim'))
    for k,v in d.get('configs',{}).items(): print('cfg',k,v['value'])
    print('sens', d.get('weight_sensitivity'), 'scaling_point', d.get('scaling_point'))
except Exception as e: print('parse failed', e)
PY
