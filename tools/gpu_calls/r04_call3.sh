#!/bin/bash
# r04: cond=relation per-step path with the fused adjusted-step tail (relation_step_k) — parity + same-box A/B against the
# three-launch form (LDM_DEV=1 LDM_REL_FUSED=0)
O=gpurun_out/r04_call3; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_r04_parity.py -m gpu -q -k "relation" 2>&1 | tail -15
cat > /tmp/rel_ab.py <<'PY'
import json, subprocess, sys, os
def run(env):
    e = dict(os.environ, **env)
    p = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--no-cpu-baseline", "--no-traffic", "--modes", "none"], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    c = d["configs"]
    return d["value"], c["relation"]["value"], c["relation"].get("kernel_breakdown_ms"), c["5_relation_T200"]["value"], c["refinement"]["value"]
for i in range(2):
    for name, env in (("fused", {}), ("three_launch", {"LDM_DEV": "1", "LDM_REL_FUSED": "0"})):
        print(name, run(env), flush=True)
PY
python /tmp/rel_ab.py 2>&1 | tee $O/relation_ab.txt
