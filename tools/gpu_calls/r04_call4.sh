#!/bin/bash
# r04: cond=relation INSIDE the one-launch loop kernel (stack_stream_k<.,2,true>) — parity + same-box A/B against the per-step
# path (LDM_DEV=1 LDM_REL_LOOP=0, itself with the fused relation_step_k tail and the LDS-staged edges)
O=gpurun_out/r04_call4; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_r04_parity.py -m gpu -q -s -k "relation" 2>&1 | grep -E "^\[|passed|failed|Error|assert" | tail -25
cat > /tmp/rel_ab.py <<'PY'
import json, subprocess, sys, os
def run(env):
    e = dict(os.environ, **env)
    p = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--no-cpu-baseline", "--no-traffic", "--modes", "none"], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    except Exception:
        print(p.stderr[-1500:]); raise
    c = d["configs"]
    return d["value"], c["relation"]["value"], c["relation"].get("kernel_breakdown_ms"), c["5_relation_T200"]["value"], c["refinement"]["value"]
for i in range(2):
    for name, env in (("loop_kernel", {}), ("per_step_fused_tail", {"LDM_DEV": "1", "LDM_REL_LOOP": "0"}), ("per_step_three_launch", {"LDM_DEV": "1", "LDM_REL_LOOP": "0", "LDM_REL_FUSED": "0"})):
        print(name, run(env), flush=True)
PY
python /tmp/rel_ab.py 2>&1 | tee $O/relation_ab.txt
