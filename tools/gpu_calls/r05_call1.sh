#!/bin/bash
# Dev tool (GPU box, via gpurun): r05 first call — the whole GPU suite at the new defaults + the driver's bench command.
set -u
TAG=${1:-r05_call1}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 1100 python -m pytest tests -m gpu -q -x --durations=15 > $O/pytest.log 2>&1; tail -25 $O/pytest.log
grep -h "^\[" $O/pytest.log | head -5
timeout 900 python -m pytest tests/test_getcond_gpu.py tests/test_r04_parity.py -m gpu -q -s -k "getcond or default or auto_selection" > $O/pytest_new_verbose.log 2>&1; grep -h "^\[" $O/pytest_new_verbose.log | head -40
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json; tail -3 $O/bench.err
python - <<'PY' "$O/bench.json"
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"], "ref-prec", d.get("reference_precision_layouts_per_s"), "fp32", d.get("fp32_mfma_layouts_per_s"))
print("cpu_baseline", json.dumps(d.get("cpu_baseline"))[:600])
print("batch_shapes", json.dumps(d.get("batch_shapes"))[:900])
print("weight_sensitivity", json.dumps(d.get("weight_sensitivity"))[:1600])
print("config", json.dumps(d["config"])[:1200])
print("line bytes", len(json.dumps(d)))
PY
