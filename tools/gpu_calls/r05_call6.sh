#!/bin/bash
# Dev tool (GPU box): a quick parity + timing round for kernels_lngemm.hip: the split-mode logits / step tests, the probe with and
# without the kernel's components, the split bench line.
set -u
O=gpurun_out/${1:-r05_call6}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "split or precision_report" > $O/pytest_split.log 2>&1; tail -3 $O/pytest_split.log
for abl in 0 7 1 4; do
  LDM_DEV=1 LDM_LNGEMM_ABL=$abl timeout 120 python tools/lngemm_probe.py 10 2>/dev/null | tail -1 | tee -a $O/lngemm_ablations.txt
done
Q="--precision split --steps 5 --warmup 1 --no-extras --no-cpu-baseline --no-traffic --modes none"
timeout 300 python bench.py $Q > $O/bench_split.json 2> $O/bench_split.err
python - "$O" <<'PY'
import json, sys
d = json.loads(open(f"{sys.argv[1]}/bench_split.json").read().strip().splitlines()[-1])
print("split", d["value"], "layouts/s", json.dumps(d.get("kernel_breakdown_ms")))
PY
