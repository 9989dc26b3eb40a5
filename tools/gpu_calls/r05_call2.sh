#!/bin/bash
# Dev tool (GPU box, via gpurun): r05 call 2 — the row-resident LayerNorm + x3 GEMM (kernels_lngemm.hip) in the split mode:
# parity (every split / auto / metrics / get_cond / relation test), then a same-box A/B against the r04 structure.
set -u
TAG=${1:-r05_call2}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "split or metrics or getcond or relation_teacher or auto_selection or default_path or precision_report or smoke" > $O/pytest_split.log 2>&1; tail -6 $O/pytest_split.log
Q="--precision split --steps 5 --warmup 1 --no-extras --no-cpu-baseline --no-traffic --modes none"
timeout 300 python bench.py $Q > $O/bench_split_lngemm.json 2> $O/bench_split_lngemm.err; tail -2 $O/bench_split_lngemm.err
LDM_DEV=1 LDM_X3_LNGEMM=0 timeout 300 python bench.py $Q > $O/bench_split_r04.json 2> $O/bench_split_r04.err; tail -2 $O/bench_split_r04.err
python - "$O" <<'PY'
import json, sys
o = sys.argv[1]
for name in ("bench_split_lngemm", "bench_split_r04"):
    try:
        d = json.loads(open(f"{o}/{name}.json").read().strip().splitlines()[-1])
        print(name, d["value"], "layouts/s", d["config"]["library"].get("kernels"), d["config"]["library"].get("knobs"))
        print("   ", json.dumps(d.get("kernel_breakdown_ms")))
        print("   ", json.dumps(d.get("roofline"))[:400])
    except Exception as e:
        print(name, "FAILED", e)
PY
bash tools/rocprof_stats.sh $O/rocprof_stats_split.txt --precision split --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-traffic --modes none > /dev/null 2>&1; head -14 $O/rocprof_stats_split.txt
