#!/bin/bash
# Dev tool (GPU box, via gpurun): the round's evidence run -> gpurun_out/$1/ (copied into profiles/$1_*).
set -u
TAG=${1:-r04_final}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json; tail -2 $O/bench.err
if [ "${LIGHT:-0}" = "1" ]; then exit 0; fi   # LIGHT=1: the suite and the bench line only (no profiler passes)
Q="--no-extras --no-cpu-baseline --no-traffic --modes none"
bash tools/rocprof_stats.sh $O/rocprof_stats_config2.txt --steps 3 --warmup 1 $Q > /dev/null 2>&1
bash tools/rocprof_stats.sh $O/rocprof_stats_exact.txt --precision exact --steps 2 --warmup 1 $Q > /dev/null 2>&1
bash tools/rocprof_stats.sh $O/rocprof_stats_split.txt --precision split --steps 2 --warmup 1 $Q > /dev/null 2>&1
# cond=relation (512 layouts, T = 100, the extras row of bench.py) under the profiler: ONE launch per sampling call
cat > /tmp/rel_probe.py <<'PY'
import sys, torch
sys.path.insert(0, "/root/repo")
from layout_dm_amd import synthetic as SP
from layout_dm_amd.binding import Engine
from layout_dm_amd.diffusion import timestep_schedule
spec = SP.SPECS["rico25"]; B = 512
e = Engine(n_category=spec.n_category, precision="fast", max_batch=B)
e.load_state_dict(SP.synth_state_dict(spec, seed=0))
cond, graph = SP.synth_cond_relation(spec, B, seed=0)
plan = e.make_relation(graph, SP.linear_bin_centres(spec.n_bin), [16, 16, 31, 31], 3e6, 3, B)
tm, tp = timestep_schedule(100, 100)
c = {"seq": cond["seq"], "mask": cond["mask"], "type": "relation"}
for i in range(3):
    tok = torch.from_numpy(cond["seq"]).int().cuda()
    e.sample_loop(tok, tm, tp, {"name": "random", "temperature": 1.0}, cond=c, seed=i, relation=plan)
torch.cuda.synchronize()
PY
D=$(mktemp -d /tmp/ldm_rel_XXXX); R=$(pwd)
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $D -o run -- python /tmp/rel_probe.py > $D/log.txt 2>&1 ) || true
( cd $R && cp /tmp/rel_probe.py $O/rel_probe.py )
python - "$D" "$O/rocprof_stats_relation.txt" <<'PY'
import glob, sqlite3, sys
d, out = sys.argv[1], sys.argv[2]
f = open(out, "w")
f.write("# rocprofv3 --kernel-trace --stats -- python rel_probe.py (3 sampling calls: rico25 cond=relation, 512 layouts, T = 100, random)\n# columns: name | total_calls | total_duration(us) | average(us) | percentage\n")
for db in glob.glob(d + "/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    top = [t for t in tabs if "top_kernels" in t] or [t for t in tabs if "kernel" in t.lower() and "summary" in t.lower()]
    for t in top[:1]:
        for row in con.execute(f"select * from '{t}'"):
            f.write(" | ".join([str(row[0])[:110]] + [str(x) for x in row[1:]]) + "\n")
f.close()
print(open(out).read()[:1200])
PY
bash tools/pmc_sq.sh $O/sq_counters_fast_loop.txt fast 100 > /dev/null 2>&1
bash tools/pmc_sq.sh $O/sq_counters_exact.txt exact 4 > /dev/null 2>&1
bash tools/pmc_sq.sh $O/sq_counters_split.txt split 4 > /dev/null 2>&1
head -10 $O/rocprof_stats_config2.txt; head -8 $O/rocprof_stats_split.txt; head -26 $O/sq_counters_fast_loop.txt
