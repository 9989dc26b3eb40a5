#!/bin/bash
# Dev tool (GPU box, via gpurun): GPU test suite, headline bench, rocprofv3 kernel stats and HBM-traffic PMC passes.
# Everything lands under gpurun_out/final/; the summaries are copied into profiles/ by hand.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 200 python bench.py > $OUT/bench_fast.json 2> $OUT/bench_fast.err; cat $OUT/bench_fast.json
timeout 120 python bench.py --batch 1024 --chunk 1024 --steps 3 --no-cpu-baseline --no-roofline > $OUT/bench_chunk1024.json 2>> $OUT/bench_fast.err; cat $OUT/bench_chunk1024.json
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/stats -o run -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/rocprof_stats.log 2>&1
python - <<PY
import glob, sqlite3
dbs = glob.glob("$OUT/stats/**/*.db", recursive=True)
out = open("$OUT/kernel_stats.txt", "w")
out.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline\n")
out.write("# columns: name, total_calls, total_duration(us), average(us), percentage\n")
for db in dbs:
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    top = [t for t in tabs if "top_kernels" in t] or [t for t in tabs if "kernel" in t.lower() and "summary" in t.lower()]
    for t in top[:1]:
        cols = [c[1] for c in con.execute(f"pragma table_info('{t}')")]
        out.write(f"# source table: {t} {cols}\n")
        for row in con.execute(f"select * from '{t}'"):
            out.write(" | ".join(str(x) for x in row) + "\n")
out.close()
print(open("$OUT/kernel_stats.txt").read()[:3000])
PY
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o run -- python $ROOT/tools/pmc_probe.py > $OUT/pmc_$c.log 2>&1
done
cd $ROOT
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1; grep -A3 "stack_stream\|posterior\|ln_rows" $OUT/pmc_summary.txt | head -60
rm -rf $OUT/stats $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
