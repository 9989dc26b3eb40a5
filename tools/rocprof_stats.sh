#!/bin/bash
# Dev tool (GPU box): rocprofv3 --kernel-trace --stats of one bench.py invocation -> a text table of the top kernels
# (name | calls | total us | average us | %).  Usage: tools/rocprof_stats.sh <out.txt> <bench.py args...>
set -u
ROOT=$(pwd)
OUTTXT=$1; shift
D=$(mktemp -d /tmp/ldm_rocprof_XXXX)
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $D -o run -- python $ROOT/bench.py "$@" > $D/log.txt 2>&1 )
python - "$D" "$OUTTXT" "$*" <<'PY'
import glob, sqlite3, sys
d, out, args = sys.argv[1], sys.argv[2], sys.argv[3]
f = open(out, "w")
f.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py {args}\n# columns: name | total_calls | total_duration(us) | average(us) | percentage\n")
for db in glob.glob(d + "/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    top = [t for t in tabs if "top_kernels" in t] or [t for t in tabs if "kernel" in t.lower() and "summary" in t.lower()]
    for t in top[:1]:
        for row in con.execute(f"select * from '{t}'"):
            name = str(row[0])
            if len(name) > 110:
                name = name[:110] + "..."
            f.write(" | ".join([name] + [str(x) for x in row[1:]]) + "\n")
f.close()
print(open(out).read()[:2500])
PY
rm -rf $D
