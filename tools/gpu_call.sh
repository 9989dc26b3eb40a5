#!/bin/bash
# ONE parametrised script for everything that runs on the GPU box (r06; it replaces the 113 one-off tools/gpu_calls/*.sh of rounds 2-5,
# indexed in tools/gpu_calls/INDEX.md).  Run it THROUGH gpurun from the build container:
#     tools/gpu.sh [--timeout S] 'bash tools/gpu_call.sh <recipe> [args]'        (tools/gpu.sh builds first and refuses a tree that does not build)
# Everything lands under gpurun_out/<tag>/ (merged back by gpurun); what is kept is copied into profiles/<tag>_* by hand and indexed in
# profiles/README.md (claim -> file).
#
#   suite [tag] [pytest args]      the GPU test suite (-m gpu)                                   -> pytest.log
#   smoke [tag]                    __graft_entry__.smoke()                                       -> smoke.log
#   bench [tag] [bench.py args]    the bench line (default: the driver's --steps 20 --warmup 2)  -> bench.json, bench.err
#   stats [tag] <precision>        rocprofv3 --kernel-trace --stats of a short bench.py run      -> rocprof_stats_<precision>.txt
#   relstats [tag]                 the same for cond=relation (512 layouts, T = 100)             -> rocprof_stats_relation.txt
#   sq [tag] <precision> <steps>   SQ / LDS / GRBM counter passes (tools/pmc_sq.sh)              -> sq_counters_<precision>.txt
#   ab [tag] "<ENV=..>" "<ENV=..>"  same-box A/B of knob settings, one process each (tools/kernel_ab.py; PROBE_PREC selects the mode) -> ab.txt
#   probe [tag] <script.py> [args] any tools/*.py probe (attnout_probe.py, split_step_probe.py, lngemm_probe.py, ...)  -> <script>.log
#   evidence [tag]                 the round's evidence run: smoke + suite + bench + stats (fast, exact, split, mixed, hybrid) + relstats + sq (fast 100, exact 4, split 4, mixed 4, hybrid 4)
#                                  (LIGHT=1: smoke + suite + bench only)
set -u
RECIPE=${1:-evidence}; shift || true
TAG=${1:-r06_${RECIPE}}; [ $# -gt 0 ] && shift
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
Q="--no-extras --no-cpu-baseline --no-traffic --modes none"

suite()    { timeout 1500 python -m pytest tests -m gpu -q "$@" > $O/pytest.log 2>&1; tail -3 $O/pytest.log; }
smoke()    { timeout 400 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log; }
bench()    { if [ $# -eq 0 ]; then set -- --steps 20 --warmup 2; fi
             timeout 1500 python bench.py "$@" > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json; tail -2 $O/bench.err; }
stats()    { local p=${1:-fast}; local a="--precision $p --steps 2 --warmup 1"; [ "$p" = fast ] && a="--steps 3 --warmup 1"
             bash tools/rocprof_stats.sh $O/rocprof_stats_$p.txt $a $Q > /dev/null 2>&1; head -12 $O/rocprof_stats_$p.txt | cut -c1-200; }
sq()       { bash tools/pmc_sq.sh $O/sq_counters_$1.txt $1 ${2:-4} > /dev/null 2>&1; head -30 $O/sq_counters_$1.txt; }
relstats() {
  local D=$(mktemp -d /tmp/ldm_rel_XXXX) R=$(pwd)
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $D -o run -- python $R/tools/rel_probe.py > $D/log.txt 2>&1 ) || true
  python - "$D" "$O/rocprof_stats_relation.txt" <<'PY'
import glob, sqlite3, sys
d, out = sys.argv[1], sys.argv[2]
f = open(out, "w")
f.write("# rocprofv3 --kernel-trace --stats -- python tools/rel_probe.py (3 sampling calls: rico25 cond=relation, 512 layouts, T = 100, random)\n# columns: name | total_calls | total_duration(us) | average(us) | percentage\n")
for db in glob.glob(d + "/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    top = [t for t in tabs if "top_kernels" in t] or [t for t in tabs if "kernel" in t.lower() and "summary" in t.lower()]
    for t in top[:1]:
        for row in con.execute(f"select * from '{t}'"):
            f.write(" | ".join([str(row[0])[:110]] + [str(x) for x in row[1:]]) + "\n")
f.close()
print(open(out).read()[:800])
PY
  rm -rf $D
}

case $RECIPE in
  suite)    suite "$@" ;;
  smoke)    smoke ;;
  bench)    bench "$@" ;;
  stats)    stats "$@" ;;
  relstats) relstats ;;
  sq)       sq "$@" ;;
  ab)       timeout 1200 python tools/kernel_ab.py "$@" > $O/ab.txt 2>&1; cat $O/ab.txt ;;
  probe)    S=$1; shift; timeout 900 python tools/$S "$@" > $O/${S%.py}.log 2>&1; grep -v amdgpu.ids $O/${S%.py}.log | tail -40 ;;
  evidence) smoke; suite; bench
            if [ "${LIGHT:-0}" != "1" ]; then stats fast; stats exact; stats split; stats mixed; stats hybrid; relstats; sq fast 100; sq exact 4; sq split 4; sq mixed 4; sq hybrid 4; fi ;;
  *)        echo "unknown recipe $RECIPE"; exit 2 ;;
esac
