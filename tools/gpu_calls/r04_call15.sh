#!/bin/bash
# r04: L2 hit rate / request counters of the split GEMMs (is gemm16x3_k bound by L2 misses or by the fill path?)
O=gpurun_out/r04_call15; mkdir -p $O
export TMPDIR=/tmp PMC_PRECISION=split PMC_STEPS=2
D=$(mktemp -d /tmp/ldm_l2_XXXX); R=$(pwd)
( cd /tmp && timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $D/p1 -o run -- python $R/tools/pmc_probe.py > $D/p1.log 2>&1 )
python - "$D" <<'PY' | tee $O/l2_counters_split.txt
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"].split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("# rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -- python tools/pmc_probe.py (split, 2 steps, 512 layouts); means per launch")
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("TCC_REQ_sum", [0]))):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    h, mi = m.get("TCC_HIT_sum", 0), m.get("TCC_MISS_sum", 0)
    print(f"{k:62s} REQ {m.get('TCC_REQ_sum', 0):.4g} HIT {h:.4g} MISS {mi:.4g} hit rate {h / max(h + mi, 1):.3f} EA_RDREQ {m.get('TCC_EA0_RDREQ_sum', 0):.4g}")
PY
tail -3 $D/p1.log
