#!/bin/bash
# Dev tool (GPU box): FETCH_SIZE / WRITE_SIZE of rocprofv3 against known byte counts (tools/microbench/pmc_calib.hip).
# Usage: tools/pmc_calibrate.sh <out.txt>
set -u
ROOT=$(pwd)
OUTTXT=$1
D=$(mktemp -d /tmp/ldm_calib_XXXX)
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $D/$c -o run -- $ROOT/tools/microbench/pmc_calib > $D/$c.log 2>&1 )
done
python - "$D" "$OUTTXT" <<'PY'
import csv, glob, sys
d, out = sys.argv[1], sys.argv[2]
known = {"calib_read_x4": 1 << 30, "calib_read_x1": 1 << 30, "calib_read_lds_dma": 1 << 30,
         "calib_read_acc_rows": (((1 << 30) // (464 * 4)) // 128 * 128) * 464 * 4, "calib_write_x4": 1 << 30, "calib_write_x1": 1 << 30}
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{d}/{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == c:
                k = row["Kernel_Name"].split("(")[0]
                vals.setdefault(k, {})[c] = vals.get(k, {}).get(c, 0.0) + float(row["Counter_Value"])
with open(out, "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- tools/microbench/pmc_calib: every kernel moves a 1-GiB\n"
            "# buffer (4x the Infinity Cache) exactly once; counters are reported in KiB; ratio = counter bytes / known bytes\n")
    f.write(f"{'kernel':24s} {'known MB':>10s} {'FETCH MB':>10s} {'ratio':>7s} {'WRITE MB':>10s} {'ratio':>7s}\n")
    for k, kb in known.items():
        v = vals.get(k, {})
        fe, wr = v.get("FETCH_SIZE", float("nan")) * 1024, v.get("WRITE_SIZE", float("nan")) * 1024
        f.write(f"{k:24s} {kb / 1e6:10.1f} {fe / 1e6:10.1f} {fe / kb:7.3f} {wr / 1e6:10.1f} {wr / kb:7.3f}\n")
print(open(out).read())
PY
rm -rf $D
