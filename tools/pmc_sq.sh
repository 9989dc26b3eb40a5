#!/bin/bash
# Dev tool (GPU box): SQ counters (MFMA busy, wave-cycle buckets, LDS conflicts) + GRBM_GUI_ACTIVE of the kernels of one
# sampling call (tools/pmc_probe.py), two --pmc passes of 8 SQ counters, kernel durations from --kernel-trace.
# Usage: tools/pmc_sq.sh <out.txt> <precision> <steps>
set -u
ROOT=$(pwd)
OUTTXT=$1; PREC=$2; STEPS=$3
D=$(mktemp -d /tmp/ldm_sq_XXXX)
export TMPDIR=/tmp PMC_PRECISION=$PREC PMC_STEPS=$STEPS
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D/p$i -o run -- python $ROOT/tools/pmc_probe.py > $D/p$i.log 2>&1 )
done
python - "$D" "$OUTTXT" "$PREC" "$STEPS" <<'PY'
import csv, glob, sys, collections
d, out, prec, steps = sys.argv[1:5]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"].split("(")[0][:70] + " grid=" + row.get("Grid_Size", "?")][row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        gs = row.get("Grid_Size") or str(int(row.get("Grid_Size_X", 0)) * int(row.get("Grid_Size_Y", 1)) * int(row.get("Grid_Size_Z", 1)))
        dur[row["Kernel_Name"].split("(")[0][:70] + " grid=" + gs].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
with open(out, "w") as f:
    f.write(f"# rocprofv3 --pmc (two passes) --kernel-trace -- python tools/pmc_probe.py   PMC_PRECISION={prec} PMC_STEPS={steps} (512 layouts)\n")
    f.write("# means per launch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE cycles\n")
    ks = sorted(acc, key=lambda k: -sum(dur.get(k, [0])))
    for k in ks[:12]:
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        us = sum(dur[k]) / max(len(dur[k]), 1)
        f.write(f"\n{k}   launches/pass={len(dur[k]) // 2}   mean duration {us:.1f} us (profiled passes)\n")
        for n, v in sorted(c.items()):
            f.write(f"   {n:28s} {v:.5g}\n")
        if c.get("SQ_WAVE_CYCLES"):
            w = c["SQ_WAVE_CYCLES"]
            f.write(f"   -> MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_WAVE_CYCLES) = {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4 * w):.3f}"
                    f"   (only meaningful with one wave per SIMD)\n")
            f.write(f"   -> wave-cycle buckets: WAIT_ANY {c.get('SQ_WAIT_ANY', 0) / w:.2f}  WAIT_INST_ANY {c.get('SQ_WAIT_INST_ANY', 0) / w:.2f}  ACTIVE_INST_ANY {c.get('SQ_ACTIVE_INST_ANY', 0) / w:.2f}\n")
        if c.get("GRBM_GUI_ACTIVE") and us:
            f.write(f"   -> effective clock = GRBM_GUI_ACTIVE / (8 XCDs x duration) = {c['GRBM_GUI_ACTIVE'] / 8 / us:.0f} MHz\n")
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE"):
            f.write(f"   -> matrix pipes busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) = {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * c['GRBM_GUI_ACTIVE'] / 8):.3f}\n")
        if c.get("SQ_LDS_IDX_ACTIVE"):
            f.write(f"   -> LDS bank conflicts = {c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:.4f} of the LDS-active cycles\n")
print(open(out).read()[:6000])
PY
rm -rf $D
