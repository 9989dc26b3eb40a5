#!/bin/bash
# GPU call 41: SQ / LDS PMC counters of the shipping kernels at HEAD (stack kernel + posterior): MFMA busy, wave cycles,
# LDS bank conflicts, wait buckets.  Two --pmc passes (8 SQ counters each), --kernel-trace only.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r02_call41
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
pass() {
  n=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$n -o run -- python $ROOT/tools/pmc_probe.py > $OUT/pmc_$n.log 2>&1
  echo "pass $n rc=$?"
}
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
pass sq SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
cd $ROOT
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
grep -A17 "stack_stream_k\|posterior_sample_k" $OUT/pmc_summary.txt | head -60
rm -rf $OUT/pmc_lds $OUT/pmc_sq
