#!/bin/bash
# GPU call 7: (a) tests with FFN CH=2 + attention V=3 defaults; (b) SQ / LDS PMC counters of the shipping fused kernels:
# are the ~45 cycles per MFMA of the weight-tile runs LDS bank conflicts?
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r02_call7
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -rA -k "fast or fid or denoiser or full_batch" 2>&1 | tail -30 > $OUT/pytest.log; tail -4 $OUT/pytest.log
cd /tmp
rocprofv3 -L 2>/dev/null | grep -i "lds\|SQ_INSTS_VMEM\|SQ_ACTIVE_INST\|SQ_WAIT\|MFMA\|SQ_BUSY\|SQ_WAVE_CYCLES\|TCP_\|TA_BUSY" | head -150 > $OUT/counters_available.txt
pass() {  # name counters...
  n=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$n -o run -- python $ROOT/tools/pmc_probe.py > $OUT/pmc_$n.log 2>&1
  echo "pass $n rc=$?"
}
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
pass sq SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
cd $ROOT
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
grep -A10 "ffn_fused2\|qkv_attn" $OUT/pmc_summary.txt | head -80
rm -rf $OUT/pmc_lds $OUT/pmc_sq
