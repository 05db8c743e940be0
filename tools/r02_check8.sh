#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/r02h
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONUNBUFFERED=1
i=0
for pmc in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf /tmp/prof_pmc
  timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d /tmp/prof_pmc -o u8 -- python $REPO/tools/u8_one.py > /tmp/pmc_run.log 2>&1
  echo "pmc [$pmc] rc=$?"
  python $REPO/tools/pmc_summary.py /tmp/prof_pmc gfilter > $OUT/pmc_$i.json
  cat $OUT/pmc_$i.json | head -40
done
