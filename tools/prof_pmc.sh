#!/bin/bash
# rocprofv3 counter passes (kernel-trace + --pmc only, one group per run) of one command, per-kernel means printed.
# usage: prof_pmc.sh <tag> <kernel-name-filter> <cmd...>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; FILT=$2; shift; shift
mkdir -p $R/gpurun_out
: > $R/gpurun_out/${TAG}_pmc.txt
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_VMEM_WR" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_WAIT_IFETCH"; do
  rm -rf /tmp/pp_$TAG
  timeout ${PROF_TIMEOUT:-240} rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d /tmp/pp_$TAG -o x -- "$@" > /tmp/pp_$TAG.log 2>&1 < /dev/null
  echo "pmc [$pmc] rc=$?"
  python3 $R/tools/pmc_summary.py /tmp/pp_$TAG "$FILT" | python3 -c "
import json,sys
j=json.load(sys.stdin)
for k,v in j.get('counters',{}).items():
    print(k[:80]); print('   ', ' '.join('%s=%s' % (a,b) for a,b in v.items()))
" | tee -a $R/gpurun_out/${TAG}_pmc.txt
done
