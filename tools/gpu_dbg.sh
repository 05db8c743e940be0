#!/bin/bash
set -u
OUT=gpurun_out/${1:-dbg}; mkdir -p $OUT; export PYTHONUNBUFFERED=1
for sh in ${SHS:-2}; do for dbg in ${DBGS:-0 16 32}; do
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/fsprof && DBG=$dbg SHARE=$sh METRIC=0 K=100 NQS=${NQ:-256} timeout 120 rocprofv3 --kernel-trace -d /tmp/fsprof -o p -- python $GRAFT_REPO_ROOT/tools/flat_nq_sweep.py < /dev/null > /dev/null 2>&1 )
  echo "share=$sh dbg=$dbg $(timeout 60 python tools/prof_kernels.py /tmp/fsprof < /dev/null | grep mshare | awk '{print $1, $3, $4}' | cut -c40-)" >> $OUT/dbg.log
done; done
cat $OUT/dbg.log
