#!/bin/bash
# GPU box: stream tests, nq sweep (IP + L2), kernel trace of one sweep.  Everything under its own timeout; nothing reads stdin.
set -u
OUT=gpurun_out/${1:-flat_f32}
mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_flat_sq8.py -x -q -k "${TESTS:-stream}" < /dev/null 2>&1 | tail -15 > $OUT/test.log
for m in 0 1; do
  METRIC=$m K=100 NQS=${NQS:-1,8,32,64,96,128,192,256,512,1000} timeout 300 python tools/flat_nq_sweep.py < /dev/null >> $OUT/sweep.log 2>&1
done
( cd /tmp && export TMPDIR=/tmp && METRIC=0 K=100 NQS=${PNQS:-1,64,96} timeout 300 rocprofv3 --kernel-trace -d /tmp/fsprof -o p -- python $GRAFT_REPO_ROOT/tools/flat_nq_sweep.py < /dev/null > $GRAFT_REPO_ROOT/$OUT/prof_run.log 2>&1 )
timeout 60 python tools/prof_kernels.py /tmp/fsprof < /dev/null > $OUT/kernels.txt 2>&1
cat $OUT/test.log $OUT/sweep.log; head -20 $OUT/kernels.txt | cut -c1-170
