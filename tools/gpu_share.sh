#!/bin/bash
# GPU box: shared-ring variants of the fp32 stream (flat_f32_share 1 / 2), timing + tests
set -u
OUT=gpurun_out/${1:-share}
mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_flat_sq8.py -x -q -k "${TESTS:-stream}" < /dev/null 2>&1 | tail -5 > $OUT/test.log
for sh in 1 2; do
  echo "== flat_f32_share $sh" >> $OUT/sweep.log
  SHARE=$sh METRIC=0 K=100 NQS=${NQS:-128,256,384,1000} timeout 300 python tools/flat_nq_sweep.py < /dev/null >> $OUT/sweep.log 2>&1
done
( cd /tmp && export TMPDIR=/tmp && SHARE=2 METRIC=0 K=100 NQS=256 timeout 300 rocprofv3 --kernel-trace -d /tmp/fsprof -o p -- python $GRAFT_REPO_ROOT/tools/flat_nq_sweep.py < /dev/null > $GRAFT_REPO_ROOT/$OUT/prof_run.log 2>&1 )
timeout 60 python tools/prof_kernels.py /tmp/fsprof < /dev/null > $OUT/kernels.txt 2>&1
cat $OUT/test.log $OUT/sweep.log; head -8 $OUT/kernels.txt | cut -c1-170
