#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/r02j
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_ivf.py tests/test_gpu_opq.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log
timeout 300 python tools/bench_ivf.py 2>&1 | grep "ivf query"
NQ=1000 timeout 300 python tools/bench_ivf.py 2>&1 | grep "ivf query"
