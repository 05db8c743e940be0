#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/r02d
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -q -m gpu --durations=15 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -40 $OUT/pytest.log
timeout 900 python bench.py > $OUT/bench_n1.log 2>&1; echo "bench n1 rc=$?"; tail -c 1500 $OUT/bench_n1.log
