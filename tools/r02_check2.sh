#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/r02b
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_sharded.py -x -q > $OUT/pytest_sharded.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_sharded.log
timeout 300 python tools/scan_timing.py > $OUT/scan_timing.log 2>&1; echo "timing rc=$?"; grep -v amdgpu $OUT/scan_timing.log
timeout 300 python tools/scan_trace.py > $OUT/scan_trace.log 2>&1; echo "trace rc=$?"; grep -v amdgpu $OUT/scan_trace.log
