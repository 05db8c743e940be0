#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/r02c
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_opq.py -x -q > $OUT/pytest_opq.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_opq.log
timeout 600 python tools/sweep_scan2.py > $OUT/sweep_1m.log 2>&1; echo "sweep rc=$?"; grep -v amdgpu $OUT/sweep_1m.log
ROWS=134217728 NQ=2048 REPS=2 CFGS=3:0:0:0,3:1:1:0,3:1:0:0,3:1:1:8,3:1:1:16 timeout 600 python tools/sweep_scan2.py > $OUT/sweep_128m.log 2>&1; echo "sweep2 rc=$?"; grep -v amdgpu $OUT/sweep_128m.log
