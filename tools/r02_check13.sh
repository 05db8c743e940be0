#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/r02k
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 300 python tools/scan_timing.py > $OUT/scan_timing.log 2>&1; echo "timing rc=$?"; grep -v amdgpu $OUT/scan_timing.log
