#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/r02f
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 300 python tools/bench_flat_u8_opt.py > $OUT/flat_u8_gf.log 2>&1; echo "gf rc=$?"; grep -v amdgpu $OUT/flat_u8_gf.log | tail -24
