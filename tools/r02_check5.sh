#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/r02e
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_ivf.py tests/test_gpu_hnsw.py -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest.log
timeout 600 python tools/bench_flat_u8_opt.py > $OUT/flat_u8_opt.log 2>&1; echo "opt rc=$?"; grep -v amdgpu $OUT/flat_u8_opt.log
