#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd $REPO
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_flat_sq8.py -q -x 2>&1 | tail -3
python tools/bench_sq8.py 2>&1 | grep -v amdgpu
