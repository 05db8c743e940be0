#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd $REPO
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_opq.py tests/test_gpu_host_cli.py -q -x 2>&1 | tail -3
CFGS=3:1:1:0,3:1:1:0,3:0:0:0,3:1:1:0 timeout 300 python tools/sweep_scan2.py 2>&1 | grep -v amdgpu
