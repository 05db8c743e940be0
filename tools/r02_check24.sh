#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02n
timeout 900 python -m pytest tests/test_gpu_flat_sq8.py -x -q -m gpu -k "mid_batch or tiny or config3" > gpurun_out/r02n/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02n/pytest.log
METRIC=2 ROWS=10000000 D=512 NQS=8,9,16,32,33,64,65,128,129,256 timeout 300 python tools/flat_nq_sweep.py 2>&1 | grep -v amdgpu.ids
METRIC=2 ROWS=10000000 D=128 NQS=16,17,32,64,128,129 timeout 300 python tools/flat_nq_sweep.py 2>&1 | grep -v amdgpu.ids
