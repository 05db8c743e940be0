#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02n
timeout 900 python -m pytest tests/test_gpu_flat_sq8.py -x -q -m gpu > gpurun_out/r02n/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02n/pytest.log
for k in 10 64 100 128; do K=$k METRIC=2 ROWS=10000000 D=512 NQS=129,200,256,1000,4096 timeout 300 python tools/flat_nq_sweep.py 2>&1 | grep -v amdgpu.ids; done
K=10 METRIC=2 ROWS=10000000 D=128 NQS=129,500,1000,4096 timeout 300 python tools/flat_nq_sweep.py 2>&1 | grep -v amdgpu.ids
