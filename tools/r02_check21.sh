#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02n
timeout 900 python -m pytest tests/test_gpu_flat_sq8.py -x -q -m gpu -k "f32" > gpurun_out/r02n/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02n/pytest.log
for m in 0 1; do METRIC=$m NQS=8,12,16,24,32,48,64 timeout 300 python tools/flat_nq_sweep.py 2>&1 | grep -v amdgpu.ids; done
METRIC=1 ROWS=10000000 NQS=8,16,32 timeout 300 python tools/flat_nq_sweep.py 2>&1 | grep -v amdgpu.ids
