#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02n
timeout 900 python -m pytest tests/test_gpu_flat_sq8.py tests/test_gpu_host_cli.py -x -q -m gpu > gpurun_out/r02n/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02n/pytest.log
METRIC=2 ROWS=10000000 D=512 NQS=1,2,4,8,16,32,64,65,128,129,200,255,256,257,384,512,1000,4096 timeout 300 python tools/flat_nq_sweep.py 2>&1 | grep -v amdgpu.ids
