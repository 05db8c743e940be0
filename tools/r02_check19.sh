#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02n
timeout 900 python -m pytest tests/test_gpu_flat_sq8.py -x -q -m gpu > gpurun_out/r02n/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02n/pytest.log
timeout 600 python tools/bench_flat_u8_opt.py 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/u8_nq_sweep.py 2>&1 | grep -v amdgpu.ids
