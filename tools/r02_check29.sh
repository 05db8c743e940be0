#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02n
timeout 1500 python -m pytest tests/test_gpu_hnsw.py -x -q -m gpu > gpurun_out/r02n/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02n/pytest.log
timeout 900 python tools/bench_hnsw.py 2>&1 | grep -v amdgpu.ids | tail -20
