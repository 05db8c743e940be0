#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02n
timeout 60 python -m pytest tests/test_gpu_hnsw.py -x -q -m gpu -k "golden" > gpurun_out/r02n/pytest.log 2>&1; rc=$?; echo "golden rc=$rc"; tail -2 gpurun_out/r02n/pytest.log
if [ $rc -ne 0 ]; then exit 0; fi
timeout 150 python -m pytest tests/test_gpu_hnsw.py tests/test_gpu_host_cli.py -x -q -m gpu > gpurun_out/r02n/pytest.log 2>&1; echo "all rc=$?"; tail -2 gpurun_out/r02n/pytest.log
timeout 110 python tools/bench_hnsw.py 2>&1 | grep -v amdgpu.ids | grep "ef=" > gpurun_out/r02n/hnsw_dyn.txt; cat gpurun_out/r02n/hnsw_dyn.txt
