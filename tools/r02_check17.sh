#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02n
timeout 900 python -m pytest tests/test_gpu_flat_sq8.py -x -q -m gpu -k "filter_pipeline or config3 or mfma_query_tiles" > gpurun_out/r02n/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02n/pytest.log
timeout 600 python tools/bench_flat_u8_opt.py 2>&1 | grep -v amdgpu.ids
D=128 timeout 600 python tools/bench_flat_u8_opt.py 2>&1 | grep -v amdgpu.ids
