#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02n
timeout 600 python -m pytest tests/test_gpu_flat_sq8.py -x -q -m gpu -k "tiny_batch or seeded or config3" > gpurun_out/r02n/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02n/pytest.log
timeout 300 python tools/u8_tiny.py > gpurun_out/r02n/tiny512.log 2>&1; echo "rc=$?"; cat gpurun_out/r02n/tiny512.log
D=128 timeout 300 python tools/u8_tiny.py > gpurun_out/r02n/tiny128.log 2>&1; echo "rc=$?"; cat gpurun_out/r02n/tiny128.log
