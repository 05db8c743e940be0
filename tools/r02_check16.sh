#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02n
timeout 600 python -m pytest tests/test_gpu_flat_sq8.py -x -q -m gpu -k "tiny_batch or config3" > gpurun_out/r02n/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02n/pytest.log
for b in 1024 2048 4096 8192; do
BLOCKS=$b VARIANTS=0 timeout 300 python tools/u8_tiny.py 2>&1 | grep -v amdgpu.ids | sed "s/^/blocks=$b /"
done
BLOCKS=2048 D=128 timeout 300 python tools/u8_tiny.py 2>&1 | grep -v amdgpu.ids
K=100 timeout 300 python tools/u8_tiny.py 2>&1 | grep -v amdgpu.ids
