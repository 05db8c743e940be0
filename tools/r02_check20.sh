#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02n
timeout 900 python -m pytest tests/test_gpu_flat_sq8.py -x -q -m gpu -k "mfma_query_tiles or tiny" > gpurun_out/r02n/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02n/pytest.log
NQS=4,5,6,7,8 timeout 300 python tools/u8_nq_sweep.py 2>&1 | grep -v amdgpu.ids
