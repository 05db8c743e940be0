#!/bin/bash
cd /root/repo
timeout 200 python -m pytest tests/test_gpu_flat_sq8.py -x -q -m gpu 2>&1 | tail -2
METRIC=2 ROWS=10000000 D=128 NQS=128,256,512,1000 timeout 100 python tools/flat_nq_sweep.py 2>&1 | grep -v amdgpu.ids
