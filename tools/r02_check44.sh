#!/bin/bash
cd /root/repo
timeout 150 python -m pytest tests/test_gpu_flat_sq8.py -x -q -m gpu 2>&1 | tail -2
METRIC=2 ROWS=10000000 D=128 NQS=1,32,64,128 timeout 60 python tools/flat_nq_sweep.py 2>&1 | grep -v amdgpu.ids
METRIC=2 ROWS=10000000 D=512 NQS=1,32,64,128 timeout 60 python tools/flat_nq_sweep.py 2>&1 | grep -v amdgpu.ids
SEED=3 CASES=60 timeout 100 python tools/fuzz_flat_u8.py 2>&1 | grep -v amdgpu.ids | grep -c " ok"
