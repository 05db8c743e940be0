#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02n
timeout 900 python -m pytest tests/test_gpu_flat_sq8.py -x -q -m gpu -k "tiny or config3 or mfma_query" > gpurun_out/r02n/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02n/pytest.log
METRIC=2 ROWS=10000000 D=512 NQS=4,5,8,9,12,16,17,32 timeout 300 python tools/flat_nq_sweep.py 2>&1 | grep -v amdgpu.ids
cat > /tmp/v1.py <<'PY'
import os, sys
sys.path.insert(0, '.')
import cvt_amd
cvt_amd.set_tuning("flat_variant", 1)
exec(open("tools/flat_nq_sweep.py").read())
PY
METRIC=2 ROWS=10000000 D=512 NQS=8,16 timeout 300 python /tmp/v1.py 2>&1 | grep -v amdgpu.ids
METRIC=2 ROWS=10000000 D=128 NQS=4,8,16,17 timeout 300 python tools/flat_nq_sweep.py 2>&1 | grep -v amdgpu.ids
