#!/bin/bash
cd /root/repo
bash tools/r02_check25.sh
for m in 3 129; do
python - <<PY 2>&1 | grep -v amdgpu.ids
import sys; sys.path.insert(0, '.')
import os, torch, cvt_amd
cvt_amd.set_tuning("flat_u8_mstream_min", $m)
os.environ.update(METRIC="2", ROWS="10000000", D=os.environ.get("DD", "512"), NQS="1,2,3,4,5,8")
print("mstream_min", $m)
exec(open("tools/flat_nq_sweep.py").read())
PY
done
