#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/r02i
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_opq.py tests/test_gpu_flat_sq8.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ivf
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ivf -o ivf -- python $REPO/tools/bench_ivf.py > $OUT/ivf.log 2>&1
grep "ivf query" $OUT/ivf.log
f=$(ls /tmp/prof_ivf/*/*kernel_stats.csv /tmp/prof_ivf/*kernel_stats.csv 2>/dev/null | head -1)
cp $f $OUT/ivf_kernel_stats.csv; head -12 $OUT/ivf_kernel_stats.csv | cut -c1-160
cd $REPO
python tools/bench_sq8.py 2>&1 | grep -v amdgpu > $OUT/sq8.log; cat $OUT/sq8.log
