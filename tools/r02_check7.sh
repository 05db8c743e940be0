#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/r02g
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONUNBUFFERED=1
rm -rf /tmp/prof_u8
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_u8 -o u8 -- python $REPO/tools/bench_flat_u8_opt.py > $OUT/run.log 2>&1
echo "rc=$?"
f=$(ls /tmp/prof_u8/*/*kernel_stats.csv /tmp/prof_u8/*kernel_stats.csv 2>/dev/null | head -1)
cp $f $OUT/u8_kernel_stats.csv
head -25 $OUT/u8_kernel_stats.csv | cut -c1-220
