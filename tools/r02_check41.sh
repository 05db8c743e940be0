#!/bin/bash
REPO=/root/repo
OUT=$REPO/gpurun_out/r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_full
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_full -o r02 -- python $REPO/bench.py --cpu-sample 0 --recall-sample 0 --steps 10 --warmup 3 > $OUT/stats_full_run.log 2>&1
echo "full stats rc=$?"
cp $(ls /tmp/prof_full/*/*kernel_stats.csv /tmp/prof_full/*kernel_stats.csv 2>/dev/null | head -1) $OUT/r02_kernel_stats_full_line.csv
grep '"metric"' $OUT/stats_full_run.log > $OUT/r02_bench_full_line_under_rocprof.json
head -12 $OUT/r02_kernel_stats_full_line.csv | cut -c1-160
