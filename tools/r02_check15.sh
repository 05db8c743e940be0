#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02n
export TMPDIR=/tmp
rm -rf /tmp/prof_tiny
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tiny -o tiny --output-format csv -- python tools/u8_tiny.py > gpurun_out/r02n/tiny_prof.log 2>&1; echo "rc=$?"
f=$(find /tmp/prof_tiny -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r02n/tiny_kernel_stats.csv
head -20 "$f" | cut -c1-200
