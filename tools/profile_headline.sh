#!/bin/bash
# The headline part of tools/profile_round.sh alone (sections 1a and the two HBM-traffic PMC passes): re-taken when only the scan kernel
# changed late in a round.   usage: gpurun --timeout 900 -- bash tools/profile_headline.sh r05
set -u
TAG=${1:-r05}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/${TAG}_headline
mkdir -p $OUT
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
HEAD="python $REPO/bench.py --cpu-sample 0 --recall-sample 0 --large-rows 0 --secondary 0 --host-api 0 --ref-rows 0"
rm -rf /tmp/prof_stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o $TAG -- $HEAD --steps 10 --warmup 3 > $OUT/stats_run.log 2>&1
echo "stats rc=$?"
cp $(ls /tmp/prof_stats/*/*kernel_stats.csv /tmp/prof_stats/*kernel_stats.csv 2>/dev/null | head -1) $OUT/${TAG}_kernel_stats.csv
python $REPO/tools/pmc_summary.py /tmp/prof_stats > $OUT/${TAG}_kernel_trace_summary.json
grep '"metric"' $OUT/stats_run.log > $OUT/${TAG}_bench_under_rocprof.json
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf /tmp/prof_pmc
  timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d /tmp/prof_pmc -o $TAG -- $HEAD --steps 3 --warmup 1 > /tmp/pmc_run.log 2>&1
  echo "pmc [$pmc] rc=$?"
  python $REPO/tools/pmc_summary.py /tmp/prof_pmc > $OUT/pmc_$i.json
done
head -5 $OUT/${TAG}_kernel_stats.csv
