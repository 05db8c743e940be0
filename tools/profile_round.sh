#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats + separate PMC passes for bench.py,
# summaries only (the raw .db/.csv stay in /tmp) -> gpurun_out/<tag>/
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --cpu-sample 0 --recall-sample 0 --large-rows 0"
rm -rf /tmp/prof_stats
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o $TAG -- $B --steps 10 --warmup 3 > $OUT/stats_run.log 2>&1
echo "stats rc=$?"
cp /tmp/prof_stats/*kernel_stats.csv $OUT/ 2>/dev/null
python $REPO/tools/pmc_summary.py /tmp/prof_stats > $OUT/kernel_trace_summary.json
grep '"metric"' $OUT/stats_run.log > $OUT/bench_under_rocprof.json
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVES" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  rm -rf /tmp/prof_pmc
  timeout 900 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d /tmp/prof_pmc -o $TAG -- $B --steps 3 --warmup 1 > /tmp/pmc_run.log 2>&1
  echo "pmc [$pmc] rc=$?"
  python $REPO/tools/pmc_summary.py /tmp/prof_pmc > $OUT/pmc_$i.json
done
python - <<PY
import json, glob, os
out = "$OUT"
ctr = {}
for f in sorted(glob.glob(os.path.join(out, "pmc_*.json"))):
    d = json.load(open(f)).get("counters", {})
    for k, v in d.items():
        ctr.setdefault(k, {}).update(v)
json.dump(ctr, open(os.path.join(out, "pmc_by_kernel.json"), "w"), indent=1)
scan = [k for k in ctr if "adc_scan" in k]
if scan:
    c = ctr[scan[0]]
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE (KB) reports 1/2 of a wide coalesced stream on gfx950 -> x2; WRITE_SIZE uncalibrated
    fetch = c.get("FETCH_SIZE", 0) * 1024 * 2
    write = c.get("WRITE_SIZE", 0) * 1024
    json.dump({"kernel": scan[0], "hbm_bytes_per_launch": int(fetch + write), "fetch_bytes_x2_corrected": int(fetch),
               "write_bytes": int(write), "note": "FETCH_SIZE x1024 x2 (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE x1024; separate --pmc passes"},
              open(os.path.join(out, "scan_traffic.json"), "w"), indent=1)
PY
# one line per secondary kernel / sibling path (not profiled, just timed)
cd $REPO
{ python tools/bench_kernels.py; python tools/bench_sq8.py; python tools/bench_flat_f32.py;
  ROWS=10000000 CASES=1000:10,4096:10,1:10 python tools/bench_flat_u8.py; python tools/bench_train.py; python tools/bench_hnsw.py; python tools/bench_pca.py; python tools/bench_encode.py; python tools/bench_assign.py; python tools/bench_flat_filter.py; } 2>&1 | grep -v amdgpu > $OUT/other_kernels.txt
ls -la $OUT
