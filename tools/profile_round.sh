#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats + separate PMC passes, summaries only (the raw .db/.csv stay
# in /tmp) -> gpurun_out/<tag>/, from where they are copied to profiles/.
#   1. bench.py as the driver runs it (kernel-trace --stats): every kernel of the N = 1 line incl. sift1b and the secondary configs
#   2. bench.py, headline only, once per PMC group (counters never share a run with a trace domain other than kernel-trace)
#   3. the scan at an HBM-resident shard (128 M rows = the per-GPU shard of SIFT-1B): kernel-trace + FETCH_SIZE / WRITE_SIZE passes
set -u
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
HEAD="python $REPO/bench.py --cpu-sample 0 --recall-sample 0 --large-rows 0 --secondary 0 --host-api 0 --ref-rows 0"
# 1a. the headline alone: every launch of the scan kernel in this trace is a C2 launch, so its average is comparable with roofline.kernel_ms
rm -rf /tmp/prof_stats
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o $TAG -- $HEAD --steps 10 --warmup 3 > $OUT/stats_run.log 2>&1
echo "stats rc=$?"
cp $(ls /tmp/prof_stats/*/*kernel_stats.csv /tmp/prof_stats/*kernel_stats.csv 2>/dev/null | head -1) $OUT/${TAG}_kernel_stats.csv
python $REPO/tools/pmc_summary.py /tmp/prof_stats > $OUT/${TAG}_kernel_trace_summary.json
grep '"metric"' $OUT/stats_run.log > $OUT/${TAG}_bench_under_rocprof.json
# 1b. the whole N = 1 line as the driver runs it (sift1b + secondary configs: their kernels, and the scan kernel's 2^30-row launches)
rm -rf /tmp/prof_full
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_full -o $TAG -- python $REPO/bench.py --cpu-sample 0 --recall-sample 0 --steps 10 --warmup 3 > $OUT/stats_full_run.log 2>&1
echo "full stats rc=$?"
cp $(ls /tmp/prof_full/*/*kernel_stats.csv /tmp/prof_full/*kernel_stats.csv 2>/dev/null | head -1) $OUT/${TAG}_kernel_stats_full_line.csv
grep '"metric"' $OUT/stats_full_run.log > $OUT/${TAG}_bench_full_line_under_rocprof.json
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVES" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  rm -rf /tmp/prof_pmc
  timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d /tmp/prof_pmc -o $TAG -- $HEAD --steps 3 --warmup 1 > /tmp/pmc_run.log 2>&1
  echo "pmc [$pmc] rc=$?"
  python $REPO/tools/pmc_summary.py /tmp/prof_pmc > $OUT/pmc_$i.json
done
# 3. HBM-resident shard: 128 M rows, 2048 queries
BIG="python $REPO/bench.py --cpu-sample 0 --recall-sample 0 --large-rows 0 --secondary 0 --host-api 0 --ref-rows 0 --rows 134217728 --nq 2048 --steps 3 --warmup 1"
j=0
for pmc in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  j=$((j+1))
  rm -rf /tmp/prof_big
  timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d /tmp/prof_big -o big -- $BIG > /tmp/big_run.log 2>&1
  echo "big pmc [$pmc] rc=$?"
  python $REPO/tools/pmc_summary.py /tmp/prof_big adc_scan > $OUT/big_pmc_$j.json
  grep '"metric"' /tmp/big_run.log > $OUT/${TAG}_bench_128m_under_rocprof.json
done
# SKIP_FLAT=1 leaves sections 4 and 5 out (rounds that did not change the flat kernels keep the previous summaries)
if [ -z "${SKIP_FLAT:-}" ]; then
# 4. config 3 kernels (10 M x 512-d uint8): the filter kernel at nq = 4096 and the streaming kernel at nq = 1 / 64, one PMC group per run
u=0
for pmc in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"; do
  u=$((u+1))
  for nq in 4096 64 1; do
    rm -rf /tmp/prof_u8
    NQ=$nq VARIANT=0 REPS=3 timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d /tmp/prof_u8 -o u8 -- python $REPO/tools/u8_one.py > /tmp/u8_run.log 2>&1
    echo "u8 nq=$nq pmc [$pmc] rc=$?"
    python $REPO/tools/pmc_summary.py /tmp/prof_u8 flat_u8 > $OUT/u8_pmc_${u}_$nq.json
  done
done
python - <<PY
import json, glob, os
out, tag = "$OUT", "$TAG"
u8 = {}
for f in sorted(glob.glob(os.path.join(out, "u8_pmc_*_*.json"))):
    nq = f.rsplit("_", 1)[1].split(".")[0]
    d = json.load(open(f))
    for k, v in d.get("counters", {}).items():
        if "gfilter" in k or "mstream" in k or "stream_finish" in k:
            e = u8.setdefault("nq=" + nq, {}).setdefault(k, {})
            e.update(v)
            e.setdefault("kernel_trace", {}).update(d.get("kernel_trace", {}).get(k, {}))
json.dump({"what": "10 M x 512-d uint8 rows, top-10, default dispatch; counters are sums over the run's dispatches of that kernel (3 searches + warm-up); FETCH_SIZE is in KB and needs the x2 gfx950 correction", "by_batch": u8},
          open(os.path.join(out, tag + "_pmc_flat_u8.json"), "w"), indent=1)
PY
# 5. fp32 flat search (1 M x 128-d, top-100): the stream kernels at nq = 1 / 64 / 1000, one PMC group per run
v=0
for pmc in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"; do
  v=$((v+1))
  for nq in 1000 64 1; do
    rm -rf /tmp/prof_f32
    NQ=$nq REPS=3 timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d /tmp/prof_f32 -o f32 -- python $REPO/tools/f32_one.py > /tmp/f32_run.log 2>&1
    echo "f32 nq=$nq pmc [$pmc] rc=$?"
    python $REPO/tools/pmc_summary.py /tmp/prof_f32 flat_f32 > $OUT/f32_pmc_${v}_$nq.json
  done
done
python - <<PY
import json, glob, os
out, tag = "$OUT", "$TAG"
f32 = {}
for f in sorted(glob.glob(os.path.join(out, "f32_pmc_*_*.json"))):
    nq = f.rsplit("_", 1)[1].split(".")[0]
    d = json.load(open(f))
    for k, v in d.get("counters", {}).items():
        e = f32.setdefault("nq=" + nq, {}).setdefault(k, {})
        e.update(v)
        e.setdefault("kernel_trace", {}).update(d.get("kernel_trace", {}).get(k, {}))
json.dump({"what": "1 M x 128-d fp32 rows (inner product), top-100, default dispatch; counters are per-dispatch means of that kernel; FETCH_SIZE is in KB and needs the x2 gfx950 correction", "by_batch": f32},
          open(os.path.join(out, tag + "_pmc_flat_f32.json"), "w"), indent=1)
PY
fi
python - <<PY
import json, glob, os
out, tag = "$OUT", "$TAG"
def merge(pattern):
    ctr, trace = {}, {}
    for f in sorted(glob.glob(os.path.join(out, pattern))):
        d = json.load(open(f))
        for k, v in d.get("counters", {}).items():
            ctr.setdefault(k, {}).update(v)
        trace.update(d.get("kernel_trace", {}))
    return ctr, trace
ctr, _ = merge("pmc_*.json")
json.dump(ctr, open(os.path.join(out, tag + "_pmc_by_kernel.json"), "w"), indent=1)
note = "FETCH_SIZE x1024 x2 (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE x1024; separate --pmc passes"
def traffic(ctr, trace, name, **shape):
    scan = [k for k in ctr if "adc_scan" in k]
    if not scan: return
    c = ctr[scan[0]]
    fetch, write = c.get("FETCH_SIZE", 0) * 1024 * 2, c.get("WRITE_SIZE", 0) * 1024
    t = trace.get(scan[0], {})
    json.dump({"kernel": scan[0], "hbm_bytes_per_launch": int(fetch + write), "fetch_bytes_x2_corrected": int(fetch), "write_bytes": int(write),
               "kernel_avg_us_under_pmc": t.get("avg_us"), "l2_hit_rate": (c.get("TCC_HIT_sum", 0) / max(1.0, c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0))),
               "note": note, **shape}, open(os.path.join(out, name), "w"), indent=1)
c1, t1 = merge("pmc_*.json"); traffic(c1, t1, tag + "_scan_traffic.json", rows=1000000, nq=10000, k=100)
c2, t2 = merge("big_pmc_*.json"); traffic(c2, t2, tag + "_scan_traffic_128m.json", rows=134217728, nq=2048, k=100)
PY
cd $REPO
{ exec < /dev/null; python tools/bench_kernels.py; python tools/bench_sq8.py; METRIC=0 python tools/flat_nq_sweep.py; python tools/bench_flat_u8_opt.py; python tools/bench_train.py; python tools/bench_pca.py; python tools/bench_encode.py; python tools/bench_assign.py; python tools/bench_flat_filter.py; python tools/bench_ivf.py; NQ=9 python tools/bench_ivf.py; python tools/bench_hnsw.py; METRIC=2 ROWS=10000000 D=512 python tools/flat_nq_sweep.py; METRIC=2 ROWS=10000000 D=128 python tools/flat_nq_sweep.py; METRIC=1 python tools/flat_nq_sweep.py; python tools/opq_nq_sweep.py; } 2>&1 | grep -v amdgpu > $OUT/${TAG}_other_kernels.txt
rm -f $OUT/pmc_*.json $OUT/big_pmc_*.json $OUT/u8_pmc_*.json $OUT/f32_pmc_*.json
ls -la $OUT
