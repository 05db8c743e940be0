#!/bin/bash
# GPU box: the driver's bench command (N = 1) + a 2-rank run of the N > 1 path on this one GPU through the host transport
set -u
OUT=gpurun_out/${1:-bench}; mkdir -p $OUT; export PYTHONUNBUFFERED=1
timeout ${T1:-900} python bench.py ${ARGS:-} < /dev/null > $OUT/bench.log 2> $OUT/bench.err
echo "rc=$?"; grep '"metric"' $OUT/bench.log > $OUT/bench.json; tail -3 $OUT/bench.err
if [ "${TWO:-0}" = "1" ]; then
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --backend host --steps 2 --warmup 1 --large-rows 4000000 --large-nq 512 --nq 512 < /dev/null > $OUT/bench2.log 2>&1
  echo "two-rank rc=$?"; grep '"metric"' $OUT/bench2.log | cut -c1-600; tail -3 $OUT/bench2.log | cut -c1-300
fi
python - <<'PY'
import json,sys,os
p=os.path.join("gpurun_out", os.environ.get("TAG","bench"), "bench.json")
try:
    d=json.loads(open(p).read().strip().splitlines()[-1])
except Exception as e:
    print("no json", e); sys.exit(0)
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "lds_frac", d["roofline"]["lds_frac"], "hbm_meas", d["roofline"].get("hbm_frac_measured"))
print("host_pointer", d.get("host_pointer_api"))
print("cpu", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline_all_cores",{}).get("value"))
s=d.get("secondary",{})
if "error" in s: print("secondary error", s["error"])
for k in ("rotation","encode","rotate_encode","reorder_encode","sq8"):
    print(k, json.dumps(s.get(k))[:400])
f=s.get("flat_f32",{})
print("flat_f32", json.dumps(f)[:1500])
c=s.get("flat_u8_c3",{})
print("flat_u8_c3", json.dumps(c)[:1500])
print("sift1b", json.dumps(d.get("sift1b"))[:400])
print("hnsw", json.dumps(s.get("hnsw_c5"))[:800])
PY
