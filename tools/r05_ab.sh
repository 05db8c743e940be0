#!/bin/bash
# round 5 A/B on the GPU box: HNSW ADC tables in LDS vs read from the table scratch, SQ8 decision filter on / off
set -u
OUT=gpurun_out/r05b; mkdir -p $OUT
export PYTHONUNBUFFERED=1
exec < /dev/null
timeout 600 python -m pytest tests/test_gpu_flat_sq8.py tests/test_gpu_hnsw.py -q -k "sq8 or hnsw" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
for T in "hnsw_adc_tables=0" "hnsw_adc_tables=1"; do
  echo "== $T"
  timeout 400 python bench.py --steps 3 --warmup 1 --only hnsw_c5 --large-rows 0 --cpu-sample 0 --ref-rows 0 --host-api 0 --tune "$T" > $OUT/hnsw_$T.json 2> $OUT/hnsw_$T.err
  python - "$OUT/hnsw_$T.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
h = j["secondary"]["hnsw_c5"]
if "error" in h:
    print(h)
else:
    for m in ("fp32", "adc"):
        for k, v in h[m].items():
            print(m, k, v)
PY
done
for T in "sq8_filter=1" "sq8_filter=0"; do
  echo "== $T"
  timeout 300 python bench.py --steps 3 --warmup 1 --only sq8 --large-rows 0 --cpu-sample 0 --ref-rows 0 --host-api 0 --tune "$T" > $OUT/sq8_$T.json 2> $OUT/sq8_$T.err
  python - "$OUT/sq8_$T.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
s = j["secondary"].get("sq8", j["secondary"])
for k, v in s.items():
    if isinstance(v, dict): print(k, v.get("achieved"), v.get("frac"), v.get("traffic_GBps", ""))
PY
done
