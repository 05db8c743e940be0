# development: secondary.hnsw_c5 of bench.py alone (TUNE="name=value,..." NODES=...), after the HNSW GPU tests
[ -n "$SKIP_TESTS" ] || timeout 300 python -m pytest tests/test_gpu_hnsw.py -x -q 2>&1 | tail -2
for T in ${TUNES:-""}; do
echo "== tune: $T"
timeout 500 python bench.py --steps 3 --warmup 1 --only hnsw_c5 --large-rows 0 --cpu-sample 0 --ref-rows 0 --hnsw-nodes ${NODES:-1000000} --tune "$T" > gpurun_out/b_hnsw.json 2> gpurun_out/b_hnsw.err < /dev/null
python - <<PY
import json
j=json.loads(open("gpurun_out/b_hnsw.json").read().strip().splitlines()[-1])
h=j["secondary"]["hnsw_c5"]
if "error" in h: print(h)
else:
    print("build", h.get("graph_build_s_host"), h.get("graph_build_threads"))
    for m in ("fp32","adc"):
        for k,v in h[m].items(): print(m,k,v)
PY
done
