cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl -o x -- python $R/bench.py --steps 5 --warmup 2 --secondary 0 --large-rows 0 --cpu-sample 0 --ref-rows 0 --host-api 0 > /tmp/pl.log 2>&1 < /dev/null
f=$(find /tmp/pl -name "*kernel_stats.csv" | head -1)
grep -E "lut|rotate_gemm|adc_scan16q" "$f" < /dev/null | sed -E "s/\(.*\)\"/\"/" | cut -c1-150
grep '"metric"' /tmp/pl.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms'])"
