#!/bin/bash
# the round's closing check on a GPU box: every -m gpu test, the default bench line, smoke -- each under its own timeout
cd /root/repo
mkdir -p gpurun_out/r02full
timeout 420 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r02full/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/r02full/pytest.log | tail -3
timeout 300 python bench.py > gpurun_out/r02full/bench.json 2> gpurun_out/r02full/bench.err; echo "bench rc=$?"; head -c 400 gpurun_out/r02full/bench.json; echo
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
{ timeout 100 python tools/bench_sq8.py; timeout 120 python tools/bench_hnsw.py; } 2>&1 | grep -v amdgpu > gpurun_out/r02full/sq8_hnsw.txt; tail -4 gpurun_out/r02full/sq8_hnsw.txt
