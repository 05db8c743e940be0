#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02full
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r02full/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02full/pytest.log
timeout 900 python bench.py > gpurun_out/r02full/bench.json 2> gpurun_out/r02full/bench.err; echo "bench rc=$?"; tail -c 6000 gpurun_out/r02full/bench.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
