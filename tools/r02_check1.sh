#!/bin/bash
# first GPU pass of round 2: full -m gpu suite, N=1 bench, N=2 bench on one GPU (gloo transport through the library path)
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/r02a
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
timeout 900 python bench.py > $OUT/bench_n1.log 2>&1; echo "bench n1 rc=$?"; tail -c 3000 $OUT/bench_n1.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --backend gloo --large-rows 134217728 --steps 3 --warmup 1 > $OUT/bench_n2_gloo.log 2>&1; echo "bench n2 rc=$?"; tail -c 2500 $OUT/bench_n2_gloo.log
