#!/bin/bash
# The round's closing check on a GPU box (through gpurun): every -m gpu test, smoke(), the driver's bench command, and a 2-rank run
# of the N > 1 bench path on this one GPU through the host transport -- each under its own time-out, nothing reads stdin.
#   usage: gpurun --timeout 1500 -- bash tools/round_close.sh r03
set -u
TAG=${1:-close}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export PYTHONUNBUFFERED=1
exec < /dev/null
timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" $OUT/pytest.log | tail -3
timeout 180 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; head -c 300 $OUT/bench.json; echo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 \
    --backend host --nq 1000 --steps 2 --warmup 1 --large-rows 67108864 > $OUT/bench_2ranks_1gpu.json 2> $OUT/bench_2ranks_1gpu.err
echo "2-rank bench (launcher) rc=$?"; tail -1 $OUT/bench_2ranks_1gpu.json | head -c 300; echo
# the same with NO launcher (bench.py starts its own ranks) and the RCCL request a one-GPU box cannot meet: the line must still appear
timeout 600 python bench.py --gpus 2 --nq 1000 --steps 2 --warmup 1 --large-rows 67108864 > $OUT/bench_2ranks_selflaunch.json 2> $OUT/bench_2ranks_selflaunch.err
echo "2-rank bench (self-launch) rc=$?"; tail -1 $OUT/bench_2ranks_selflaunch.json | head -c 300; echo
