#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02n
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --large-rows 134217728 > gpurun_out/r02n/bench2.json 2> gpurun_out/r02n/bench2.err; echo "rc=$?"; tail -3 gpurun_out/r02n/bench2.err | cut -c1-300; cat gpurun_out/r02n/bench2.json | cut -c1-3000
