#!/bin/bash
cd /root/repo
for k in 1 10 100 128; do K=$k NQS=1000,10000 timeout 300 python tools/opq_nq_sweep.py 2>&1 | grep -v amdgpu.ids; done
for k in 1 64 65 100 128; do K=$k METRIC=2 ROWS=10000000 D=512 NQS=1,64,1000,4096 timeout 300 python tools/flat_nq_sweep.py 2>&1 | grep -v amdgpu.ids; done
for k in 1 100 128; do K=$k METRIC=1 NQS=1,32,1000 timeout 300 python tools/flat_nq_sweep.py 2>&1 | grep -v amdgpu.ids; done
