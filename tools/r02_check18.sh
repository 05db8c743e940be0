#!/bin/bash
cd /root/repo
timeout 600 python tools/bench_flat_u8_opt.py 2>&1 | grep -v amdgpu.ids
