#!/bin/bash
cd /root/repo
for cap in 512 1024 256; do
  sed -i "s/^constexpr int HN_LCAP = [0-9]*;/constexpr int HN_LCAP = $cap;/" cvt_amd/csrc/hnsw.hip
  make -C cvt_amd/csrc -j8 2>&1 | grep -E " error" | head -3
  echo "HN_LCAP=$cap"
  timeout 600 python tools/bench_hnsw.py 2>&1 | grep -v amdgpu.ids | grep "ef="
done
