#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02n
timeout 900 python -m pytest tests/test_gpu_opq.py -x -q -m gpu -k "rotate_encode or golden_rotate or encode" > gpurun_out/r02n/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02n/pytest.log
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, numpy as np, torch
sys.path.insert(0, '.')
import cvt_amd
from cvt_amd import synth
import bench
dev = torch.device("cuda", 0)
D, M, K, n = 128, 16, 256, 1 << 20
zero = np.zeros((1, D), np.float32)
tmp = cvt_amd.OpqIndex(zero, np.zeros((M, K, D // M), np.float32), perm=synth.random_permutation(D, seed=5))
books = synth.train_books(tmp.rotate(synth.sift_like(100_000, D, seed=0xC0FFEE, device=dev)), M, K, iters=4)
tmp.close()
x = synth.sift_like(n, D, seed=0xC0FFEE, device=dev)
ixp = cvt_amd.OpqIndex(zero, books, perm=synth.random_permutation(D, seed=5))
for name, f in (("permute+encode", lambda: ixp.encode(ixp.rotate(x))), ("rotate_encode (gather)", lambda: ixp.rotate_encode(x)), ("encode only", lambda: ixp.encode(x))):
    ms = bench._ev_ms(torch, f)
    print("%-24s %.3f ms  %.3f G rows/s" % (name, ms, n / ms / 1e6))
a = ixp.encode(ixp.rotate(x)); b = ixp.rotate_encode(x)
print("same codes:", bool(torch.equal(a[1], b[1])), "same lists:", bool(torch.equal(a[0], b[0])))
PY
