#!/usr/bin/env python3
"""Codebook training time on the GPU (reference shape: coarseK = 8192, M = 16, K = 256, 128-d)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, cvt_amd
from cvt_amd import synth
n = int(os.environ.get("ROWS", 200_000))
x = synth.sift_like(n, 128, seed=3, device="cuda")
for k, it in ((256, 10), (8192, 5)):
    t0 = time.perf_counter(); c, a, done = cvt_amd.kmeans(x, k, it, 1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("kmeans n=%d d=128 k=%d: %d iterations in %.3f s (%.1f ms per iteration)" % (n, k, done, dt, dt / max(1, done) * 1e3), flush=True)
t0 = time.perf_counter(); coarse, books = cvt_amd.opq_train(x, 1, 16, 256, 10, 1); torch.cuda.synchronize()
print("opq_train n=%d coarseK=1 M=16 K=256, 10 iterations each: %.3f s" % (n, time.perf_counter() - t0))
