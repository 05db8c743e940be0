#!/usr/bin/env python3
"""Nearest-centroid assignment (coarse argmin of IVFOPQ::Add; k-means assignment pass): the VALU kernels (reference
chain for every centroid) against the bf16 matrix-core filter + exact resolution.  Same lists / centroids required."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, cvt_amd
from cvt_amd import synth
dev = "cuda"
D, M, K = 128, 16, 256
n = int(os.environ.get("ROWS", 1 << 18))
x = synth.sift_like(n, D, device=dev)
books = (np.random.default_rng(0).normal(size=(M, K, D // M)) * 0.05).astype(np.float32)
for cK in (1024, 8192):
    coarse = x[torch.randperm(n, device=dev)[:cK]].cpu().numpy() + 0.01 * np.random.default_rng(1).normal(size=(cK, D)).astype(np.float32)
    idx = cvt_amd.OpqIndex(coarse.astype(np.float32), books)
    out = {}
    for v, name in ((1, "VALU chain"), (2, "matrix-core filter")):
        cvt_amd.set_tuning("assign_variant", v)
        idx.encode(x); torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 2 if v == 1 else 5
        for _ in range(reps): out[v] = idx.encode(x)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
        print("coarse assign + PQ encode, coarseK=%d n=%d, %-18s: %.2f ms  %.1f M rows/s" % (cK, n, name, ms, n / ms / 1e3), flush=True)
    print("   lists identical:", torch.equal(out[1][0], out[2][0]), " codes identical:", torch.equal(out[1][1], out[2][1]))
xs = x[:200_000].contiguous()
for k, it in ((256, 10), (8192, 5)):
    res = {}
    for v, name in ((1, "VALU chain"), (2, "matrix-core filter")):
        cvt_amd.set_tuning("assign_variant", v)
        t0 = time.perf_counter(); c, a, done = cvt_amd.kmeans(xs, k, it, 1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res[v] = (c, a)
        print("kmeans n=%d d=128 k=%d, %-18s: %d iterations, %.1f ms per iteration" % (xs.shape[0], k, name, done, dt / max(1, done) * 1e3), flush=True)
    print("   centroids identical:", torch.equal(res[1][0].view(torch.int32), res[2][0].view(torch.int32)), " assignments identical:", torch.equal(res[1][1], res[2][1]))
cvt_amd.set_tuning("assign_variant", 0)
