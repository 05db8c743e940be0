#!/usr/bin/env python3
"""HNSW search over batch sizes (the reference's own CLI searches one query at a time: hnsw_sifts_retrieval/siftsIndex.cpp): wall ms per
call and queries/s for NQS at EFS, fp32 vectors and OPQ codes, on a graph built by the reference's build on the host.  ROWS / NQS / EFS env."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch, cvt_amd
from oracle import binding as ob
n, D = int(os.environ.get("ROWS", 200_000)), 128
rng = np.random.default_rng(5)
cen = rng.normal(size=(1000, D)).astype(np.float32)
x = cen[rng.integers(0, 1000, n)] + 0.6 * rng.normal(size=(n, D)).astype(np.float32)
x /= np.linalg.norm(x, axis=1, keepdims=True)
nqmax = 10000
q = x[rng.integers(0, n, nqmax)] + 0.15 * rng.normal(size=(nqmax, D)).astype(np.float32)
q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
path = os.path.join(tempfile.gettempdir(), "sweep.hnsw")
rh = ob.RefHnsw()
t0 = time.time(); rh.build(0, x, path, 32, 80); print("reference build: %d rows %.1f s" % (n, time.time() - t0), flush=True)
ix = cvt_amd.HnswIndex(open(path, "rb").read(), 0, D)
qd = torch.from_numpy(q).cuda()
for ef in [int(v) for v in os.environ.get("EFS", "64,1000").split(",")]:
    for nq in [int(v) for v in os.environ.get("NQS", "1,8,64,256,1000,10000").split(",")]:
        qq = qd[:nq].contiguous(); qh = q[:nq].copy()
        k = 5
        for _ in range(3): ix.search(qq, k, ef)
        torch.cuda.synchronize()
        reps = 20 if nq <= 256 else 3
        t0 = time.perf_counter()
        for _ in range(reps):
            ix.search(qq, k, ef); torch.cuda.synchronize()
        td = (time.perf_counter() - t0) / reps
        for _ in range(2): ix.search(qh, k, ef)
        t0 = time.perf_counter()
        for _ in range(reps): ix.search(qh, k, ef)
        th = (time.perf_counter() - t0) / reps
        print("ef=%d nq=%d: device pointers %.3f ms (%.0f queries/s), host pointers %.3f ms" % (ef, nq, td * 1e3, nq / td, th * 1e3), flush=True)
