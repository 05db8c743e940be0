#!/usr/bin/env python3
"""OPQ search (rotate + tables + scan + top-k) on the bench's data over batch sizes NQS: wall ms per search, plan the library chose."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cvt_amd
from cvt_amd import synth
dev = torch.device("cuda", 0)
D, M, K = 128, 16, 256
rows, k = int(os.environ.get("ROWS", 1_000_000)), int(os.environ.get("K", 100))
zero = np.zeros((1, D), np.float32)
R = synth.random_rotation(D, seed=7)
tmp = cvt_amd.OpqIndex(zero, np.zeros((M, K, D // M), np.float32), R=R)
books = synth.train_books(tmp.rotate(synth.sift_like(100_000, D, seed=0xC0FFEE, device=dev)), M, K, iters=4)
tmp.close()
idx = cvt_amd.OpqIndex(zero, books, R=R)
idx.reserve(rows)
step = synth.CHUNK * 4
for a in range(0, rows, step):
    b = min(rows, a + step)
    _, codes = idx.encode(idx.rotate(synth.sift_like(b - a, D, seed=0xC0FFEE, row_begin=a, device=dev)))
    idx.add_codes(codes)
qs = synth.sift_like(20_000, D, seed=0xBEEF, device=dev)
idx.set_param("profile", 1)
cvt_amd.set_tuning("scans_dbg", int(os.environ.get("SDBG", 0)))
for nq in [int(v) for v in os.environ.get("NQS", "1,2,4,8,16,32,64,128,256,512,1000,2000,3000,4096,5000,6000,8000,10000,12000,16000,20000").split(",")]:
    q = qs[:nq].contiguous()
    for _ in range(2):
        idx.search(q, k)
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        idx.search(q, k)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / reps * 1e3
    s = idx.last_scan()
    print("rows=%d nq=%d k=%d -> variant=%s qtile=%d splits=%d: scan %.3f ms, wall %.3f ms, %.0f q/s, %.2f us per query, alg %.0f GB/s" % (
        rows, nq, k, s.get("variant"), s["qtile"], s["splits"], s["ms"], wall, nq / wall * 1e3, wall * 1e3 / nq, s["code_bytes"] / s["ms"] / 1e6), flush=True)
