#!/usr/bin/env python3
"""The reference's index build is IVFOPQ::Add of one video at a time -- a few hundred frames (opq/src/IVFOPQ.cpp:135-163): wall time of
encode (+ coarse assignment over 8192 lists) and append for small row counts, device and host pointers."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, cvt_amd
dev = torch.device("cuda", 0)
D, M, K = 128, 16, 256
rng = np.random.default_rng(0)
g = torch.Generator(device=dev); g.manual_seed(1)
books = (rng.normal(size=(M, K, D // M)) * 0.3).astype(np.float32)
for L in (1, 8192):
    coarse = np.zeros((1, D), np.float32) if L == 1 else rng.normal(size=(L, D)).astype(np.float32)
    ix = cvt_amd.OpqIndex(coarse, books, perm=np.arange(D, dtype=np.int32)[::-1].copy())
    ix.reserve(4_000_000)
    for n in (1, 64, 300, 1024, 4096, 8192, 65536):
        x = torch.randn((n, D), generator=g, device=dev); xh = x.cpu().numpy()
        vid = np.zeros(n, np.int32)
        for _ in range(3):
            lists, codes = ix.encode(ix.rotate(x))
        torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            lists, codes = ix.encode(ix.rotate(x)); torch.cuda.synchronize()
        td = (time.perf_counter() - t0) / reps * 1e3
        for _ in range(2):
            xr = ix.rotate(xh); lh, ch = ix.encode(xr)
        t0 = time.perf_counter()
        for _ in range(reps):
            xr = ix.rotate(xh); lh, ch = ix.encode(xr); ix.add_codes(ch, lh if L > 1 else None, vid)
        th = (time.perf_counter() - t0) / reps * 1e3
        print("lists=%d rows=%d: rotate + encode on device %.3f ms; rotate + encode + append through host pointers %.3f ms" % (L, n, td, th), flush=True)
    ix.close()
