#!/usr/bin/env python3
"""HBM-resident shard sizes (SURVEY.md 8d C2'/C4): ADC scan on 64 M / 128 M rows of M=16 codes per GPU.
Codes are generated directly (uniform random bytes): the scan's cost does not depend on their values."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, cvt_amd
from cvt_amd import synth
dev = torch.device("cuda", 0)
D, M, K = 128, 16, 256
rng = np.random.default_rng(0)
books = (rng.normal(size=(M, K, D // M)) * 0.05).astype(np.float32)
R = synth.random_rotation(D)
for rows in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16000000,64000000,128000000").split(",")]:
    idx = cvt_amd.OpqIndex(np.zeros((1, D), np.float32), books, R=R)
    idx.reserve(rows)
    g = torch.Generator(device=dev); g.manual_seed(1)
    for a in range(0, rows, 1 << 24):
        b = min(rows, a + (1 << 24))
        idx.add_codes(torch.randint(0, 256, (b - a, M), generator=g, device=dev, dtype=torch.uint8))
    idx.set_param("profile", 1)
    for nq in (8, 64, 1024, 10000):
        if nq * rows > 3e12:
            continue
        q = synth.sift_like(nq, D, seed=0xBEEF, device=dev)
        idx.search(q, 100); torch.cuda.synchronize(); idx.last_scan()
        t0 = time.perf_counter(); reps = 2
        for _ in range(reps):
            idx.search(q, 100)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / reps * 1e3
        s = idx.last_scan()
        print("rows=%dM nq=%d: scan %.2f ms wall %.2f ms  QPS %.0f  qtile %d splits %d  algorithmic %.0f GB/s (%.2f of 8 TB/s)  per-query %.1f GB/s" % (
            rows // 1000000, nq, s["ms"], wall, nq / wall * 1e3, s["qtile"], s["splits"], s["code_bytes"] / s["ms"] / 1e6,
            s["code_bytes"] / s["ms"] / 1e6 / 8000, nq * rows * M / s["ms"] / 1e6), flush=True)
    idx.close(); del idx
    torch.cuda.empty_cache()
