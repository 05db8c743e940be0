#!/usr/bin/env python3
"""OPQ ADC search for M = 16 / 8 / 4 sub-quantisers (D = 128) over batch sizes: only M = 16 has the skewed 15-bit scan."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, cvt_amd
from cvt_amd import synth
dev = torch.device("cuda", 0)
D, K, k = 128, 256, int(os.environ.get("K", 100))
rows = int(os.environ.get("ROWS", 1_000_000))
rng = np.random.default_rng(0)
g = torch.Generator(device=dev); g.manual_seed(1)
for M in [int(v) for v in os.environ.get("MS", "16,8,4").split(",")]:
    books = (rng.normal(size=(M, K, D // M)) * 0.05).astype(np.float32)
    idx = cvt_amd.OpqIndex(np.zeros((1, D), np.float32), books, R=synth.random_rotation(D))
    idx.add_codes(torch.randint(0, 256, (rows, M), generator=g, device=dev, dtype=torch.uint8))
    for nq in [int(v) for v in os.environ.get("NQS", "1,8,64,1000,10000").split(",")]:
        q = torch.randn((nq, D), generator=g, device=dev) * 0.1
        for _ in range(2): idx.search(q, k)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); reps = 4
        for _ in range(reps): idx.search(q, k)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        print("M=%d rows=%d nq=%d k=%d: %.3f ms, %.0f queries/s, %.1f G code bytes x queries / s" % (M, rows, nq, k, ms, nq / ms * 1e3, rows * M * nq / ms / 1e6), flush=True)
    idx.close()
