#!/usr/bin/env python3
"""fp32 flat search over row widths (the stream takes 32 / 64 / 96 / 128 / 192 / 256-d): ms per search and the fraction of the time one
pass over the rows takes at 6.4 TB/s."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, cvt_amd
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5)
k = 10
for D in (32, 48, 64, 100, 128, 160, 192, 256, 384, 512, 1024):
    n = min(2_000_000, (1 << 30) // (4 * D))
    for metric in (1,):
        ix = cvt_amd.FlatIndex(metric, D)
        for a in range(0, n, 1 << 19):
            ix.add(torch.randn((min(n, a + (1 << 19)) - a, D), generator=g, device=dev))
        for nq in (1, 16, 100, 1000):
            q = torch.randn((nq, D), generator=g, device=dev)
            for _ in range(3): ix.search(q, k)
            torch.cuda.synchronize()
            reps = 10
            t0 = time.perf_counter()
            for _ in range(reps): ix.search(q, k)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            floor_ms = n * D * 4 / 6.4e12 * 1e3
            print("D=%d rows=%d nq=%d: %.3f ms (route %d); one pass over the rows at 6.4 TB/s = %.3f ms -> x%.1f" % (D, n, nq, ms, ix.last_search()[0], floor_ms, ms / floor_ms), flush=True)
        ix.close()
