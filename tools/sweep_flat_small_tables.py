#!/usr/bin/env python3
"""Flat search on small tables: the dispatch (flat_variant 0) against the exact kernels only (1), fp32 128-d and uint8 512-d --
are the lower bounds of the streaming kernels (32 768 / 262 144 rows) where they should be?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, cvt_amd
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5)
def run(ix, q, k):
    t = {}
    ref = None
    for var in (0, 1):
        cvt_amd.set_tuning("flat_variant", var)
        for _ in range(3): d, i = ix.search(q, k)
        torch.cuda.synchronize()
        if ref is None: ref = i.clone()
        assert torch.equal(i, ref), (var,)
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps): ix.search(q, k)
        torch.cuda.synchronize()
        t[var] = (time.perf_counter() - t0) / reps * 1e3
    cvt_amd.set_tuning("flat_variant", 0)
    return t
if os.environ.get("U8_MIN_ROWS"): cvt_amd.set_tuning("flat_u8_mstream_min_rows", int(os.environ["U8_MIN_ROWS"]))
CASES = ((1, 128, 100, (4096, 8192, 16384, 32768, 65536, 131072, 300000)), (2, 512, 10, (8192, 20000, 32768, 65536, 131072, 200000, 262144, 400000)),
         (2, 128, 10, (8192, 32768, 100000, 262144)), (2, 256, 100, (8192, 32768, 100000, 262144)))
if os.environ.get("ONLY_U8"): CASES = CASES[1:]
for metric, D, k, sizes in CASES:
    for n in sizes:
        ix = cvt_amd.FlatIndex(metric, D)
        ix.add(torch.randn((n, D), generator=g, device=dev) if metric != 2 else torch.randint(0, 256, (n, D), generator=g, device=dev, dtype=torch.uint8))
        for nq in (1, 16, 100, 1000):
            q = torch.randn((nq, D), generator=g, device=dev) if metric != 2 else torch.randint(0, 256, (nq, D), generator=g, device=dev, dtype=torch.uint8)
            t = run(ix, q, k)
            print("metric=%d rows=%d D=%d nq=%d: dispatch %.3f ms, exact kernels %.3f ms%s" % (
                metric, n, D, nq, t[0], t[1], "   <-- dispatch loses %.0f %%" % (100 * (t[0] / t[1] - 1)) if t[0] > 1.07 * t[1] else ""), flush=True)
        ix.close()
