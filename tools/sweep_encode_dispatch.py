#!/usr/bin/env python3
"""PQ encode: the VALU kernel (encode_variant 1) against the matrix-core filter (2) over row counts -- is the switch at 8192 rows right?
D = 128; M = 16 / 8."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, cvt_amd
dev = torch.device("cuda", 0)
D, K = 128, 256
rng = np.random.default_rng(0)
g = torch.Generator(device=dev); g.manual_seed(1)
for M in (16, 8):
    books = (rng.normal(size=(M, K, D // M)) * 0.3).astype(np.float32)
    ix = cvt_amd.OpqIndex(np.zeros((1, D), np.float32), books)
    for n in (64, 300, 1024, 2048, 4096, 8192, 16384, 65536, 1 << 20):
        x = torch.randn((n, D), generator=g, device=dev)
        t, ref = {}, None
        for var in (0, 1, 2):
            ix.set_param("encode_variant", var)
            for _ in range(3): _, c = ix.encode(x)
            torch.cuda.synchronize()
            if ref is None: ref = c.clone()
            assert torch.equal(c, ref), (M, n, var)
            reps = 20 if n <= 65536 else 5
            t0 = time.perf_counter()
            for _ in range(reps): ix.encode(x)
            torch.cuda.synchronize()
            t[var] = (time.perf_counter() - t0) / reps * 1e3
        best = min((1, 2), key=lambda v: t[v])
        print("M=%d rows=%d: dispatch %.3f ms, VALU %.3f ms, matrix-core filter %.3f ms%s" % (
            M, n, t[0], t[1], t[2], "   <-- dispatch loses %.0f %%" % (100 * (t[0] / t[best] - 1)) if t[0] > 1.07 * t[best] else ""), flush=True)
    ix.close()
