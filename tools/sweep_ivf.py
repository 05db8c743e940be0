#!/usr/bin/env python3
"""IVF query path (IVFOPQ::Query semantics: coarse top-nk of 8192 lists, residual tables, per-video minima) over frames per call,
sub-quantisers and index size: ms per call and frames/s.  The reference calls it with the frames of ONE query video (a handful)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch, cvt_amd
dev = torch.device("cuda", 0)
D, L, nk = 128, 8192, 3
g = torch.Generator(device=dev); g.manual_seed(11)
rng = np.random.default_rng(3)
cen = torch.randn((L, D), generator=g, device=dev) * 0.08
for n, n_videos in ((1 << 20, 4096), (8 << 20, 32768)):
    x = cen[torch.randint(0, L, (n,), generator=g, device=dev)] + 0.03 * torch.randn((n, D), generator=g, device=dev)
    for M in (16, 8):
        books = (rng.normal(size=(M, 256, D // M)) * 0.03).astype(np.float32)
        ix = cvt_amd.OpqIndex(cen.cpu().numpy(), books)
        for a in range(0, n, 1 << 20):
            lists, codes = ix.encode(x[a:a + (1 << 20)])
            ix.add_codes(codes, lists, torch.randint(0, n_videos, (codes.shape[0],), generator=g, device=dev, dtype=torch.int32))
        for nq in (1, 9, 100, 1000, 10000):
            q = x[torch.randint(0, n, (nq,), generator=g, device=dev)] + 0.01 * torch.randn((nq, D), generator=g, device=dev)
            for _ in range(3): ix.query_video(q, nk, n_videos, rotate=False)
            torch.cuda.synchronize()
            reps = 20 if nq <= 100 else 3
            t0 = time.perf_counter()
            for _ in range(reps):
                ix.query_video(q, nk, n_videos, rotate=False); torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            print("entries=%d videos=%d M=%d frames=%d: %.3f ms per call, %.0f frames/s" % (n, n_videos, M, nq, ms, nq / ms * 1e3), flush=True)
        ix.close()
