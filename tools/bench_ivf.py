#!/usr/bin/env python3
"""IVF query path (IVFOPQ::Query semantics) at the reference's shape: coarseK = 8192, nk = 3, 1 M entries; NQ frames."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch, cvt_amd
dev = torch.device("cuda", 0)
D, L, nk, n, n_videos = 128, 8192, 3, 1 << 20, 4096
nq = int(os.environ.get("NQ", 10_000))
g = torch.Generator(device=dev); g.manual_seed(11)
rng = np.random.default_rng(3)
cen = torch.randn((L, D), generator=g, device=dev) * 0.08
x = cen[torch.randint(0, L, (n,), generator=g, device=dev)] + 0.03 * torch.randn((n, D), generator=g, device=dev)
q = x[torch.randint(0, n, (nq,), generator=g, device=dev)] + 0.01 * torch.randn((nq, D), generator=g, device=dev)
books = (rng.normal(size=(16, 256, 8)) * 0.03).astype(np.float32)
ix = cvt_amd.OpqIndex(cen.cpu().numpy(), books)
lists, codes = ix.encode(x)
ix.add_codes(codes, lists, torch.randint(0, n_videos, (n,), generator=g, device=dev, dtype=torch.int32))
ix.query_video(q, nk, n_videos, rotate=False); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): ix.query_video(q, nk, n_videos, rotate=False)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 3 * 1e3
print("ivf query: %d frames, nk=%d, %d lists, %d entries: %.3f ms = %.0f frames/s" % (nq, nk, L, n, ms, nq / ms * 1e3))
