#!/usr/bin/env python3
"""uint8 flat search over ROWS x D, default dispatch, batch sizes NQS: ms per search and TB/s of rows."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, cvt_amd
dev = torch.device("cuda", 0)
n, D = int(os.environ.get("ROWS", 10_000_000)), int(os.environ.get("D", 512))
g = torch.Generator(device=dev); g.manual_seed(5)
ix = cvt_amd.FlatIndex(2, D)
for a in range(0, n, 1 << 21):
    ix.add(torch.randint(0, 256, (min(n, a + (1 << 21)) - a, D), generator=g, device=dev, dtype=torch.uint8))
k = int(os.environ.get("K", 10))
for nq in [int(v) for v in os.environ.get("NQS", "1,2,4,5,8,16,32,64,128,256,512").split(",")]:
    q = torch.randint(0, 256, (nq, D), generator=g, device=dev, dtype=torch.uint8)
    for _ in range(2):
        ix.search(q, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        ix.search(q, k)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"rows={n} D={D} nq={nq} k={k}: {ms:.3f} ms, {n * D / ms / 1e9:.2f} TB/s of rows, {2.0 * nq * n * D / ms / 1e12:.1f} Tint-op/s", flush=True)
