#!/usr/bin/env python3
"""uint8 flat search, 1, 2, 4 queries over ROWS x D: streaming matrix-core path (flat_variant 0) vs the row-per-lane kernels (flat_variant 1)."""
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
variants = tuple(int(v) for v in os.environ.get("VARIANTS", "0,1").split(","))
for nq in (1, 2, 4):
    q = torch.randint(0, 256, (nq, D), generator=g, device=dev, dtype=torch.uint8)
    res = {}
    for v in variants:
        cvt_amd.set_tuning("flat_variant", v)
        for _ in range(3):
            out = ix.search(q, k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            out = ix.search(q, k)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res[v] = out
        print(f"rows={n} D={D} nq={nq} k={k} variant={v}: {ms:.3f} ms, {n * D / ms / 1e9:.2f} TB/s of rows", flush=True)
    if len(res) == 2:
        same = all(torch.equal(torch.as_tensor(a), torch.as_tensor(b)) for a, b in zip(res[0], res[1]))
        print("  same:", same, flush=True)
cvt_amd.set_tuning("flat_variant", 0)
