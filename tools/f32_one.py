#!/usr/bin/env python3
"""one fp32 flat-search shape for a profiler run: ROWS x D rows (METRIC 0 = IP, 1 = L2), NQ queries, top-K, REPS timed searches"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, cvt_amd
dev = torch.device("cuda", 0)
n, D, metric = int(os.environ.get("ROWS", 1_000_000)), int(os.environ.get("D", 128)), int(os.environ.get("METRIC", 0))
nq, k, reps = int(os.environ.get("NQ", 1)), int(os.environ.get("K", 100)), int(os.environ.get("REPS", 3))
g = torch.Generator(device=dev); g.manual_seed(5)
ix = cvt_amd.FlatIndex(metric, D)
ix.add(torch.randn((n, D), generator=g, device=dev))
q = torch.randn((nq, D), generator=g, device=dev)
ix.search(q, k); torch.cuda.synchronize()
for _ in range(reps):
    ix.search(q, k)
torch.cuda.synchronize()
print("done", n, D, nq, k, ix.last_search())
