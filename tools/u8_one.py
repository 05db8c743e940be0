#!/usr/bin/env python3
"""One configuration of the uint8 flat search for profiling: VARIANT (flat_variant) / GF (flat_u8_gfilter) / NQ / K env."""
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
nq, k = int(os.environ.get("NQ", 4096)), int(os.environ.get("K", 10))
q = torch.randint(0, 256, (nq, D), generator=g, device=dev, dtype=torch.uint8)
cvt_amd.set_tuning("flat_variant", int(os.environ.get("VARIANT", 2))); cvt_amd.set_tuning("flat_u8_gfilter", int(os.environ.get("GF", 1)))
for _ in range(int(os.environ.get("REPS", 3))):
    ix.search(q, k)
torch.cuda.synchronize()
