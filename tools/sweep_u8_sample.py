#!/usr/bin/env python3
"""uint8 filter pipeline (C3: 10 M x 512-d, k = 10): the exact sample through the streaming kernel (128 queries per pass) up to P passes,
the row-tile kernels beyond (cvtmi_set_tuning "flat_u8_sample_passes").  NQS / PASSES / ROWS env."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, cvt_amd
dev = torch.device("cuda", 0)
n, D, k = int(os.environ.get("ROWS", 10_000_000)), 512, int(os.environ.get("K", 10))
g = torch.Generator(device=dev); g.manual_seed(5)
ix = cvt_amd.FlatIndex(2, D)
for a in range(0, n, 1 << 21):
    ix.add(torch.randint(0, 256, (min(n, a + (1 << 21)) - a, D), generator=g, device=dev, dtype=torch.uint8))
for nq in [int(v) for v in os.environ.get("NQS", "256,384,512,640,768,1000,2048").split(",")]:
    q = torch.randint(0, 256, (nq, D), generator=g, device=dev, dtype=torch.uint8)
    ref = None
    for passes in [int(v) for v in os.environ.get("PASSES", "0,2,4,6,8").split(",")]:
        cvt_amd.set_tuning("flat_u8_sample_passes", passes)
        for _ in range(2):
            d, i = ix.search(q, k)
        torch.cuda.synchronize()
        if ref is None: ref = (d.clone(), i.clone())
        same = bool(torch.equal(d, ref[0]) and torch.equal(i, ref[1]))
        t0 = time.perf_counter()
        for _ in range(5):
            ix.search(q, k)
        torch.cuda.synchronize()
        print("nq=%d sample passes<=%d: %.3f ms, filtered=%s same=%s" % (nq, passes, (time.perf_counter() - t0) / 5 * 1e3, ix.last_search()[0], same), flush=True)
