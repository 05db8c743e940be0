#!/usr/bin/env python3
"""uint8 flat search, batches between the streaming kernel's 128 queries and the filter pipeline's old lower bound of 256:
flat_variant 0 (dispatch) against 2 (filter pipeline wherever it applies).  ROWS / D / K / NQS env."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, cvt_amd
dev = torch.device("cuda", 0)
n, D, k = int(os.environ.get("ROWS", 10_000_000)), int(os.environ.get("D", 512)), int(os.environ.get("K", 10))
g = torch.Generator(device=dev); g.manual_seed(5)
ix = cvt_amd.FlatIndex(2, D)
for a in range(0, n, 1 << 21):
    ix.add(torch.randint(0, 256, (min(n, a + (1 << 21)) - a, D), generator=g, device=dev, dtype=torch.uint8))
for nq in [int(v) for v in os.environ.get("NQS", "129,160,192,224,255,256").split(",")]:
    q = torch.randint(0, 256, (nq, D), generator=g, device=dev, dtype=torch.uint8)
    ref = None
    for var in (0, 2):
        cvt_amd.set_tuning("flat_variant", var)
        for _ in range(2): d, i = ix.search(q, k)
        torch.cuda.synchronize()
        if ref is None: ref = (d.clone(), i.clone())
        same = bool(torch.equal(d, ref[0]) and torch.equal(i, ref[1]))
        t0 = time.perf_counter()
        for _ in range(5): ix.search(q, k)
        torch.cuda.synchronize()
        print("rows=%d D=%d k=%d nq=%d flat_variant=%d: %.3f ms filtered=%s same=%s" % (n, D, k, nq, var, (time.perf_counter() - t0) / 5 * 1e3, ix.last_search()[0], same), flush=True)
