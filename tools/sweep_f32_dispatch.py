#!/usr/bin/env python3
"""fp32 flat search: the one-stream kernels (flat_variant 0) against the older sample + matrix-core filter pipeline (flat_variant 2) over
table and batch sizes.  D / K / METRIC / ROWS_LIST / NQS env."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, cvt_amd
dev = torch.device("cuda", 0)
D, k, metric = int(os.environ.get("D", 128)), int(os.environ.get("K", 100)), int(os.environ.get("METRIC", 1))
g = torch.Generator(device=dev); g.manual_seed(5)
for n in [int(v) for v in os.environ.get("ROWS_LIST", "1000000,4000000,10000000").split(",")]:
    ix = cvt_amd.FlatIndex(metric, D)
    for a in range(0, n, 1 << 21):
        ix.add(torch.randn((min(n, a + (1 << 21)) - a, D), generator=g, device=dev))
    for nq in [int(v) for v in os.environ.get("NQS", "16,64,128,256,512,1000,4096").split(",")]:
        q = torch.randn((nq, D), generator=g, device=dev)
        ref, t, how = None, {}, {}
        for var in (0, 2):
            cvt_amd.set_tuning("flat_variant", var)
            for _ in range(2): d, i = ix.search(q, k)
            torch.cuda.synchronize()
            if ref is None: ref = (d.clone(), i.clone())
            assert torch.equal(i, ref[1]) and torch.equal(d.view(torch.int32), ref[0].view(torch.int32)), (n, nq, var)
            t0 = time.perf_counter()
            for _ in range(4): ix.search(q, k)
            torch.cuda.synchronize()
            t[var] = (time.perf_counter() - t0) / 4 * 1e3; how[var] = ix.last_search()[0]
        cvt_amd.set_tuning("flat_variant", 0)
        print("rows=%d D=%d k=%d metric=%d nq=%d: dispatch %.3f ms (route %d), filter pipeline %.3f ms (route %d)%s" % (
            n, D, k, metric, nq, t[0], how[0], t[2], how[2], "   <-- pipeline faster" if t[2] < 0.95 * t[0] else ""), flush=True)
    ix.close()
