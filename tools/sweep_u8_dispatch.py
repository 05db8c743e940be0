#!/usr/bin/env python3
"""uint8 flat search: the sample + filter pipeline against passes of 128 queries through the streaming kernel, over table and batch
sizes (what the dispatch rule in api.hip flat_route is fitted to).  D / K / ROWS_LIST / NQS env."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, cvt_amd
dev = torch.device("cuda", 0)
D, k = int(os.environ.get("D", 512)), int(os.environ.get("K", 10))
g = torch.Generator(device=dev); g.manual_seed(5)
for n in [int(v) for v in os.environ.get("ROWS_LIST", "1048576,2000000,4000000,10000000").split(",")]:
    ix = cvt_amd.FlatIndex(2, D)
    for a in range(0, n, 1 << 21):
        ix.add(torch.randint(0, 256, (min(n, a + (1 << 21)) - a, D), generator=g, device=dev, dtype=torch.uint8))
    for nq in [int(v) for v in os.environ.get("NQS", "129,256,512,1000,2048,4096").split(",")]:
        q = torch.randint(0, 256, (nq, D), generator=g, device=dev, dtype=torch.uint8)
        ref, t = None, {}
        for name, min_nq in (("stream", 1 << 30), ("filter", 1), ("rule", -1)):
            if min_nq < 0:   # the shipped rule
                cvt_amd.set_tuning("flat_u8_filter_min_nq", 129); cvt_amd.set_tuning("flat_u8_filter_min_rows", 524288); cvt_amd.set_tuning("flat_u8_filter_min_work", 130)
                for _ in range(2): ix.search(q, k)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(4): ix.search(q, k)
                torch.cuda.synchronize()
                t[name] = (time.perf_counter() - t0) / 4 * 1e3
                took = ix.last_search()[0]
                continue
            cvt_amd.set_tuning("flat_u8_filter_min_nq", min_nq); cvt_amd.set_tuning("flat_u8_filter_min_rows", 0); cvt_amd.set_tuning("flat_u8_filter_min_work", 0)
            for _ in range(2): d, i = ix.search(q, k)
            torch.cuda.synchronize()
            if ref is None: ref = (d.clone(), i.clone())
            same = bool(torch.equal(d, ref[0]) and torch.equal(i, ref[1]))
            t0 = time.perf_counter()
            for _ in range(4): ix.search(q, k)
            torch.cuda.synchronize()
            t[name] = (time.perf_counter() - t0) / 4 * 1e3
            assert same, (n, nq, name)
        print("rows=%d D=%d k=%d nq=%d: stream passes %.3f ms, filter pipeline %.3f ms -> %s; the rule takes %s: %.3f ms" % (
            n, D, k, nq, t["stream"], t["filter"], "filter" if t["filter"] < t["stream"] else "stream", "filter" if took else "stream", t["rule"]), flush=True)
    ix.close()
