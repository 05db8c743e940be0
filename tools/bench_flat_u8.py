#!/usr/bin/env python3
"""uint8 L2 flat search (config C3 shape: 512-d) -- queries/s and algorithmic row traffic."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cvt_amd.capi as capi
if os.environ.get("CVTMI_LIB"): capi.LIB_PATH = os.environ["CVTMI_LIB"]
import torch, cvt_amd
dev = torch.device("cuda", 0)
n, D = int(os.environ.get("ROWS", 2_000_000)), int(os.environ.get("D", 512))
g = torch.Generator(device=dev); g.manual_seed(5)
db = torch.randint(0, 256, (n, D), generator=g, device=dev, dtype=torch.uint8)
ix = cvt_amd.FlatIndex(2, D); ix.add(db)
qs = torch.randint(0, 256, (4096, D), generator=g, device=dev, dtype=torch.uint8)
for nq, k in [(int(a), int(b)) for a, b in (x.split(":") for x in os.environ.get("CASES", "1000:10,1000:100,256:10,64:10,4096:10").split(","))]:
    q = qs[:nq].contiguous()
    res = {}
    for v, name in ((1, "row-tile kernels"), (2, "sample+filter+sort")):   # 2 = exact sample, i8 matrix-core threshold filter, sort (opt-in)
        cvt_amd.set_tuning("flat_variant", v)
        ix.search(q, k); torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps): res[v] = ix.search(q, k)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        print("flat L2 u8 %d-d n=%d nq=%d k=%d %-16s: %.3f ms  %.0f QPS  %.1f T int-MAC/s%s" % (D, n, nq, k, name, ms, nq / ms * 1e3, n * D * nq / ms / 1e9,
              "  (filter pipeline: %s)" % ix.last_search()[0] if v == 2 else ""), flush=True)
    assert torch.equal(res[2][0], res[1][0]) and torch.equal(res[2][1], res[1][1])
cvt_amd.set_tuning("flat_variant", 0)
