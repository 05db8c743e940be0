#!/usr/bin/env python3
"""fp32 exhaustive search (brute_force.cpp shape: 128-d inner product / L2, top-100): exact VALU kernels vs the bf16
matrix-core filter (exact search of a leading sample + filter + exact distances of the survivors).  Same results required."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, cvt_amd
from cvt_amd import synth
n, D = int(os.environ.get("ROWS", 1 << 20)), 128
x = synth.sift_like(n, D, device="cuda")
x = x[torch.randperm(n, device="cuda")].contiguous()
for metric, name in ((0, "IP"), (1, "L2")):
    ix = cvt_amd.FlatIndex(metric, D); ix.add(x)
    for nq, k in ((1000, 100), (4096, 10), (10000, 100), (64, 100)):
        q = synth.sift_like(nq, D, seed=0xBEEF, device="cuda")
        out = {}
        for v, vn in ((1, "exact kernels"), (2, "matrix-core filter")):
            cvt_amd.set_tuning("flat_variant", v)
            ix.search(q, k); torch.cuda.synchronize(); t0 = time.perf_counter()
            reps = 3
            for _ in range(reps): out[v] = ix.search(q, k)
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
            used, worst = ix.last_search()
            print("flat %s f32 n=%d d=%d nq=%d k=%d, %-18s: %8.3f ms  %9.0f QPS%s" % (name, n, D, nq, k, vn, ms, nq / ms * 1e3,
                  "  (filter used: %s, largest candidate list %d)" % (used, worst) if v == 2 else ""), flush=True)
        same = torch.equal(out[1][1], out[2][1]) and torch.equal(out[1][0].view(torch.int32), out[2][0].view(torch.int32))
        print("   identical:", same)
        assert same
cvt_amd.set_tuning("flat_variant", 0)
