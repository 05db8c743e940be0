#!/usr/bin/env python3
"""OPQ ADC search: the library's own choice of scan kernel / plan (scan_variant 7) against the forced alternatives (3 = adc_scan16q with
the planner's splits, 6 = the persistent grid) over table and batch sizes -- where does the dispatch leave more than 5 % on the table?
Random codes under random codebooks.  ROWS_LIST / NQS / K env."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cvt_amd
from cvt_amd import synth
dev = torch.device("cuda", 0)
D, M, K, k = 128, 16, 256, int(os.environ.get("K", 100))
rng = np.random.default_rng(0)
books = (rng.normal(size=(M, K, D // M)) * 0.05).astype(np.float32)
g = torch.Generator(device=dev); g.manual_seed(1)
for rows in [int(v) for v in os.environ.get("ROWS_LIST", "100000,300000,1000000,3000000,10000000,30000000").split(",")]:
    idx = cvt_amd.OpqIndex(np.zeros((1, D), np.float32), books, R=synth.random_rotation(D))
    idx.add_codes(torch.randint(0, 256, (rows, M), generator=g, device=dev, dtype=torch.uint8))
    for nq in [int(v) for v in os.environ.get("NQS", "1,8,64,128,256,1000,3000,4096,10000").split(",")]:
        if rows * nq > 3e11: continue
        q = torch.randn((nq, D), generator=g, device=dev) * 0.1
        t, ref = {}, None
        idx.set_param("scan_variant", 7); idx.set_param("profile", 1)
        idx.search(q, k); torch.cuda.synchronize()
        ls = idx.last_scan(); took = "variant %s, %s splits" % (ls.get("variant"), ls.get("splits"))
        idx.set_param("profile", 0)
        for var in (7, 3, 6):
            idx.set_param("scan_variant", var)
            for _ in range(2): d, i = idx.search(q, k)
            torch.cuda.synchronize()
            if ref is None: ref = (d.clone(), i.clone())
            assert torch.equal(i, ref[1]) and torch.equal(d.view(torch.int32), ref[0].view(torch.int32)), (rows, nq, var)
            reps = 4
            t0 = time.perf_counter()
            for _ in range(reps): idx.search(q, k)
            torch.cuda.synchronize()
            t[var] = (time.perf_counter() - t0) / reps * 1e3
        best = min(t, key=t.get)
        print("rows=%d nq=%d k=%d: auto %.3f ms (%s), variant 3 %.3f, variant 6 %.3f%s" % (
            rows, nq, k, t[7], took, t[3], t[6], "   <-- auto loses %.0f %% to variant %d" % (100 * (t[7] / t[best] - 1), best) if t[7] > 1.05 * t[best] else ""), flush=True)
    idx.close()
