#!/usr/bin/env python3
"""small host-pointer OPQ searches (the reference's call pattern: 1-9 query frames per call): wall time per call for NQS, with the
polling stream wait and zero-copy staging on / off (cvtmi_set_tuning "host_spin_us", "opq_small_zero_copy"; CASES=spin:zc,...); ROWS / K env."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cvt_amd
from cvt_amd import synth
dev = torch.device("cuda", 0)
D, M, K = 128, 16, 256
rows, k = int(os.environ.get("ROWS", 1_000_000)), int(os.environ.get("K", 100))
zero = np.zeros((1, D), np.float32)
rng = np.random.default_rng(1)
books = rng.normal(size=(M, K, D // M)).astype(np.float32)
idx = cvt_amd.OpqIndex(zero, books, R=synth.random_rotation(D, seed=7))
idx.add_codes(torch.randint(0, 256, (rows, M), dtype=torch.uint8, device=dev))
for nq in [int(v) for v in os.environ.get("NQS", "1,8,32,128,1000").split(",")]:
    qh = rng.normal(size=(nq, D)).astype(np.float32)
    out = (np.zeros((nq, k), np.float32), np.zeros((nq, k), np.int64))
    ref = None
    for spin, zc in [tuple(int(x) for x in v.split(":")) for v in os.environ.get("CASES", "200:0,200:1,0:1,200:0,200:1").split(",")]:
        cvt_amd.set_tuning("host_spin_us", spin); cvt_amd.set_tuning("opq_small_zero_copy", zc)
        for _ in range(20): idx.search(qh, k, out=out)
        if ref is None: ref = (out[0].copy(), out[1].copy())
        same = bool(np.array_equal(ref[1], out[1]) and np.array_equal(ref[0].view(np.uint32), out[0].view(np.uint32)))
        t0 = time.perf_counter(); reps = 300
        for _ in range(reps): idx.search(qh, k, out=out)
        print("nq=%d spin=%d us zero_copy=%d: %.1f us per call, same=%s" % (nq, spin, zc, (time.perf_counter() - t0) / reps * 1e6, same), flush=True)
