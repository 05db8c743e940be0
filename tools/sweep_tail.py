#!/usr/bin/env python3
"""C2's last, partly filled round (1250 query groups on 512 workgroup slots): adc_scan16q's two-region plan forced through
cvtmi_set_tuning("scan_tail_splits", S) -- the groups past the last full round are cut into S row splits -- against whole groups, and the
batch sizes that fill their rounds exactly (4096, 8192 queries) as the yardstick of what a perfectly packed tail would give."""
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
R = synth.random_rotation(D, seed=7)
tmp = cvt_amd.OpqIndex(zero, np.zeros((M, K, D // M), np.float32), R=R)
books = synth.train_books(tmp.rotate(synth.sift_like(100_000, D, seed=0xC0FFEE, device=dev)), M, K, iters=4)
tmp.close()
idx = cvt_amd.OpqIndex(zero, books, R=R)
idx.reserve(rows)
step = synth.CHUNK * 4
for a in range(0, rows, step):
    b = min(rows, a + step)
    _, codes = idx.rotate_encode(synth.sift_like(b - a, D, seed=0xC0FFEE, row_begin=a, device=dev))
    idx.add_codes(codes)
qall = synth.sift_like(10_000, D, seed=0xBEEF, device=dev)
idx.set_param("profile", 1)
for _ in range(30):
    idx.search(qall, k)
torch.cuda.synchronize()
for nq in [int(v) for v in os.environ.get("NQS", "10000,8192,4096,9000,6000").split(",")]:
    q = qall[:nq].contiguous()
    ref = None
    for tail in [int(v) for v in os.environ.get("TAILS", "0,2,3,4,6").split(",")]:
        cvt_amd.set_tuning("scan_tail_splits", tail)
        d, i = idx.search(q, k); torch.cuda.synchronize(); idx.last_scan()
        if ref is None:
            ref = (d.clone(), i.clone())
        same = bool(torch.equal(i, ref[1]) and torch.equal(d.view(torch.int32), ref[0].view(torch.int32)))
        t0 = time.perf_counter()
        for _ in range(20):
            idx.search(q, k)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 20 * 1e3
        s = idx.last_scan()
        print("nq=%d tail_splits=%d: scan %.3f ms, wall %.3f ms, %.3f M q/s, splits=%d, same=%s" % (nq, tail, s["ms"], wall, nq / wall / 1e3, s["splits"], same), flush=True)
cvt_amd.set_tuning("scan_tail_splits", 0)
