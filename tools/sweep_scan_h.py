#!/usr/bin/env python3
"""adc_scan16h (scan_variant 6) against adc_scan16q (3) on the bench's data: kernel ms (HIP events inside the library: tables +
scan for 6, scan for 3), wall ms per search, identical results.  ROWS / NQS / K env; CFGS = variant:balance:min_rows:splits[:share_hist],..."""
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
nqs = [int(v) for v in os.environ.get("NQS", "10000,1000,256,8,1").split(",")]
cfgs = os.environ.get("CFGS", "3:0:0:0,6:0:0:0,6:1:0:0")
reps = int(os.environ.get("REPS", 10))
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
    _, codes = idx.encode(idx.rotate(synth.sift_like(b - a, D, seed=0xC0FFEE, row_begin=a, device=dev)))
    idx.add_codes(codes)
qs = synth.sift_like(max(nqs), D, seed=0xBEEF, device=dev)
idx.set_param("profile", 1)
idx.set_param("scan_share", int(os.environ.get("SHARE", 1)))
cvt_amd.set_tuning("scanh_tail", int(os.environ.get("TAIL", 1)))
for _ in range(int(os.environ.get("WARM", 20))):
    idx.search(qs[:min(len(qs), 10000)], k)
torch.cuda.synchronize()
for nq in nqs:
    q = qs[:nq].contiguous()
    ref = None
    for cfg in cfgs.split(","):
        f = [int(v) for v in cfg.split(":")]
        var, bal, minr, sp = f[:4]
        sh = f[4] if len(f) > 4 else 1
        cvt_amd.set_tuning("scanh_fix", f[5] if len(f) > 5 else 170000)
        cvt_amd.set_tuning("scanh_share_hist", sh)
        idx.set_param("scan_variant", var); idx.set_param("splits", sp)
        cvt_amd.set_tuning("scanh_balance", bal)
        cvt_amd.set_tuning("scanh_min_rows", minr if minr else 16384)
        d, i = idx.search(q, k); torch.cuda.synchronize(); idx.last_scan()
        if ref is None:
            ref = (d.clone(), i.clone())
        same = bool(torch.equal(i, ref[1]) and torch.equal(d.view(torch.int32), ref[0].view(torch.int32)))
        for _ in range(3):
            idx.search(q, k)
        torch.cuda.synchronize(); idx.last_scan()
        t0 = time.perf_counter()
        for _ in range(reps):
            idx.search(q, k)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / reps * 1e3
        s = idx.last_scan()
        print("rows=%d nq=%d k=%d variant=%d balance=%d min_rows=%d splits(req)=%d share_hist=%d fix=%d -> lists=%d: kernel %.4f ms, wall %.4f ms, %.0f q/s, same=%s" % (
            rows, nq, k, var, bal, minr, sp, sh, f[5] if len(f) > 5 else 170000, s["splits"], s["ms"], wall, nq / wall * 1e3, same), flush=True)
