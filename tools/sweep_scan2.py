#!/usr/bin/env python3
"""Scan kernel time on the bench's own data (SIFT-shaped rows, trained codebooks) over (variant, lazy, share, splits).
ROWS / NQ env; prints one line per configuration: kernel ms (HIP events inside the library) and wall ms per search."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cvt_amd
from cvt_amd import synth
dev = torch.device("cuda", 0)
D, M, K = 128, 16, 256
rows, nq, k = int(os.environ.get("ROWS", 1_000_000)), int(os.environ.get("NQ", 10_000)), int(os.environ.get("K", 100))
tails = os.environ.get("TAILS", "")   # "groups_a:splits_b,..." tried on top of every configuration with splits(req) = 1
cfgs = os.environ.get("CFGS", "3:1:1:0,3:0:0:0,3:1:1:1,3:1:1:2,3:1:1:3,3:1:1:4,3:0:1:2,3:1:0:2,4:1:1:0,4:1:1:2")
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
q = synth.sift_like(nq, D, seed=0xBEEF, device=dev)
idx.set_param("profile", 1)
for _ in range(int(os.environ.get("WARM", 30))):   # clocks and caches settle first: the first configuration of a run used to read 10-15 % slow
    idx.search(q, k)
torch.cuda.synchronize()
if "SEED" in os.environ:
    cvt_amd.set_tuning("scan_seed", int(os.environ["SEED"]))
ref = None
for cfg in cfgs.split(","):
    var, lazy, share, sp = [int(v) for v in cfg.split(":")]
    idx.set_param("scan_variant", var); idx.set_param("scan_lazy", lazy); idx.set_param("scan_share", share); idx.set_param("splits", sp)
    d, i = idx.search(q, k); torch.cuda.synchronize(); idx.last_scan()
    if ref is None:
        ref = (d.clone(), i.clone())
    same = bool(torch.equal(i, ref[1]) and torch.equal(d.view(torch.int32), ref[0].view(torch.int32)))
    t0 = time.perf_counter()
    for _ in range(reps):
        idx.search(q, k)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / reps * 1e3
    s = idx.last_scan()
    print("rows=%d nq=%d variant=%d lazy=%d share=%d splits(req)=%d -> qtile=%d splits=%d: scan %.3f ms, wall %.3f ms, %.0f q/s, alg %.0f GB/s, same=%s" % (
        rows, nq, var, lazy, share, sp, s["qtile"], s["splits"], s["ms"], wall, nq / wall * 1e3, s["code_bytes"] / s["ms"] / 1e6, same), flush=True)
    for t in (tails.split(",") if tails and sp == 1 else []):
        ga, sb = [int(v) for v in t.split(":")]
        idx.set_param("groups_a", ga); idx.set_param("splits_b", sb)
        d, i = idx.search(q, k); torch.cuda.synchronize(); idx.last_scan()
        same = bool(torch.equal(i, ref[1]) and torch.equal(d.view(torch.int32), ref[0].view(torch.int32)))
        for _ in range(reps):
            idx.search(q, k)
        torch.cuda.synchronize()
        print("      tail: first %d groups whole, the rest in %d splits: scan %.3f ms, same=%s" % (ga, sb, idx.last_scan()["ms"], same), flush=True)
    idx.set_param("groups_a", 0); idx.set_param("splits_b", 0)
