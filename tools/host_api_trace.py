#!/usr/bin/env python3
"""A few host-pointer searches of the headline batch, to be run under
   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -- python tools/host_api_trace.py
(timeline of the pieces: which copies run under which scans).  PINNED=1: page-locked arrays; TUNE="name=value,...\""""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cvt_amd
from cvt_amd import synth
dev = torch.device("cuda", 0)
D, M, K = 128, 16, 256
rows, nq, k = int(os.environ.get("ROWS", 1_000_000)), int(os.environ.get("NQ", 10_000)), int(os.environ.get("K", 100))
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
for kv in [v for v in os.environ.get("TUNE", "").split(",") if v]:
    name, value = kv.split("=")
    cvt_amd.set_tuning(name, float(value))
if os.environ.get("PINNED", "0") == "1":
    qh = cvt_amd.pinned_empty((nq, D), np.float32); qh[:] = q.cpu().numpy()
    out = (cvt_amd.pinned_empty((nq, k), np.float32), cvt_amd.pinned_empty((nq, k), np.int64))
else:
    qh = q.cpu().numpy()
    out = (np.zeros((nq, k), np.float32), np.zeros((nq, k), np.int64))
for _ in range(int(os.environ.get("REPS", 6))):
    idx.search(qh, k, rotate=True, out=out)
torch.cuda.synchronize()
print("done")
