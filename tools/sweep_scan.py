#!/usr/bin/env python3
"""Build the bench index once and time the scan kernel (HIP events inside the library) over a grid of
(qtile, splits).  Development aid, prints one line per configuration."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cvt_amd
from cvt_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1_000_000)
ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--k", type=int, default=100)
ap.add_argument("--M", type=int, default=16)
ap.add_argument("--qtiles", default="1,2,4")
ap.add_argument("--splits", default="1,8,16,32,64")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--variants", default="1,0")
ap.add_argument("--tail", type=int, default=1)
ap.add_argument("--prerot", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda", 0)
D, M, K = 128, a.M, 256
R = synth.random_rotation(D)
zero = np.zeros((1, D), np.float32)
rng = np.random.default_rng(0)
books = (rng.normal(size=(M, K, D // M)) * 0.05).astype(np.float32)
idx = cvt_amd.OpqIndex(zero, books, R=R)
g = torch.Generator(device=dev); g.manual_seed(1)
codes = torch.randint(0, 256, (a.rows, M), generator=g, device=dev, dtype=torch.uint8)
idx.add_codes(codes)
q = synth.sift_like(a.nq, D, seed=0xBEEF, device=dev)
idx.set_param("profile", 1)
idx.set_param("tail_split", a.tail)
idx.set_param("prerotate", a.prerot)
for var in [int(x) for x in a.variants.split(",")]:
  for qt in [int(x) for x in a.qtiles.split(",")]:
    for sp in [int(x) for x in a.splits.split(",")]:
        idx.set_param("scan_variant", var); idx.set_param("qtile", qt); idx.set_param("splits", sp)
        try:
            idx.search(q, a.k); torch.cuda.synchronize(); idx.last_scan()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                idx.search(q, a.k)
            torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / a.reps * 1e3
            s = idx.last_scan()
            print("variant=%d(req) " % var + "qtile=%d splits=%-4d scan=%.3f ms wall=%.3f ms  q-lookups/s=%.2fT  alg=%.0f GB/s  QPS=%.0f" % (
                s["qtile"], s["splits"], s["ms"], wall, a.nq * a.rows * M / s["ms"] / 1e9, s["code_bytes"] / s["ms"] / 1e6,
                a.nq / wall * 1e3), flush=True)
        except Exception as e:
            print("qtile=%d splits=%d failed: %s" % (qt, sp, e), flush=True)
