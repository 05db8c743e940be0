#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity sweep over the main entry points (development aid; the fixed cases live in
tests/).  Exits non-zero on the first mismatch and prints the configuration."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch, cvt_amd
from oracle import binding as ob
ob.build()
orc = ob.Oracle()
bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
seed = int(os.environ.get("SEED", 1)); iters = int(os.environ.get("ITERS", 40))
rng = np.random.default_rng(seed)
for it in range(iters):
    M = int(rng.choice([16, 16, 16, 8, 4])); K = int(rng.choice([256, 256, 200, 64, 17])); step = int(rng.choice([8, 4, 2, 16]))
    D = M * step
    n = int(rng.integers(1, 60000)); nq = int(rng.integers(1, 90)); k = int(rng.integers(1, 129))
    scale = float(rng.choice([1.0, 1e-3, 30.0]))
    books = (rng.normal(size=(M, K, step)) * scale).astype(np.float32)
    codes = rng.integers(0, K, size=(n, M), dtype=np.uint8)
    if rng.random() < 0.3:
        codes[rng.integers(0, n, size=max(1, n // 10))] = codes[0]          # many exact ties
    q = (rng.normal(size=(nq, D)) * scale).astype(np.float32)
    if rng.random() < 0.2:
        q[0] = np.inf if rng.random() < 0.5 else np.nan
    idx = cvt_amd.OpqIndex(np.zeros((1, D), np.float32), books)
    idx.add_codes(codes)
    for variant in (3, 4, 1, 0):
        idx.set_param("scan_variant", variant); idx.set_param("splits", int(rng.choice([0, 0, 1, 2, 5])))
        idx.set_param("prerotate", int(rng.integers(0, 2)))
        d, i = idx.search(q, k, rotate=False)
        od, oi = orc.adc_search(q, books, codes, k)
        fin = np.isfinite(od)
        if not (np.array_equal(i[fin], oi[fin]) and np.array_equal(bits(d)[fin], bits(od)[fin])):
            print("MISMATCH adc", dict(it=it, M=M, K=K, D=D, n=n, nq=nq, k=k, scale=scale, variant=variant)); sys.exit(1)
    # flat fp32 + u8
    Df = int(rng.choice([128, 64, 36, 20, 7, 512])); nf = int(rng.integers(1, 30000)); kf = int(rng.integers(1, 129)); nqf = int(rng.integers(1, 300))
    for metric in (0, 1):
        x = rng.normal(size=(nf, Df)).astype(np.float32); qq = rng.normal(size=(nqf, Df)).astype(np.float32)
        x[nf // 2] = x[0]
        fi = cvt_amd.FlatIndex(metric, Df); fi.add(x[: nf // 3 + 1]); fi.add(x[nf // 3 + 1:])
        d, i = fi.search(qq, kf)
        od, _, oi = orc.flat_search(metric, x, qq, kf)
        if not (np.array_equal(i, oi) and np.array_equal(bits(d), bits(od))):
            print("MISMATCH flat", dict(it=it, metric=metric, D=Df, n=nf, nq=nqf, k=kf)); sys.exit(1)
    Du = int(rng.choice([512, 128, 96, 32, 21])); ku = int(rng.choice([1, 5, 10, 16, 24, 40, 80, 100]))
    xu = rng.integers(0, 256, size=(nf, Du), dtype=np.uint8); qu = rng.integers(0, 256, size=(nqf, Du), dtype=np.uint8)
    xu[nf // 2] = xu[0]; qu[0] = xu[0]
    fi = cvt_amd.FlatIndex(2, Du); fi.add(xu)
    d, i = fi.search(qu, ku)
    _, odi, oi = orc.flat_search(2, xu, qu, ku)
    if not (np.array_equal(i, oi) and np.array_equal(d, odi)):
        print("MISMATCH u8", dict(it=it, D=Du, n=nf, nq=nqf, k=ku)); sys.exit(1)
    print("iter %d ok (adc M=%d K=%d n=%d nq=%d k=%d | flat D=%d n=%d nq=%d k=%d | u8 D=%d k=%d)" % (it, M, K, n, nq, k, Df, nf, nqf, kf, Du, ku), flush=True)
print("fuzz: %d iterations, no mismatch" % iters)
