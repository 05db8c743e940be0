#!/usr/bin/env python3
"""Randomised parity sweep of the matrix-core PQ encode (encode_variant 2) against the CPU checker and the VALU
kernel: clustered / integer-valued / heavy-tailed / tiny / huge rows, codebooks sampled from the rows (exact hits),
duplicated and nearly duplicated codewords, coarse lists.  Exits non-zero on the first mismatch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch, cvt_amd
from oracle import binding as ob
ob.build()
orc = ob.Oracle()
seed = int(os.environ.get("SEED", 1)); iters = int(os.environ.get("ITERS", 30))
rng = np.random.default_rng(seed)
tot = 0
for it in range(iters):
    M, step = [(16, 8), (8, 16), (8, 8), (4, 16), (16, 8)][int(rng.integers(0, 5))]
    D, K = M * step, 256
    n = int(rng.choice([1, 33, 4000, 9000, 30000, 70000]))
    kind = int(rng.integers(0, 6))
    scale = float(rng.choice([1.0, 1e-3, 1e3, 1e-18, 3e4, 1e12]))
    cen = rng.normal(size=(K * 2, D))
    x = cen[rng.integers(0, K * 2, n)] + float(rng.choice([0.01, 0.3, 1.0])) * rng.normal(size=(n, D))
    if kind == 1: x = np.round(x * 4)                       # small integers: exact ties everywhere
    if kind == 2: x = np.abs(x) ** 3                        # heavy tail
    if kind == 3: x = np.maximum(x, 0); x[:, ::3] = 0       # sparse, SIFT-like zeros
    x = (x * scale).astype(np.float32)
    books = np.ascontiguousarray(np.stack([x[rng.integers(0, n, K), m * step:(m + 1) * step] for m in range(M)])).astype(np.float32)
    if kind == 4: books += (rng.normal(size=books.shape) * 1e-4 * scale).astype(np.float32)
    if rng.random() < 0.5: books[:, 100] = books[:, 7]
    if rng.random() < 0.5: books[:, 101] = np.nextafter(books[:, 8], np.float32(np.inf))
    if rng.random() < 0.2: x[rng.integers(0, n)] = np.nan
    if rng.random() < 0.2: x[rng.integers(0, n), rng.integers(0, D)] = np.inf
    cK = int(rng.choice([1, 1, 1, 7]))
    coarse = np.zeros((1, D), np.float32) if cK == 1 else (rng.normal(size=(cK, D)) * scale).astype(np.float32)
    if cK == 1 and rng.random() < 0.3: coarse[0] = (rng.normal(size=D) * scale).astype(np.float32)
    with np.errstate(all="ignore"):
        ol, oc = orc.pq_encode(x, coarse, books)
    idx = cvt_amd.OpqIndex(coarse, books)
    for v in (2, 1):
        idx.set_param("encode_variant", v)
        l, c = idx.encode(x)
        if not (np.array_equal(c, oc) and np.array_equal(l, ol)):
            bad = np.argwhere(c != oc)
            print("MISMATCH", dict(it=it, seed=seed, M=M, step=step, n=n, kind=kind, scale=scale, cK=cK, variant=v), bad[:5], (l != ol).sum()); sys.exit(1)
    tot += n * M
    print("it %d ok  M=%d step=%d n=%d kind=%d scale=%g coarseK=%d" % (it, M, step, n, kind, scale, cK), flush=True)
print("fuzz_encode: %d iterations, %d (row, m) pairs, no mismatch" % (iters, tot))
