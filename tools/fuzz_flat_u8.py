#!/usr/bin/env python3
"""Randomised parity of the uint8 flat search: default dispatch (streaming matrix-core kernel / filter pipeline) against the row-tile kernels
(flat_variant 1) over random row counts, widths, batch sizes, k and data shapes (uniform bytes, few distinct values, planted duplicates,
appended chunks).  CASES env (default 60); prints the first mismatch and exits 1."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch, cvt_amd
dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(os.environ.get("SEED", 1)))
cases = int(os.environ.get("CASES", 60))
bad = 0
for c in range(cases):
    D = int(rng.choice([128, 256, 512]))
    n = int(rng.integers(262_144, 700_000))
    nq = int(rng.choice([1, 2, 3, 5, 8, 9, 16, 17, 31, 33, 64, 65, 100, 128, 129, 200, 257, 300]))
    k = int(rng.choice([1, 2, 7, 10, 33, 64, 65, 100, 128]))
    hi = int(rng.choice([256, 256, 16, 3]))
    g = torch.Generator(device=dev); g.manual_seed(int(rng.integers(1 << 30)))
    x = torch.randint(0, hi, (n, D), generator=g, device=dev, dtype=torch.uint8)
    ndup = int(rng.integers(0, 50))
    if ndup:
        src = int(rng.integers(0, n))
        x[torch.randint(0, n, (ndup,), generator=g, device=dev)] = x[src].clone()
    q = x[torch.randint(0, n, (nq,), generator=g, device=dev)].clone()
    q[:, : int(rng.integers(0, 4))] ^= 1
    cut = int(rng.integers(1, n))
    out = {}
    for v in (0, 1):
        cvt_amd.set_tuning("flat_variant", v)
        ix = cvt_amd.FlatIndex(2, D); ix.add(x[:cut]); ix.add(x[cut:])
        d, i = ix.search(q, k)
        out[v] = (d.clone(), i.clone())
        ix.close()
    ok = bool(torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]))
    print("case %d: n=%d D=%d nq=%d k=%d hi=%d dup=%d %s" % (c, n, D, nq, k, hi, ndup, "ok" if ok else "MISMATCH"), flush=True)
    if not ok:
        bad += 1
        w = (out[0][1] != out[1][1]).nonzero()[:5].tolist()
        print("  first differences (query, rank):", w)
        break
cvt_amd.set_tuning("flat_variant", 0)
sys.exit(1 if bad else 0)
