#!/usr/bin/env python3
"""Randomised parity sweep of the SQ8 wave-per-row kernels' decision filter (round 5, csrc/sq8.hip) against the oracle's chain:
d = 256 / 512, 4096 .. 40 000 rows (the shapes that take those kernels), data families that sit on the filter's edges (half-zero rows,
signed rows, 40 binades inside a row, rows that are multiples of 1/255 of their norm so that many quotients land ON code boundaries,
constant columns, negative zeros, tiny / huge / non-finite rows), ranges trained on a subset and then perturbed (clamps on both sides,
vdiff 0 / tiny / huge / negative / NaN, |vmin| >> vdiff).  Codes, written-back rows and trained ranges must equal the oracle's; exits
non-zero on the first mismatch.  SEED / ITERS env."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch, cvt_amd
from oracle import binding as ob
ob.build()
orc = ob.Oracle()
bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
seed, iters = int(os.environ.get("SEED", 1)), int(os.environ.get("ITERS", 40))
rng = np.random.default_rng(seed)
for it in range(iters):
    d = int(rng.choice([512, 256])); n = int(rng.integers(4096, 40000))
    fam = int(rng.integers(0, 5))
    x = rng.normal(size=(n, d)).astype(np.float32)
    if fam == 0:
        x = np.maximum(x, 0)
    elif fam == 1:
        x = (x * np.exp2(rng.integers(-20, 20, size=(n, d)))).astype(np.float32)
    elif fam == 2:   # rows whose entries are small integers: after normalisation many quotients coincide / sit on boundaries
        x = rng.integers(-3, 9, size=(n, d)).astype(np.float32)
    elif fam == 3:
        x[rng.random(size=(n, d)) < 0.8] = 0
    x[rng.random(size=(n, d)) < 0.02] = -0.0
    for r in rng.integers(0, n, size=6):
        x[r] *= np.float32(rng.choice([1e-30, 1e30, 1e-20, 1e18, 0.0]))
    if rng.random() < 0.5:
        x[int(rng.integers(0, n)), int(rng.integers(0, d))] = rng.choice([np.inf, -np.inf, np.nan])
    x[:, int(rng.integers(0, d))] = np.float32(rng.normal())          # a constant column (before normalisation)
    sub = x[np.all(np.isfinite(x), axis=1)][: max(64, n // 3)]
    l2 = int(rng.choice([1, 2, 0]))   # 1 in-place normalisation, 2 normalised codes / rows left alone, 0 no normalisation
    vm, vd = orc.sq8_train(sub.copy(), l2norm=(l2 != 0))
    vm, vd = vm.copy(), vd.copy()
    for c in rng.integers(0, d, size=12):
        kind = int(rng.integers(0, 8))
        if kind == 0: vd[c] = 0.0
        elif kind == 1: vd[c] = np.float32(1e-41)
        elif kind == 2: vd[c] = np.float32(1e30)
        elif kind == 3: vd[c] = -abs(vd[c]) - np.float32(1e-3)
        elif kind == 4: vm[c] = np.float32(rng.normal() * 100)
        elif kind == 5: vd[c] *= np.float32(0.3)
        elif kind == 6: vm[c] += np.float32(0.4) * vd[c]
        else: vd[c] = np.nan
    oc, ox = orc.sq8_encode(vm, vd, x, l2norm=(l2 != 0))
    xt = torch.from_numpy(x.copy()).cuda()
    codes = cvt_amd.sq8_encode(torch.from_numpy(vm).cuda(), torch.from_numpy(vd).cuda(), xt, l2norm=l2).cpu().numpy()
    fin, finc = np.all(np.isfinite(ox), axis=1), np.isfinite(vd)      # (int) NaN is undefined in the reference itself
    ok = np.array_equal(codes[fin][:, finc], oc[fin][:, finc])
    ok = ok and (np.array_equal(bits(xt.cpu().numpy()), bits(ox)) if l2 == 1 else np.array_equal(bits(xt.cpu().numpy()), bits(x)))
    xf = x[np.all(np.isfinite(x), axis=1)]
    tv, td = cvt_amd.sq8_train(torch.from_numpy(xf.copy()).cuda(), l2norm=True)
    ovm, ovd = orc.sq8_train(xf.copy(), l2norm=True)
    tv, td = tv.cpu().numpy(), td.cpu().numpy()
    ok_t = np.array_equal(bits(tv), bits(ovm)) and np.array_equal(bits(td), bits(ovd))
    if not (ok and ok_t):
        bad = np.argwhere(codes[fin][:, finc] != oc[fin][:, finc])[:5]
        print("MISMATCH", dict(it=it, seed=seed, d=d, n=n, fam=fam, l2=l2, encode_ok=bool(ok), train_ok=bool(ok_t), first_bad=bad.tolist())); sys.exit(1)
    print("iter %d ok (d=%d n=%d family=%d l2norm=%d)" % (it, d, n, fam, l2), flush=True)
print("fuzz_sq8: %d iterations, no mismatch" % iters)
