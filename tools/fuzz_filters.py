#!/usr/bin/env python3
"""Randomised sweep of the two other matrix-core filters against the exact kernels (bit-identical results required):
fp32 exhaustive search (flat_variant 2 vs 1) and nearest-centroid assignment (assign_variant 2 vs 1), over widths,
magnitudes, clustered / integer / sparse rows, duplicates, shuffled or sorted row order."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, cvt_amd
seed = int(os.environ.get("SEED", 1)); iters = int(os.environ.get("ITERS", 20))
rng = np.random.default_rng(seed)
dev = "cuda"
used = 0
for it in range(iters):
    D = int(rng.choice([32, 48, 64, 96, 128])); metric = int(rng.integers(0, 2))
    n = int(rng.choice([131072, 140001, 200000, 300000])); nq = int(rng.choice([64, 100, 257, 1000])); k = int(rng.choice([1, 5, 10, 100, 128]))
    kind = int(rng.integers(0, 5)); scale = float(rng.choice([1.0, 1e-4, 300.0, 1e6]))
    g = torch.Generator(device=dev); g.manual_seed(seed * 1000 + it)
    cen = torch.randn((300, D), generator=g, device=dev)
    x = cen[torch.randint(0, 300, (n,), generator=g, device=dev)] + float(rng.choice([0.05, 0.5])) * torch.randn((n, D), generator=g, device=dev)
    if kind == 1: x = torch.round(x * 3)                      # integers: many exact ties
    if kind == 2: x = x.relu(); x[:, ::3] = 0                 # sparse
    if kind == 3: x = x / x.norm(dim=1, keepdim=True).clamp_min(1e-9)
    x = (x * scale).contiguous()
    if rng.random() < 0.5: x[n // 2:n // 2 + 500] = x[3]      # duplicates
    if kind == 4:                                             # sorted by one coordinate: the leading sample is unrepresentative
        x = x[torch.argsort(x[:, 0])].contiguous()
    q = (x[torch.randint(0, n, (nq,), generator=g, device=dev)] + 0.1 * scale * torch.randn((nq, D), generator=g, device=dev)).contiguous()
    q[0] = x[3]
    out = {}
    for v in (2, 1):
        cvt_amd.set_tuning("flat_variant", v)
        ix = cvt_amd.FlatIndex(metric, D); ix.add(x)
        out[v] = ix.search(q, k)
        if v == 2: u, worst = ix.last_search(); used += u
    ok = torch.equal(out[1][1], out[2][1]) and torch.equal(out[1][0].view(torch.int32), out[2][0].view(torch.int32))
    # assignment: the same rows against random / sampled centroids
    kc = int(rng.choice([64, 100, 500, 3000]))
    cent = (x[torch.randint(0, n, (kc,), generator=g, device=dev)] + (0.0 if rng.random() < 0.3 else 0.01 * scale) * torch.randn((kc, D), generator=g, device=dev)).cpu().numpy()
    cent[kc // 2] = cent[1]
    books = np.zeros((D // 8, 16, 8), np.float32); books[:, 1:] = (np.random.default_rng(it).normal(size=(D // 8, 15, 8)) * scale).astype(np.float32)
    res = {}
    for v in (2, 1):
        cvt_amd.set_tuning("assign_variant", v)
        res[v] = cvt_amd.OpqIndex(cent, books).encode(x[:150000])[0]
    ok2 = torch.equal(res[1], res[2])
    print("it %d flat(metric=%d D=%d n=%d nq=%d k=%d kind=%d scale=%g filter=%s worst=%d) %s | assign(k=%d) %s" % (
        it, metric, D, n, nq, k, kind, scale, u, worst, "ok" if ok else "MISMATCH", kc, "ok" if ok2 else "MISMATCH"), flush=True)
    if not ok:
        di = (out[1][1] != out[2][1]) | (out[1][0].view(torch.int32) != out[2][0].view(torch.int32))
        qs = torch.nonzero(di.any(dim=1)).ravel()
        print('  queries affected:', qs.numel(), qs[:8].tolist())
        r = int(qs[0]); c = int(torch.nonzero(di[r]).ravel()[0])
        lo = max(0, c - 2)
        print('  query', r, 'col', c, 'exact ids', out[1][1][r, lo:c + 4].tolist(), 'd', out[1][0][r, lo:c + 4].tolist())
        print('  query', r, 'col', c, 'filt  ids', out[2][1][r, lo:c + 4].tolist(), 'd', out[2][0][r, lo:c + 4].tolist())
    if not (ok and ok2): sys.exit(1)
cvt_amd.set_tuning("flat_variant", 0); cvt_amd.set_tuning("assign_variant", 0)
print("fuzz_filters: %d iterations, filter answered %d of them, no mismatch" % (iters, used))
