#!/usr/bin/env python3
"""SQ8 kernels against the HBM roofline (config C3 width: 512-d).  Algorithmic bytes per row: train 4d,
encode 4d + d, decode d + 4d; the in-place normalisation of Int8Encode (int8_quan.cc:76-79) adds a 4d write."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, cvt_amd
dev = torch.device("cuda", 0)
n, d = int(os.environ.get("ROWS", 4_000_000)), int(os.environ.get("D", 512))
g = torch.Generator(device=dev); g.manual_seed(5)
x = torch.randn((n, d), generator=g, device=dev).relu_()

def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

vmin, vdiff = cvt_amd.sq8_train(x, l2norm=True)
for l2 in (True, False):
    ms = timeit(lambda: cvt_amd.sq8_train(x, l2norm=l2))
    print("sq8_train  %dM x %d l2norm=%d: %.3f ms  algorithmic %.2f TB/s (= HBM traffic)" % (n // 10**6, d, l2, ms, n * d * 4 / ms / 1e9))
xc = x.clone()
codes = cvt_amd.sq8_encode(vmin, vdiff, xc, l2norm=True)
for l2 in (True, False):
    ms = timeit(lambda: cvt_amd.sq8_encode(vmin, vdiff, xc, l2norm=l2))
    moved = n * d * (9 if l2 else 5)
    print("sq8_encode %dM x %d l2norm=%d: %.3f ms  algorithmic %.2f TB/s, HBM traffic %.2f TB/s" % (n // 10**6, d, l2, ms, n * d * 5 / ms / 1e9, moved / ms / 1e9))
ms = timeit(lambda: cvt_amd.sq8_decode(vmin, vdiff, codes))
print("sq8_decode %dM x %d: %.3f ms  algorithmic %.2f TB/s (= HBM traffic)" % (n // 10**6, d, ms, n * d * 5 / ms / 1e9))
