#!/usr/bin/env python3
"""Per-kernel throughput of the non-scan parts of the path: rotation GEMM, PQ encode, LUT, flat search, SQ8.
Development aid (HIP events via torch on the current stream, which the library launches on)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cvt_amd
from cvt_amd import synth

dev = torch.device("cuda", 0)


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


D, M, K = 128, 16, 256
n = 1 << 20
x = synth.sift_like(n, D, device=dev)
R = synth.random_rotation(D)
rng = np.random.default_rng(0)
books = (rng.normal(size=(M, K, D // M)) * 0.05).astype(np.float32)
zero = np.zeros((1, D), np.float32)
idx = cvt_amd.OpqIndex(zero, books, R=R)
ms = timeit(lambda: idx.rotate(x))
print("rotate (MFMA fp32 GEMM)  n=%d: %.3f ms  %.1f TFLOP/s  %.2f TB/s (in+out)" % (n, ms, 2 * n * D * D / ms / 1e9, 2 * n * D * 4 / ms / 1e9))
perm = synth.random_permutation(D)
idxp = cvt_amd.OpqIndex(zero, books, perm=perm)
ms = timeit(lambda: idxp.rotate(x))
print("rotate (permutation)     n=%d: %.3f ms  %.2f TB/s (in+out)" % (n, ms, 2 * n * D * 4 / ms / 1e9))
xr = idx.rotate(x)
ms = timeit(lambda: idx.encode(xr))
print("pq_encode M=16           n=%d: %.3f ms  %.1f M rows/s  %.1f T ops/s (3*D*K per row)" % (n, ms, n / ms / 1e3, 3 * D * K * n / ms / 1e9))
books8 = (rng.normal(size=(8, K, 16)) * 0.05).astype(np.float32)
idx8 = cvt_amd.OpqIndex(zero, books8, R=R)
ms = timeit(lambda: idx8.encode(xr))
print("pq_encode M=8            n=%d: %.3f ms  %.1f M rows/s" % (n, ms, n / ms / 1e3))
coarse = rng.normal(size=(1024, D)).astype(np.float32) * 0.1
idxc = cvt_amd.OpqIndex(coarse, books, R=R)
nn = 1 << 17
ms = timeit(lambda: idxc.encode(xr[:nn]), reps=2, warm=1)
print("encode coarseK=1024      n=%d: %.3f ms  %.2f M rows/s" % (nn, ms, nn / ms / 1e3))
q = synth.sift_like(10000, D, seed=0xBEEF, device=dev)
ms = timeit(lambda: idx.lut(q))
print("lut nq=10000: %.3f ms" % ms)
# flat
for metric, name, dt in ((0, "IP f32", torch.float32), (1, "L2 f32", torch.float32)):
    fi = cvt_amd.FlatIndex(metric, D); fi.add(x)
    for nq in (1, 64, 1000):
        ms = timeit(lambda: fi.search(q[:nq].contiguous(), 100), reps=2, warm=1)
        print("flat %s n=%d nq=%d k=100: %.3f ms  %.1f QPS  %.2f TB/s-equivalent per query" % (name, n, nq, ms, nq / ms * 1e3, n * D * 4 * nq / ms / 1e9))
g = torch.Generator(device=dev); g.manual_seed(5)
u8 = torch.randint(0, 256, (2_000_000, 512), generator=g, device=dev, dtype=torch.uint8)
fi = cvt_amd.FlatIndex(2, 512); fi.add(u8)
for nq in (1, 64, 1000):
    qq = u8[:nq].contiguous()
    ms = timeit(lambda: fi.search(qq, 10), reps=2, warm=1)
    print("flat L2 u8 512-d n=2M nq=%d k=10: %.3f ms  %.1f QPS  %.1f T int-MAC/s" % (nq, ms, nq / ms * 1e3, 2e6 * 512 * nq / ms / 1e9))
xf = torch.randn((1_000_000, 512), generator=g, device=dev).relu_()
vmin, vdiff = cvt_amd.sq8_train(xf)
ms = timeit(lambda: cvt_amd.sq8_train(xf), reps=2, warm=1)
print("sq8_train 1M x 512: %.3f ms  %.2f TB/s" % (ms, 1e6 * 512 * 4 * 2 / ms / 1e9))
xc = xf.clone()
ms = timeit(lambda: cvt_amd.sq8_encode(vmin, vdiff, xc), reps=2, warm=1)
print("sq8_encode 1M x 512 (norm + encode, in place): %.3f ms  %.2f TB/s" % (ms, 1e6 * 512 * (4 * 3 + 1) / ms / 1e9))
