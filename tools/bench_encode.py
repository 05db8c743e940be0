#!/usr/bin/env python3
"""PQ encode (IVFOPQ::Add, IVFOPQ.cpp:135-163): the VALU kernel (reference chain for every centroid) against the
matrix-core filter + exact resolution kernel, on SIFT-shaped rows with k-means-trained codebooks (so that near
ties occur at their natural rate).  Codes of the two kernels must be identical."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, cvt_amd
from cvt_amd import synth
dev = torch.device("cuda", 0)

def timeit(f, reps=5, warm=3):
    for _ in range(warm): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

D, K = 128, 256
n = int(os.environ.get("ROWS", 1 << 20))
x = synth.sift_like(n, D, device=dev)
R = synth.random_rotation(D)
for M in (16, 8):
    idx0 = cvt_amd.OpqIndex(np.zeros((1, D), np.float32), np.zeros((M, K, D // M), np.float32), R=R)
    xr = idx0.rotate(x)
    books = synth.train_books(xr[:50000], M, K, iters=3)
    idx = cvt_amd.OpqIndex(np.zeros((1, D), np.float32), books, R=R)
    out = {}
    for v, name in ((1, "VALU chain"), (2, "matrix-core filter")):
        idx.set_param("encode_variant", v)
        ms = timeit(lambda: idx.encode(xr))
        out[v] = idx.encode(xr)[1]
        print("pq_encode M=%-2d %-18s n=%d: %.3f ms  %.1f M rows/s  (%.1f T reference ops/s)" % (M, name, n, ms, n / ms / 1e3, 3 * D * K * n / ms / 1e9))
    same = torch.equal(out[1], out[2])
    print("   codes identical:", same)
    assert same

# fallback rate (needs tools/ubench/libcvtmi_encstats.so built with -DCVTMI_ENC_STATS; see tools/README.md)
stats_lib = os.path.join(ROOT, "tools", "ubench", "libcvtmi_encstats.so")
if os.environ.get("ENC_STATS") and os.path.exists(stats_lib):
    import subprocess
    code = r'''
import ctypes as C, os, sys
sys.path.insert(0, %r)
import cvt_amd.capi as capi
capi.LIB_PATH = %r
import numpy as np, torch, cvt_amd
from cvt_amd import synth
D, K, n = 128, 256, 1 << 18
x = synth.sift_like(n, D, device="cuda"); R = synth.random_rotation(D)
for M in (16, 8):
    idx0 = cvt_amd.OpqIndex(np.zeros((1, D), np.float32), np.zeros((M, K, D // M), np.float32), R=R)
    xr = idx0.rotate(x)
    for label, books in (("k-means books", synth.train_books(xr[:50000], M, K, iters=3)), ("random books", (np.random.default_rng(0).normal(size=(M, K, D // M)) * 0.05).astype(np.float32))):
        idx = cvt_amd.OpqIndex(np.zeros((1, D), np.float32), books, R=R)
        idx.set_param("encode_variant", 2)
        out = (C.c_ulonglong * 2)()
        cvt_amd.lib().cvtmi_debug_encode_stats(out, 1)
        cvt_amd.lib().cvtmi_debug_encode_mode(8)  # count
        idx.encode(xr); torch.cuda.synchronize()
        cvt_amd.lib().cvtmi_debug_encode_stats(out, 1)
        print("M=%%d %%s: %%d (row, m) pairs, %%d through the exact chain = %%.3f %%%%" %% (M, label, out[0], out[1], 100.0 * out[1] / max(out[0], 1)))
''' % (ROOT, stats_lib)
    subprocess.run([sys.executable, "-c", code], check=True)
