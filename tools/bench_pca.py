#!/usr/bin/env python3
"""PCA projection (pca_utils.cc:25-35) against the fp32 MFMA roofline: 2*din*dout flop per row
(157 TFLOP/s dense fp32 matrix peak), 4*(din + dout) bytes per row."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, cvt_amd
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5)

def timeit(f, reps=20):
    for _ in range(5): f()  # the first launches after an allocation run at ramping clocks
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

for din, dout, n in ((1024, 128, 1 << 20), (2048, 256, 1 << 19), (512, 64, 1 << 21), (1024, 128, 4096)):
    x = torch.randn((n, din), generator=g, device=dev).relu_()
    e = torch.linalg.qr(torch.randn((din, din), generator=g, device=dev))[0][:dout].contiguous()
    mean = x[:1000].mean(dim=0).contiguous()
    for l2 in (True, False):
        ms = timeit(lambda: cvt_amd.pca_project(mean, e, x, l2norm=l2))
        print("pca_project %7d x %d -> %d l2norm=%d: %.3f ms  %.1f TFLOP/s (%.0f%% of 157)  %.2f TB/s  %.1f M rows/s" % (
            n, din, dout, l2, ms, 2.0 * n * din * dout / ms / 1e9, 2.0 * n * din * dout / ms / 1e9 / 1.57, n * 4 * (din + dout) / ms / 1e9, n / ms / 1e3))
    del x
