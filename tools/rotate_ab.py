"""Rotation GEMM (Y = X R^T, 128-d) over batch sizes.  What the five-launch timing of rounds 1-5 measured (1 M rows 0.57-0.60 of the fp32 matrix peak, 8 M rows 0.72) was the clocks' ramp after
an idle gap: 400 back-to-back launches of 1 M rows run at 0.73 (bench.py: _ev_ms_steady).
Round 6 also measured, and removed: R in registers with one wave per SIMD (512 registers, no R in LDS: 0.54-0.61 / 0.66-0.72) and slabs handed out by an
atomic counter instead of a fixed stride (0.535 / 0.57) -- bits equal, neither faster."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, cvt_amd
from cvt_amd import synth
dev = torch.device("cuda", 0)
D, M = 128, 16
R = synth.random_rotation(D, seed=7)
books = np.random.default_rng(1).normal(size=(M, 256, D // M)).astype(np.float32)
ix = cvt_amd.OpqIndex(np.zeros((1, D), np.float32), books, R=R)
for n in (1 << 20, (1 << 20) + 77, 40_000, 1 << 23):
    x = synth.sift_like(n, D, device=dev)
    out = {}
    for v in (0,):
        for _ in range(3): y = ix.rotate(x)
        torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 10
        for _ in range(reps): y = ix.rotate(x)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
        out[v] = y
        print("rows %d (%d): %.4f ms = %.1f TF (%.3f of 157.3) %.0f GB/s" % (n, v, ms, 2.0 * n * D * D / ms / 1e9, 2.0 * n * D * D / ms / 1e9 / 157.3, n * D * 8 / ms / 1e6), flush=True)
