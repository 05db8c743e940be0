#!/usr/bin/env python3
"""SQ8 wave kernels (d = 512): train / normalising encode rates over the round-5 switches -- decision filter on / off, wave sums on DPP or
through the ds_bpermute butterfly, workgroups per CU.  ROWS env (default 4 M)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, cvt_amd
dev = torch.device("cuda", 0)
n, d = int(os.environ.get("ROWS", 4_000_000)), int(os.environ.get("D", 512))
g = torch.Generator(device=dev); g.manual_seed(5)
x = torch.randn((n, d), generator=g, device=dev).relu_()
xc = x.clone()


def timeit(f, reps=5):
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


vmin, vdiff = cvt_amd.sq8_train(x, l2norm=True)
for _ in range(3):   # clocks
    cvt_amd.sq8_train(x, l2norm=True)
nb = n * d * 4
for filt, flags, blocks in ((0, 1, 3), (1, 0, 3), (1, 1, 3), (1, 1, 2), (1, 1, 4), (1, 1, 6), (1, 1, 8), (1, 0, 6)):
    cvt_amd.set_tuning("sq8_filter", filt); cvt_amd.set_tuning("sq8_flags", flags); cvt_amd.set_tuning("sq8_wave_blocks", blocks)
    t = timeit(lambda: cvt_amd.sq8_train(x, l2norm=True))
    e2 = timeit(lambda: cvt_amd.sq8_encode(vmin, vdiff, x, l2norm=2))
    e1 = timeit(lambda: cvt_amd.sq8_encode(vmin, vdiff, xc, l2norm=True))
    e0 = timeit(lambda: cvt_amd.sq8_encode(vmin, vdiff, x, l2norm=False))
    print("filter=%d dpp=%d wg/CU=%d: train %.2f TB/s | encode norm, no write-back %.2f TB/s alg | norm + write-back %.2f alg / %.2f traffic | no norm %.2f" % (
        filt, flags, blocks, nb / t / 1e9, nb * 1.25 / e2 / 1e9, nb * 1.25 / e1 / 1e9, nb * 2.25 / e1 / 1e9, nb * 1.25 / e0 / 1e9), flush=True)
