#!/usr/bin/env python3
"""Phase timing of adc_scan16q (needs tools/ubench/libcvtmi_timing.so built with -DCVTMI_SCAN_TIMING)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cvt_amd.capi as capi
capi.LIB_PATH = os.path.join(ROOT, "tools", "ubench", "libcvtmi_timing.so")
import torch, cvt_amd
from cvt_amd import synth
dev = torch.device("cuda", 0)
D, M, K = 128, 16, 256
rows, nq, k = 1_000_000, 10_000, int(os.environ.get("K", 100))
if os.environ.get("DATA", "bench") == "random":   # uniform random codes under random codebooks: crowded bands, exact-key compactions
    rng = np.random.default_rng(0)
    books = (rng.normal(size=(M, K, D // M)) * 0.05).astype(np.float32)
    idx = cvt_amd.OpqIndex(np.zeros((1, D), np.float32), books, R=synth.random_rotation(D))
    g = torch.Generator(device=dev); g.manual_seed(1)
    idx.add_codes(torch.randint(0, 256, (rows, M), generator=g, device=dev, dtype=torch.uint8))
else:                                              # the bench's data: SIFT-shaped rows under trained codebooks
    zero = np.zeros((1, D), np.float32)
    R = synth.random_rotation(D, seed=7)
    tmp = cvt_amd.OpqIndex(zero, np.zeros((M, K, D // M), np.float32), R=R)
    books = synth.train_books(tmp.rotate(synth.sift_like(100_000, D, seed=0xC0FFEE, device=dev)), M, K, iters=4)
    tmp.close()
    idx = cvt_amd.OpqIndex(zero, books, R=R)
    idx.reserve(rows)
    step = synth.CHUNK * 4
    for a in range(0, rows, step):
        b = min(rows, a + step)
        _, codes = idx.encode(idx.rotate(synth.sift_like(b - a, D, seed=0xC0FFEE, row_begin=a, device=dev)))
        idx.add_codes(codes)
q = synth.sift_like(nq, D, seed=0xBEEF, device=dev)
lib = cvt_amd.lib()
for var, sp in [tuple(int(v) for v in c.split(':')) for c in os.environ.get('CFGS', '3:1,5:1,5:2').split(',')]:
    idx.set_param("scan_variant", var); idx.set_param("splits", sp); idx.set_param("profile", 1)
    idx.search(q, k); torch.cuda.synchronize(); idx.last_scan()
    out = (C.c_ulonglong * 8)()
    lib.cvtmi_debug_scan_async(None, 1)
    lib.cvtmi_debug_scan_timing(out, 1)
    tko = (C.c_ulonglong * 4)()
    lib.cvtmi_debug_topk(tko, 1)
    idx.search(q, k); torch.cuda.synchronize()
    s = idx.last_scan()
    lib.cvtmi_debug_scan_timing(out, 1)
    lib.cvtmi_debug_topk(tko, 1)
    asy = (C.c_ulonglong * 8)()
    lib.cvtmi_debug_scan_async(asy, 1)
    nb = (nq + 7) // 8 * sp
    names = ["prologue", "lookups+push", "wait at checkpoint", "compaction + barrier", "final compaction"]
    print("variant %d splits %d k %d: kernel %.3f ms, %d blocks; per-block us (shader clock @ ~2.35 GHz): " % (var, sp, k, s["ms"], nb) +
          ", ".join("%s %.1f" % (n, out[i] / nb / 2350.0) for i, n in enumerate(names)), flush=True)
    print("   compactions of query 0 per block %.1f, new entries each %.1f, exact-fix %.2f us, sort %.2f us (thread 0, @2.35 GHz)" % (
        tko[0] / nb, tko[3] / max(1, tko[0]), tko[1] / max(1, tko[0]) / 2350.0, tko[2] / max(1, tko[0]) / 2350.0), flush=True)
    if var == 5:
        nw = nb * 15
        us = lambda c: c / 2350.0
        print("   seed %.1f us; compactions per block %.1f, %.2f us each (%.1f us per block over all waves); rounds of waiting for room per scanning wave %.1f" % (
            us(out[3]) / nb, asy[2] / nb, us(asy[3]) / max(1, asy[2]), us(asy[3]) / nb, asy[5] / nw), flush=True)
        c = max(1, asy[2])
        print("   per compaction: entries taken %.1f, kept %.1f, lazy %.2f, counter beyond the capacity at the lock %.3f" % (
            asy[0] / c, asy[1] / c, asy[6] / c, asy[7] / c), flush=True)
