#!/usr/bin/env python3
"""Phase timing of adc_scan16h (scan_variant 6).  Needs the instrumented build:
   make -C cvt_amd/csrc OUT=$PWD/tools/ubench/timing EXTRA=-DCVTMI_SCAN_TIMING
Per item (wave 0's clocks): tables in, seed, look-ups + candidates, final selection, output; candidates stored per (item, query),
rare-path entries / cycles and bound updates of wave 0.  NQS / CFGS (balance:min_rows:splits) / ROWS / K env."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cvt_amd.capi as capi
capi.LIB_PATH = os.path.join(ROOT, "tools", "ubench", "timing", "libcvtmi.so")
import torch, cvt_amd
from cvt_amd import synth
dev = torch.device("cuda", 0)
D, M, K = 128, 16, 256
rows, k = int(os.environ.get("ROWS", 1_000_000)), int(os.environ.get("K", 100))
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
nqs = [int(v) for v in os.environ.get("NQS", "10000,1000,8").split(",")]
qs = synth.sift_like(max(nqs), D, seed=0xBEEF, device=dev)
lib = cvt_amd.lib()
idx.set_param("scan_variant", 6); idx.set_param("profile", 1)
GHZ = 2100.0   # cycles per us at the clock the scan sustains
for nq in nqs:
    q = qs[:nq].contiguous()
    for cfg in os.environ.get("CFGS", "0:0:0,2:0:0").split(","):
        bal, minr, sp = [int(v) for v in cfg.split(":")]
        cvt_amd.set_tuning("scanh_balance", bal); cvt_amd.set_tuning("scanh_min_rows", minr if minr else 16384); idx.set_param("splits", sp)
        for _ in range(3):
            idx.search(q, k)
        torch.cuda.synchronize(); idx.last_scan()
        out = (C.c_ulonglong * 8)(); cnt = (C.c_ulonglong * 8)()
        lib.cvtmi_debug_scanh_timing(out, 1); lib.cvtmi_debug_scanh_counters(cnt, 1)
        idx.search(q, k); torch.cuda.synchronize()
        s = idx.last_scan()
        lib.cvtmi_debug_scanh_timing(out, 1); lib.cvtmi_debug_scanh_counters(cnt, 1)
        items = max(1, cnt[5])
        names = ["tables in", "seed", "look-ups + candidates", "final selection", "output"]
        print("nq %d balance %d min_rows %d splits %d: kernel %.3f ms, %d items; per item us (@%.2f GHz): " % (nq, bal, minr, sp, s["ms"], items, GHZ / 1e3) +
              ", ".join("%s %.1f" % (n, out[i] / items / GHZ) for i, n in enumerate(names)), flush=True)
        print("   candidates per (item, query) %.0f; wave 0 per item: %.0f chunks, %.3f us per chunk all in; rare-path entries %.1f (%.2f us each), bound updates %.1f (%.2f us each); stops %d" % (
            cnt[0] / items / 8, cnt[7] / items, out[2] / max(1, cnt[7]) / GHZ, cnt[1] / items, cnt[2] / max(1, cnt[1]) / GHZ, cnt[3] / items, cnt[4] / max(1, cnt[3]) / GHZ, cnt[6]), flush=True)
