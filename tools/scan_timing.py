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
rng = np.random.default_rng(0)
books = (rng.normal(size=(M, K, D // M)) * 0.05).astype(np.float32)
idx = cvt_amd.OpqIndex(np.zeros((1, D), np.float32), books, R=synth.random_rotation(D))
g = torch.Generator(device=dev); g.manual_seed(1)
idx.add_codes(torch.randint(0, 256, (rows, M), generator=g, device=dev, dtype=torch.uint8))
q = synth.sift_like(nq, D, seed=0xBEEF, device=dev)
lib = cvt_amd.lib()
for var, sp in ((4, 1), (3, 1)):
    idx.set_param("scan_variant", var); idx.set_param("splits", sp); idx.set_param("profile", 1)
    idx.search(q, k); torch.cuda.synchronize(); idx.last_scan()
    out = (C.c_ulonglong * 8)()
    lib.cvtmi_debug_scan_timing(out, 1)
    tko = (C.c_ulonglong * 4)()
    lib.cvtmi_debug_topk(tko, 1)
    idx.search(q, k); torch.cuda.synchronize()
    s = idx.last_scan()
    lib.cvtmi_debug_scan_timing(out, 1)
    lib.cvtmi_debug_topk(tko, 1)
    nb = (nq + 7) // 8 * sp
    names = ["prologue", "lookups+push", "wait at checkpoint", "compaction + barrier", "final compaction"]
    print("variant %d splits %d k %d: kernel %.3f ms, %d blocks; per-block us (shader clock @ ~2.35 GHz): " % (var, sp, k, s["ms"], nb) +
          ", ".join("%s %.1f" % (n, out[i] / nb / 2350.0) for i, n in enumerate(names)), flush=True)
    print("   compactions of query 0 per block %.1f, new entries each %.1f, exact-fix %.2f us, sort %.2f us (thread 0, @2.35 GHz)" % (
        tko[0] / nb, tko[3] / max(1, tko[0]), tko[1] / max(1, tko[0]) / 2350.0, tko[2] / max(1, tko[0]) / 2350.0), flush=True)
