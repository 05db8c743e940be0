#!/usr/bin/env python3
"""Phase clocks of scan16h_prep_kernel (instrumented build: make -C cvt_amd/csrc OUT=$PWD/tools/ubench/timing EXTRA=-DCVTMI_SCAN_TIMING)."""
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
rng = np.random.default_rng(0)
books = (rng.normal(size=(M, K, D // M)) * 0.05).astype(np.float32)
idx = cvt_amd.OpqIndex(np.zeros((1, D), np.float32), books, R=synth.random_rotation(D))
g = torch.Generator(device=dev); g.manual_seed(1)
idx.add_codes(torch.randint(0, 256, (200_000, M), generator=g, device=dev, dtype=torch.uint8))
lib = cvt_amd.lib()
for nq, var in ((8, 7), (1000, 6), (10000, 6)):
    q = synth.sift_like(nq, D, seed=0xBEEF, device=dev)
    idx.set_param("scan_variant", var)
    for _ in range(3):
        idx.search(q, 100)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)()
    lib.cvtmi_debug_prep_timing(out, 1)
    idx.search(q, 100); torch.cuda.synchronize()
    lib.cvtmi_debug_prep_timing(out, 1)
    wgs = (nq + 7) // 8
    names = ["queries in", "entries + fp32 tables + ranges", "barrier", "scale + quantise", "image out"]
    print("nq %d: %d workgroups; per workgroup us @2.1 GHz: " % (nq, wgs) + ", ".join("%s %.1f" % (n, out[i] / wgs / 2100.0) for i, n in enumerate(names)), flush=True)
