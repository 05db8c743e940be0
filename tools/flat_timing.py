#!/usr/bin/env python3
"""Phase timing of flat_u8_rowtile_kernel (needs tools/ubench/libcvtmi_ftiming.so built with -DCVTMI_FLAT_TIMING)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cvt_amd.capi as capi
capi.LIB_PATH = os.path.join(ROOT, "tools", "ubench", "libcvtmi_ftiming.so")
import torch, cvt_amd
dev = torch.device("cuda", 0)
n, D = int(os.environ.get("ROWS", 2_000_000)), 512
g = torch.Generator(device=dev); g.manual_seed(5)
db = torch.randint(0, 256, (n, D), generator=g, device=dev, dtype=torch.uint8)
ix = cvt_amd.FlatIndex(2, D); ix.add(db)
q = torch.randint(0, 256, (1000, D), generator=g, device=dev, dtype=torch.uint8)
lib = cvt_amd.lib()
ix.search(q, 10); torch.cuda.synchronize()
out = (C.c_ulonglong * 8)()
lib.cvtmi_debug_flat_timing(out, 1)
import time
t0 = time.perf_counter(); ix.search(q, 10); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
lib.cvtmi_debug_flat_timing(out, 1)
tiles = (n + 31) // 32 * 4   # 4 query groups
names = ["LDS reads + MFMA issue", "test + pushes + compactions", "prefetch wait + LDS store", "barrier"]
print("search %.3f ms; per tile (wave 0 of each workgroup, shader clocks): " % ms + ", ".join("%s %.0f" % (nm, out[i] / tiles) for i, nm in enumerate(names)))
