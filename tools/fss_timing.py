#!/usr/bin/env python3
"""Section clocks of the fp32 shared-ring stream kernel (waves 0 and 4 of workgroup 0; tools/ubench/libcvtmi_fstiming.so, -DCVTMI_FS_TIMING)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cvt_amd.capi as capi
capi.LIB_PATH = os.path.join(ROOT, "tools", "ubench", "libcvtmi_fstiming.so")
import torch, cvt_amd
dev = torch.device("cuda", 0)
n, D = int(os.environ.get("ROWS", 1_000_000)), int(os.environ.get("D", 128))
g = torch.Generator(device=dev); g.manual_seed(5)
ix = cvt_amd.FlatIndex(0, D); ix.add(torch.randn((n, D), generator=g, device=dev))
lib = cvt_amd.lib()
out = (C.c_ulonglong * 16)()
names = ["barrier", "products", "update", "requests", "vmcnt wait", "convert", "-", "-"]
for share in (2, 1):
    cvt_amd.set_tuning("flat_f32_share", share)
    nq = 256
    q = torch.randn((nq, D), generator=g, device=dev)
    ix.search(q, 100); torch.cuda.synchronize()
    ix.search(q, 100); torch.cuda.synchronize()
    lib.cvtmi_debug_fss_timing(out)
    tiles = (n + 31) // 32 / 256
    for w in (0, 1):
        print("share=%d wave %d, clocks per tile: " % (share, 4 * w) + ", ".join("%s %.0f" % (names[i], out[8 * w + i] / tiles) for i in range(6)), flush=True)
