#!/usr/bin/env python3
"""Phase timing of the fp32 stream's collect / finish kernels (needs tools/ubench/libcvtmi_fstiming.so, built with -DCVTMI_FS_TIMING)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cvt_amd.capi as capi
capi.LIB_PATH = os.path.join(ROOT, "tools", "ubench", "libcvtmi_fstiming.so")
import torch, cvt_amd
dev = torch.device("cuda", 0)
n, D = int(os.environ.get("ROWS", 1_000_000)), int(os.environ.get("D", 128))
g = torch.Generator(device=dev); g.manual_seed(5)
ix = cvt_amd.FlatIndex(int(os.environ.get("METRIC", 0)), D); ix.add(torch.randn((n, D), generator=g, device=dev))
lib = cvt_amd.lib()
out = (C.c_ulonglong * 16)()
for nq in [int(v) for v in os.environ.get("NQS", "1,8,96").split(",")]:
    q = torch.randn((nq, D), generator=g, device=dev)
    ix.search(q, 100); torch.cuda.synchronize()
    lib.cvtmi_debug_fs_timing(out, 1)
    reps = 5
    for _ in range(reps): ix.search(q, 100)
    torch.cuda.synchronize()
    lib.cvtmi_debug_fs_timing(out, 1)
    names = {0: "collect: load maxima + |q|^2", 1: "collect: select", 2: "collect: scan slice", 8: "finish: load list", 9: "finish: select",
             10: "finish: candidate rows", 11: "finish: exact distances", 12: "finish: rank + write"}
    print("nq=%d (workgroup 0, us): " % nq + ", ".join("%s %.2f" % (names[i], out[i] / reps / 100.0) for i in sorted(names)), flush=True)
