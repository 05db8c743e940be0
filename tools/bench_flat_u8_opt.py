#!/usr/bin/env python3
"""uint8 L2 flat search at the config 3 shape: row-tile kernels vs the filter pipeline with the old / the software-pipelined
(LDS-DMA) filter kernel.  Results compared bit for bit."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, cvt_amd
if os.environ.get("DBG_LIB"):   # a -DCVTMI_GF_DBG build of the library (tools/README.md)
    from cvt_amd import capi
    capi.LIB_PATH = os.environ["DBG_LIB"]
dev = torch.device("cuda", 0)
n, D = int(os.environ.get("ROWS", 10_000_000)), int(os.environ.get("D", 512))
g = torch.Generator(device=dev); g.manual_seed(5)
ix = cvt_amd.FlatIndex(2, D)
for a in range(0, n, 1 << 21):
    ix.add(torch.randint(0, 256, (min(n, a + (1 << 21)) - a, D), generator=g, device=dev, dtype=torch.uint8))
qs = torch.randint(0, 256, (4096, D), generator=g, device=dev, dtype=torch.uint8)
DBG = [int(v) for v in os.environ.get("DBG", "0").split(",")]
for nq, k in ((4096, 10),) if len(DBG) > 1 else ((4096, 10), (1000, 10), (3840, 10)):
    q = qs[:nq].contiguous()
    ref = None
    for dbg in DBG:
     if dbg: cvt_amd.set_tuning("flat_u8_dbg", dbg)   # needs a -DCVTMI_GF_DBG build (tools/README.md); results are wrong then
     if len(DBG) > 1: print("dbg", dbg)
     for name, fv, gf in (("row-tile", 1, 0), ("filter(8 waves x1)", 2, 3), ("filter(1 wave/SIMD)", 2, 4), ("filter(8 waves x1)", 2, 3), ("filter(1 wave/SIMD)", 2, 4)):
        cvt_amd.set_tuning("flat_variant", fv); cvt_amd.set_tuning("flat_u8_gfilter", gf)
        out = ix.search(q, k); torch.cuda.synchronize()
        if ref is None: ref = out
        same = bool(torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]))
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps): ix.search(q, k)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        print("flat L2 u8 %d-d n=%d nq=%d k=%d %-28s: %.3f ms  %.0f QPS  %.0f TOP/s  filtered=%s same=%s" % (
            D, n, nq, k, name, ms, nq / ms * 1e3, 2.0 * n * D * nq / ms / 1e9, ix.last_search()[0], same), flush=True)
cvt_amd.set_tuning("flat_variant", 0); cvt_amd.set_tuning("flat_u8_gfilter", 1)
