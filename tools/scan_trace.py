#!/usr/bin/env python3
"""Per-workgroup timeline of adc_scan16q (timing build, see scan_timing.py): start/end on the 100 MHz wall
clock, the CU each workgroup ran on, how many ran at a time."""
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
rows, nq, k = int(os.environ.get("ROWS", 1_000_000)), int(os.environ.get("NQ", 10_000)), int(os.environ.get("K", 100))
rng = np.random.default_rng(0)
books = (rng.normal(size=(M, K, D // M)) * 0.05).astype(np.float32)
idx = cvt_amd.OpqIndex(np.zeros((1, D), np.float32), books, R=synth.random_rotation(D))
g = torch.Generator(device=dev); g.manual_seed(1)
idx.add_codes(torch.randint(0, 256, (rows, M), generator=g, device=dev, dtype=torch.uint8))
q = synth.sift_like(nq, D, seed=0xBEEF, device=dev)
lib = cvt_amd.lib()
for var, tail in ((4, 0), (4, 1), (3, 0)):
    idx.set_param("scan_variant", var); idx.set_param("splits", 0); idx.set_param("tail_split", tail); idx.set_param("profile", 1)
    idx.search(q, k); torch.cuda.synchronize(); idx.last_scan()
    idx.search(q, k); torch.cuda.synchronize()
    s = idx.last_scan()
    nb = 16384
    tr = np.zeros((nb, 4), np.uint64)
    lib.cvtmi_debug_scan_trace(tr.ctypes.data_as(C.c_void_p), nb)
    tr = tr[tr[:, 1] > 0]
    t0 = tr[:, 0].min()
    st = (tr[:, 0] - t0).astype(np.float64) / 100.0  # us
    en = (tr[:, 1] - t0).astype(np.float64) / 100.0
    dur = en - st
    hw = tr[:, 3] & np.uint64(0xffffffff); xcc = (tr[:, 3] >> np.uint64(32)) & np.uint64(0xf)
    cu = (hw >> np.uint64(8)) & np.uint64(0xf); sh = (hw >> np.uint64(12)) & np.uint64(1); se = (hw >> np.uint64(13)) & np.uint64(7)
    cuid = xcc * np.uint64(1024) + se * np.uint64(32) + sh * np.uint64(16) + cu
    print("variant %d tail %d: kernel %.3f ms (events), %d workgroups traced, span %.1f us" % (var, tail, s["ms"], len(tr), en.max()))
    print("  duration us: min %.0f p10 %.0f median %.0f p90 %.0f max %.0f; shader clock during a workgroup: %.2f GHz" % (
        dur.min(), np.percentile(dur, 10), np.median(dur), np.percentile(dur, 90), dur.max(),
        np.median(tr[:, 2].astype(np.float64) / dur) / 1e3))
    print("  distinct CUs used: %d; first start spread %.1f us" % (len(np.unique(cuid)), np.sort(st)[min(511, len(st) - 1)]))
    # concurrency over time
    edges = np.linspace(0, en.max(), 41)
    conc = [(np.minimum(en, b) - np.maximum(st, a)).clip(0).sum() / (b - a) for a, b in zip(edges[:-1], edges[1:])]
    print("  workgroups in flight per 1/40 of the span: " + " ".join("%d" % round(c) for c in conc))
    # by dispatch order
    order = np.argsort(st)
    print("  durations by start order (mean of each eighth): " + " ".join("%.0f" % dur[order][i * len(dur) // 8:(i + 1) * len(dur) // 8].mean() for i in range(8)))
