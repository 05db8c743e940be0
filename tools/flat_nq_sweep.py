#!/usr/bin/env python3
"""flat search over ROWS x D (METRIC 0 = IP, 1 = L2 fp32, 2 = L2 uint8), default dispatch, batch sizes NQS: ms per search, TB/s of rows."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, cvt_amd
dev = torch.device("cuda", 0)
n, D, metric = int(os.environ.get("ROWS", 1_000_000)), int(os.environ.get("D", 128)), int(os.environ.get("METRIC", 0))
g = torch.Generator(device=dev); g.manual_seed(5)
ix = cvt_amd.FlatIndex(metric, D)
def rows(m):
    if metric == 2: return torch.randint(0, 256, (m, D), generator=g, device=dev, dtype=torch.uint8)
    return torch.randn((m, D), generator=g, device=dev)
for a in range(0, n, 1 << 21):
    ix.add(rows(min(n, a + (1 << 21)) - a))
k = int(os.environ.get("K", 10))
if os.environ.get("DBG"): cvt_amd.set_tuning("flat_f32_dbg", int(os.environ["DBG"]))
if os.environ.get("SHARE"): cvt_amd.set_tuning("flat_f32_share", int(os.environ["SHARE"]))
eb = 1 if metric == 2 else 4
for nq in [int(v) for v in os.environ.get("NQS", "1,2,4,8,16,32,64,128,256,512,1000,4096").split(",")]:
    q = rows(nq)
    for _ in range(2):
        ix.search(q, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        ix.search(q, k)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"metric={metric} rows={n} D={D} nq={nq} k={k}: {ms:.3f} ms, {n * D * eb / ms / 1e9:.2f} TB/s of rows, {2.0 * nq * n * D / ms / 1e9:.1f} Tflop/s, filtered={ix.last_search()[0]}", flush=True)
