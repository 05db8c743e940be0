#!/usr/bin/env python3
"""HNSW traversals per CU (config 5: 1 M nodes, M = 32, efC = 80, 10 000 queries): queries/s over the slot cap, fp32 vectors and OPQ codes."""
import os, sys, time, subprocess, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, cvt_amd
from cvt_amd import synth
dev = torch.device("cuda", 0)
D, n, nq = 128, int(os.environ.get("NODES", 1_000_000)), int(os.environ.get("NQ", 10_000))
rng = np.random.default_rng(5)
cen = rng.normal(size=(1000, D)).astype(np.float32)
x = cen[rng.integers(0, 1000, n)] + 0.6 * rng.normal(size=(n, D)).astype(np.float32)
x /= np.linalg.norm(x, axis=1, keepdims=True)
q = x[rng.integers(0, n, nq)] + 0.15 * rng.normal(size=(nq, D)).astype(np.float32)
q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
tmpd = tempfile.mkdtemp()
rows_p, idx_p = os.path.join(tmpd, "rows.bin"), os.path.join(tmpd, "graph.hnsw")
x.tofile(rows_p)
subprocess.run([os.path.join(ROOT, "cvt_amd", "bin", "hnsw_build"), rows_p, str(D), "32", "80", idx_p, "ip", "-", "0"], check=True, capture_output=True)
ix = cvt_amd.HnswIndex(open(idx_p, "rb").read(), cvt_amd.IP, D)
qd = torch.from_numpy(q).to(dev)
R = synth.random_rotation(D, seed=7)
tmp = cvt_amd.OpqIndex(np.zeros((1, D), np.float32), np.zeros((16, 256, D // 16), np.float32), R=R)
xr = tmp.rotate(torch.from_numpy(x).to(dev))
_, books5 = cvt_amd.opq_train(xr[:50_000].contiguous(), 1, 16, 256, 8, 1)
opq = cvt_amd.OpqIndex(np.zeros((1, D), np.float32), books5.cpu().numpy(), R=R)
_, codes = opq.encode(xr)
opq.add_codes(codes)


def ms(f, reps=2):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for cap in (0, 28, 24, 20, 16, 12):
    cvt_amd.set_tuning("hnsw_slots", cap)
    r = []
    for kk, ef in ((5, 1000), (10, 64)):
        r.append("fp32 ef=%d %.0f K q/s" % (ef, nq / ms(lambda: ix.search(qd, kk, ef))))
        r.append("adc ef=%d %.0f K q/s" % (ef, nq / ms(lambda: ix.search_adc(opq, qd, kk, ef))))
    print("slots cap %d: %s" % (cap, " | ".join(r)), flush=True)
