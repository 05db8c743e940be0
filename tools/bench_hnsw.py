#!/usr/bin/env python3
"""HNSW search throughput (BASELINE config 5 shape: batched 10 K queries over one graph in HBM).
The graph is built on the host by the reference itself (oracle/_ref/libref_hnsw.so, test infrastructure) with
the reference's own parameters M = 32, efConstruction = 80 (makeIdx.cpp:303-304), ef = 1000
(siftsIndex.cpp:51); the CPU column is the reference's searchKnn on one host core, same graph, same queries."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch, cvt_amd
from oracle import binding as ob
n, D = int(os.environ.get("ROWS", 100_000)), int(os.environ.get("D", 128))
nq = int(os.environ.get("NQ", 10_000))
M, efc = int(os.environ.get("M", 32)), int(os.environ.get("EFC", 80))
rng = np.random.default_rng(5)
cen = rng.normal(size=(1000, D)).astype(np.float32)
x = cen[rng.integers(0, 1000, n)] + 0.6 * rng.normal(size=(n, D)).astype(np.float32)
x /= np.linalg.norm(x, axis=1, keepdims=True)
q = x[rng.integers(0, n, nq)] + 0.15 * rng.normal(size=(nq, D)).astype(np.float32)
q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
path = os.path.join(tempfile.gettempdir(), "bench.hnsw")
rh = ob.RefHnsw()
t0 = time.time(); rh.build(0, x, path, M, efc); print("reference build: %d x %d, M=%d efC=%d: %.1f s" % (n, D, M, efc, time.time() - t0), flush=True)
blob = open(path, "rb").read()
ix = cvt_amd.HnswIndex(blob, 0, D)
qd = torch.from_numpy(q).cuda()
exact = torch.argmax(qd @ torch.from_numpy(x).cuda().T, dim=1).cpu().numpy()
for k, ef in ((5, 1000), (5, 200), (10, 64)):
    ix.search(qd, k, ef); torch.cuda.synchronize()
    t0 = time.perf_counter(); d, lab = ix.search(qd, k, ef); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    cs = 200
    t0 = time.perf_counter(); rd, rl = rh.search(0, D, path, q[:cs], k, ef); t_cpu = time.perf_counter() - t0
    same = np.array_equal(rl, lab[:cs].cpu().numpy()) and np.array_equal(rd.view(np.uint32), d[:cs].cpu().numpy().view(np.uint32))
    print("k=%d ef=%d: GPU %.1f ms for %d queries = %.0f queries/s; reference CPU (1 core, %d queries) %.0f queries/s; identical=%s; recall@1 %.3f" % (
        k, ef, ms, nq, nq / ms * 1e3, cs, cs / t_cpu, same, float((lab[:, 0].cpu().numpy() == exact).mean())), flush=True)
# ---- over OPQ-compressed vectors (BASELINE config 5): same graph, 16-byte codes instead of 512-byte vectors ----
from cvt_amd import synth
R = synth.random_rotation(D, seed=7)
tmp = cvt_amd.OpqIndex(np.zeros((1, D), np.float32), np.zeros((16, 256, D // 16), np.float32), R=R)
xr = tmp.rotate(torch.from_numpy(x).cuda())
_, books = cvt_amd.opq_train(xr[:50_000].contiguous(), 1, 16, 256, 8, 1)
opq = cvt_amd.OpqIndex(np.zeros((1, D), np.float32), books.cpu().numpy(), R=R)
_, codes = opq.encode(xr)
opq.add_codes(codes)
for k, ef in ((5, 1000), (5, 200), (10, 64)):
    ix.search_adc(opq, qd, k, ef); torch.cuda.synchronize()
    t0 = time.perf_counter(); d, lab = ix.search_adc(opq, qd, k, ef); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    print("ADC over OPQ codes (M=16) k=%d ef=%d: GPU %.1f ms for %d queries = %.0f queries/s; recall@1 vs exact %.3f; recall@%d %.3f" % (
        k, ef, ms, nq, nq / ms * 1e3, float((lab[:, 0].cpu().numpy() == exact).mean()), k,
        float((lab.cpu().numpy() == exact[:, None]).any(axis=1).mean())), flush=True)
os.remove(path)
