#!/usr/bin/env python3
"""cvtmi_opq_search with HOST buffers (the reference's call shape) on the bench's data: queries/s over the number of chunks the
batch is pipelined in (cvtmi_set_tuning "opq_host_chunks"), next to the device-pointer rate.  ROWS / NQ / K env."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cvt_amd
from cvt_amd import synth
dev = torch.device("cuda", 0)
D, M, K = 128, 16, 256
rows, nq, k = int(os.environ.get("ROWS", 1_000_000)), int(os.environ.get("NQ", 10_000)), int(os.environ.get("K", 100))
zero = np.zeros((1, D), np.float32)
R = synth.random_rotation(D, seed=7)
tmp = cvt_amd.OpqIndex(zero, np.zeros((M, K, D // M), np.float32), R=R)
books = synth.train_books(tmp.rotate(synth.sift_like(100_000, D, seed=0xC0FFEE, device=dev)), M, K, iters=4)
tmp.close()
idx = cvt_amd.OpqIndex(zero, books, R=R)
idx.reserve(rows)
step = synth.CHUNK * 4
for a in range(0, rows, step):
    b = min(rows, a + step)
    _, codes = idx.encode(idx.rotate(synth.sift_like(b - a, D, seed=0xC0FFEE, row_begin=a, device=dev)))
    idx.add_codes(codes)
q = synth.sift_like(nq, D, seed=0xBEEF, device=dev)
for _ in range(10):
    d0, i0 = idx.search(q, k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    idx.search(q, k)
torch.cuda.synchronize()
print("device pointers: %.3f ms per batch of %d, %.0f queries/s" % ((time.perf_counter() - t0) / 10 * 1e3, nq, nq * 10 / (time.perf_counter() - t0)), flush=True)
qh = q.cpu().numpy()
out = (np.zeros((nq, k), np.float32), np.zeros((nq, k), np.int64))
ref = (d0.cpu().numpy(), i0.cpu().numpy())
for chunks in [int(v) for v in os.environ.get("CHUNKS", "0,2048,4096,4104,5000,8192").split(",")]:
    cvt_amd.set_tuning("opq_host_chunk", chunks)
    for _ in range(3):
        idx.search(qh, k, rotate=True, out=out)
    same = bool(np.array_equal(out[1], ref[1]) and np.array_equal(out[0].view(np.uint32), ref[0].view(np.uint32)))
    t0 = time.perf_counter()
    for _ in range(8):
        idx.search(qh, k, rotate=True, out=out)
    el = (time.perf_counter() - t0) / 8
    print("host pointers, pieces of %d queries: %.3f ms per batch, %.0f queries/s, same=%s" % (chunks, el * 1e3, nq / el, same), flush=True)

cvt_amd.set_tuning("opq_host_chunk", 4096)
qp = cvt_amd.pinned_empty((nq, D), np.float32); qp[:] = qh
outp = (cvt_amd.pinned_empty((nq, k), np.float32), cvt_amd.pinned_empty((nq, k), np.int64))
# page-locked result arrays (cvtmi_host_alloc): 0 = pipelined pieces + copy engines, 1 = the kernels write them, one launch chain (round 5)
for zc in (0, 1):
    cvt_amd.set_tuning("opq_host_zero_copy", zc)
    for name, qa in (("pageable", qh), ("page-locked", qp)):
        for nqs in [int(v) for v in os.environ.get("NQS", str(nq)).split(",")]:
            oz = (outp[0][:nqs], outp[1][:nqs])
            for _ in range(3):
                idx.search(qa[:nqs], k, rotate=True, out=oz)
            same = bool(np.array_equal(oz[1], ref[1][:nqs]) and np.array_equal(oz[0].view(np.uint32), ref[0][:nqs].view(np.uint32)))
            t0 = time.perf_counter()
            for _ in range(8):
                idx.search(qa[:nqs], k, rotate=True, out=oz)
            el = (time.perf_counter() - t0) / 8
            print("host pointers, page-locked results, %s queries, zero_copy=%d, %d queries: %.3f ms per batch, %.0f queries/s, same=%s" % (
                name, zc, nqs, el * 1e3, nqs / el, same), flush=True)
