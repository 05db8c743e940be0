#!/usr/bin/env python3
"""fp32 flat search (brute_force_search shape: 1 M x 128-d) -- queries/s by metric, batch size and k."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, cvt_amd
from cvt_amd import synth
dev = torch.device("cuda", 0)
n, D = int(os.environ.get("ROWS", 1 << 20)), int(os.environ.get("D", 128))
x = synth.sift_like(n, D, device=dev)
q = synth.sift_like(4096, D, seed=0xBEEF, device=dev)
for metric, name in ((0, "IP"), (1, "L2")):
    fi = cvt_amd.FlatIndex(metric, D); fi.add(x)
    for nq, k in [(int(a), int(b)) for a, b in (c.split(":") for c in os.environ.get("CASES", "1000:100,1000:10,4096:10,64:100,1:100").split(","))]:
        qq = q[:nq].contiguous()
        fi.search(qq, k); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): fi.search(qq, k)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
        print("flat %s f32 n=%d d=%d nq=%d k=%d: %.3f ms  %.0f QPS  %.1f T fp32 mul-add pairs/s" % (name, n, D, nq, k, ms, nq / ms * 1e3, n * D * nq / ms / 1e9), flush=True)
