"""fp32 threshold filter on non-negative unit-norm rows (RootSIFT-shaped, any width): ms per 1000-query search with one / two / three bf16 products
per term ("flat_f32_tfilter" 1 / 2 / 3) against the shipped choice (4) and the exact kernels -- python tools/f32_products_sweep.py rows D"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, cvt_amd as amd
from cvt_amd import synth
n, D = int(sys.argv[1]), int(sys.argv[2])
x = synth.sift_like(n, D, device=torch.device("cuda", 0))
q = synth.sift_like(1000, D, seed=0xBEEF, device=torch.device("cuda", 0))
for metric in (1, 0):
    ix = amd.FlatIndex(metric, D); ix.add(x)
    for k in (10, 100, 128, 512, 2048):
        row = []
        for v in (4, 1, 2, 3):
            amd.set_tuning("flat_f32_tfilter", v)
            for _ in range(2): ix.search(q, k)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(5): ix.search(q, k)
            torch.cuda.synchronize()
            row.append("%d:%.3f(%d)" % (v, (time.perf_counter() - t) / 5 * 1e3, ix.last_search()[0]))
        print("metric=%d n=%d D=%d k=%d  " % (metric, n, D, k) + "  ".join(row), flush=True)
    ix.close()
amd.set_tuning("flat_f32_tfilter", 4)
