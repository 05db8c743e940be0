"""fp32 threshold filter, k = 129 .. 2048: one against three bf16 products per term by table size -- ms per 1000 queries (auto = the shipped rule), path
in brackets.  python tools/f32_products_by_k.py rows D"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, cvt_amd as amd
from cvt_amd import synth
n, D = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
x = synth.sift_like(n, D, device=dev)
q = synth.sift_like(1000, D, seed=0xBEEF, device=dev)
for metric in (0, 1):
    ix = amd.FlatIndex(metric, D); ix.add(x)
    for k in (129, 256, 384, 512, 768, 1024, 1536, 2048):
        row = []
        for v in (4, 1, 3):
            amd.set_tuning("flat_f32_tfilter", v)
            for _ in range(2): ix.search(q, k)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(4): ix.search(q, k)
            torch.cuda.synchronize()
            row.append("%s:%.3f(%d,%d)" % ({4: "auto", 1: "one", 3: "three"}[v], (time.perf_counter() - t) / 4 * 1e3, *ix.last_search()))
        print("metric=%d n=%d D=%d k=%d  " % (metric, n, D, k) + "  ".join(row), flush=True)
    ix.close()
amd.set_tuning("flat_f32_tfilter", 4)
