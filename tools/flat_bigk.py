"""fp32 flat search with k beyond 128 (round 6: the threshold filter with 4096 sample maxima, lists of 32 768 candidates and a workgroup-wide
selection) against the exact kernels ("flat_f32_tfilter_bigk" 0): wall time, lists and distance bits compared."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, cvt_amd
from cvt_amd import synth
dev = torch.device("cuda", 0)
n, D = 1_000_000, 128
x = synth.sift_like(n, D, device=dev)
for metric in (1, 0):
    ix = cvt_amd.FlatIndex(metric, D); ix.add(x)
    for nq in (1, 100, 1000):
        q = synth.sift_like(nq, D, seed=0xBEEF, device=dev)
        for k in (129, 256, 1000, 2048):
            res = {}
            for big in (0, 1):
                cvt_amd.set_tuning("flat_f32_tfilter_bigk", big)
                for _ in range(2): ix.search(q, k)
                torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 3
                for _ in range(reps): d, i = ix.search(q, k)
                torch.cuda.synchronize(); res[big] = ((time.perf_counter() - t0) / reps * 1e3, d, i, ix.last_search()[0])
            same = bool(torch.equal(res[0][2], res[1][2]) and torch.equal(res[0][1].view(torch.int32), res[1][1].view(torch.int32)))
            print("metric %d nq %d k %d: path %d %.3f ms -> path %d %.3f ms identical=%s" % (metric, nq, k, res[0][3], res[0][0], res[1][3], res[1][0], same), flush=True)
    ix.close()
cvt_amd.set_tuning("flat_f32_tfilter_bigk", 1)
