import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, cvt_amd
from cvt_amd import synth
dev = torch.device("cuda", 0)
n, D, k = 1_000_000, 128, 100
x = synth.sift_like(n, D, device=dev)
ref = {}
for metric in (0, 1):
    ix = cvt_amd.FlatIndex(metric, D); ix.add(x)
    for share in (0, 3):
        cvt_amd.set_tuning("flat_f32_share", share)
        for nq in (512, 1000, 2000):
            q = synth.sift_like(nq, D, seed=0xBEEF, device=dev)
            for _ in range(2): ix.search(q, k)
            torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 5
            for _ in range(reps): d, i = ix.search(q, k)
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
            key = (metric, nq)
            same = ""
            if key in ref: same = " identical=%s" % bool(torch.equal(i, ref[key][1]) and torch.equal(d.view(torch.int32), ref[key][0].view(torch.int32)))
            else: ref[key] = (d, i)
            print("metric %d share %d nq %d: %.3f ms%s" % (metric, share, nq, ms, same), flush=True)
    cvt_amd.set_tuning("flat_f32_share", 0)
    ix.close()
