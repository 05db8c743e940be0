"""fp32 flat search, large batches: the stream kernels against the round-6 threshold filter (flat_f32_tfilter), results compared."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, cvt_amd
from cvt_amd import synth
dev = torch.device("cuda", 0)
n, D, k = int(os.environ.get("ROWS", 1_000_000)), int(os.environ.get("D", 128)), int(os.environ.get("K", 100))
x = synth.sift_like(n, D, device=dev) if D == 128 else torch.randn((n, D), device=dev)
for metric in (0, 1):
    ix = cvt_amd.FlatIndex(metric, D); ix.add(x)
    for nq in [int(v) for v in os.environ.get("NQS", "130,256,512,1000,2000,4096").split(",")]:
        q = synth.sift_like(nq, D, seed=0xBEEF, device=dev) if D == 128 else torch.randn((nq, D), device=dev)
        ref = None
        for tf in [int(v) for v in os.environ.get("TFS", "0,3,2,1").split(",")]:
            cvt_amd.set_tuning("flat_f32_tfilter", tf)
            for _ in range(2): ix.search(q, k)
            torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 5
            for _ in range(reps): d, i = ix.search(q, k)
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
            same = ""
            if ref is not None: same = " identical=%s" % bool(torch.equal(i, ref[1]) and torch.equal(d.view(torch.int32), ref[0].view(torch.int32)))
            else: ref = (d, i)
            print("metric %d nq %d tfilter %d: %.3f ms%s" % (metric, nq, tf, ms, same), flush=True)
    cvt_amd.set_tuning("flat_f32_tfilter", 4)
    ix.close()
