"""fp32 flat search: the threshold filter (flat_f32_tfilter 1 / 2 / 3 products) against the stream kernels, result by result."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, cvt_amd
from cvt_amd import synth
dev = torch.device("cuda", 0)
k = 100
for D in (128, 64):
    n = 300_000
    x = synth.sift_like(n, D, device=dev) if D == 128 else torch.randn((n, D), device=dev)
    for metric in (1, 0):
        ix = cvt_amd.FlatIndex(metric, D); ix.add(x)
        cvt_amd.set_tuning("flat_f32_tfilter_min", 1)
        for nq in (32, 100, 256, 700):
            q = synth.sift_like(nq, D, seed=0xBEEF, device=dev) if D == 128 else torch.randn((nq, D), device=dev)
            cvt_amd.set_tuning("flat_f32_tfilter", 0); d0, i0 = ix.search(q, k)
            for tf in (3, 2, 1):
                cvt_amd.set_tuning("flat_f32_tfilter", tf); d, i = ix.search(q, k); torch.cuda.synchronize()
                bad = (i != i0).any(dim=1)
                badd = (d.view(torch.int32) != d0.view(torch.int32)).any(dim=1)
                msg = ""
                if bad.any():
                    b = int(bad.nonzero()[0])
                    msg = " first bad query %d: %d ids differ, first at rank %d; missing from the set: %d" % (
                        b, int((i[b] != i0[b]).sum()), int((i[b] != i0[b]).nonzero()[0]), len(set(i0[b].tolist()) - set(i[b].tolist())))
                print("D %d metric %d nq %d tf %d: bad ids %d bad dists %d%s" % (D, metric, nq, tf, int(bad.sum()), int(badd.sum()), msg), flush=True)
        cvt_amd.set_tuning("flat_f32_tfilter", 4)
        ix.close()
