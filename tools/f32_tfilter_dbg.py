"""fp32 threshold filter, timing experiments (flat_f32_dbg: 1 = rows fetched once, 2 = nothing passes, 4 = no operand reads): results are wrong."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, cvt_amd
from cvt_amd import synth
dev = torch.device("cuda", 0)
n, D, k, nq = 1_000_000, 128, 100, int(os.environ.get("NQ", 1000))
x = synth.sift_like(n, D, device=dev)
ix = cvt_amd.FlatIndex(1, D); ix.add(x)
q = synth.sift_like(nq, D, seed=0xBEEF, device=dev)
for tf in [int(v) for v in os.environ.get("TFS", "2").split(",")]:
    cvt_amd.set_tuning("flat_f32_tfilter", tf)
    for dbg in (8, 8 + 32, 8 + 96, 8 + 16, 8 + 16 + 96):
        cvt_amd.set_tuning("flat_f32_dbg", dbg)
        for _ in range(2): ix.search(q, k)
        torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 5
        for _ in range(reps): ix.search(q, k)
        torch.cuda.synchronize(); print("tf %d dbg %d: %.3f ms" % (tf, dbg, (time.perf_counter() - t0) / reps * 1e3), flush=True)
cvt_amd.set_tuning("flat_f32_dbg", 0)
