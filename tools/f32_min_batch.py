"""fp32 flat search, 16 .. 128 queries: the threshold filter ("flat_f32_tfilter_min" 1) against the stream kernels by batch size -- where the automatic bound (ft_auto_min) sits"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, cvt_amd as amd
from cvt_amd import synth
dev = torch.device("cuda", 0)
def ms(f, reps=20):
    f(); f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
for n, D in ((1_000_000, 128), (4_000_000, 128), (1_000_000, 64), (1_000_000, 256)):
    x = synth.sift_like(n, D, device=dev)
    ix = amd.FlatIndex(0, D); ix.add(x)
    for k in (10, 100):
        row = []
        for nq in (16, 32, 48, 64, 80, 96, 112, 128):
            q = synth.sift_like(nq, D, seed=0xBEEF, device=dev)
            amd.set_tuning("flat_f32_tfilter_min", 1); t1 = ms(lambda: ix.search(q, k)); p1 = ix.last_search()[0]
            amd.set_tuning("flat_f32_tfilter_min", 1 << 20); t0 = ms(lambda: ix.search(q, k)); p0 = ix.last_search()[0]
            row.append("%d:%.3f(%d)/%.3f(%d)" % (nq, t1, p1, t0, p0))
        print("n=%d D=%d k=%d (filter / stream)  " % (n, D, k) + "  ".join(row), flush=True)
    ix.close()
amd.set_tuning("flat_f32_tfilter_min", 0)
