import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, cvt_amd as amd
from cvt_amd import synth
dev = torch.device("cuda", 0)
def ms(f, reps=10):
    f(); f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
for n, D in ((524_288, 512), (262_144, 1024), (1_000_000, 100), (100_000, 512), (300_000, 2048)):
    x = synth.sift_like(n, D, device=dev)
    for metric in (0, 1):
        ix = amd.FlatIndex(metric, D); ix.add(x)
        for k in (10, 100):
            row = []
            for nq in (1, 2, 4, 8, 15):
                q = synth.sift_like(nq, D, seed=0xBEEF, device=dev)
                amd.set_tuning("flat_f32_tfilter_min", 1)
                t1 = ms(lambda: ix.search(q, k)); a = ix.search(q, k); how = ix.last_search()[0]
                amd.set_tuning("flat_f32_tfilter_min", 0)
                t0 = ms(lambda: ix.search(q, k)); b = ix.search(q, k); how0 = ix.last_search()[0]
                same = bool(torch.equal(a[1], b[1]) and torch.equal(a[0].view(torch.int32), b[0].view(torch.int32)))
                row.append("%d:%.3f(%d)/%.3f(%d)%s" % (nq, t1, how, t0, how0, "" if same else " DIFFERENT"))
            print("n=%d D=%d metric=%d k=%d (min 1 / default)  " % (n, D, metric, k) + "  ".join(row), flush=True)
        ix.close()
