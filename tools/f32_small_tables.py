"""fp32 threshold filter on tables under 262 144 rows ("flat_f32_tfilter_min_rows" 65536 against the default), k = 10 / 100, by batch size; lists and bits compared"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, cvt_amd as amd
from cvt_amd import synth
dev = torch.device("cuda", 0)
def ms(f, reps=10):
    f(); f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
for n, D in ((65_536, 128), (131_072, 128), (200_000, 256), (100_000, 512), (65_536, 1024), (200_000, 1024), (100_000, 100)):
    x = synth.sift_like(n, D, device=dev)
    for metric in (0, 1):
        ix = amd.FlatIndex(metric, D); ix.add(x)
        for k in (10, 100):
            row = []
            for nq in (16, 65, 97, 129, 256, 1000, 4096):
                q = synth.sift_like(nq, D, seed=0xBEEF, device=dev)
                amd.set_tuning("flat_f32_tfilter_min_rows", 65536)
                t1 = ms(lambda: ix.search(q, k)); a = ix.search(q, k); how = ix.last_search()[0]
                amd.set_tuning("flat_f32_tfilter_min_rows", 262144)
                t0 = ms(lambda: ix.search(q, k)); b = ix.search(q, k); how0 = ix.last_search()[0]
                same = bool(torch.equal(a[1], b[1]) and torch.equal(a[0].view(torch.int32), b[0].view(torch.int32)))
                row.append("%d:%.3f(%d)/%.3f(%d)%s" % (nq, t1, how, t0, how0, "" if same else " DIFFERENT"))
            print("n=%d D=%d metric=%d k=%d (min_rows 65536 / default)  " % (n, D, metric, k) + "  ".join(row), flush=True)
        ix.close()
