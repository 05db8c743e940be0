import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, cvt_amd as amd
from cvt_amd import synth
dev = torch.device("cuda", 0)
for n, D in ((65_536, 128), (100_000, 128), (100_000, 512), (200_000, 1024), (70_001, 96)):
    x = synth.sift_like(n, D, device=dev)
    for metric in (1, 0):
        ix = amd.FlatIndex(metric, D); ix.add(x)
        for nq in (1, 1000):
            q = synth.sift_like(nq, D, seed=0xBEEF, device=dev)
            for k in (129, 1000, 2048):
                res = {}
                for big in (1, 0):
                    amd.set_tuning("flat_f32_tfilter_bigk", big)
                    for _ in range(2): ix.search(q, k)
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for _ in range(3): d, i = ix.search(q, k)
                    torch.cuda.synchronize(); res[big] = ((time.perf_counter() - t0) / 3 * 1e3, d, i, ix.last_search()[0])
                same = bool(torch.equal(res[0][2], res[1][2]) and torch.equal(res[0][1].view(torch.int32), res[1][1].view(torch.int32)))
                print("n %d D %d metric %d nq %d k %d: path %d %.3f ms -> path %d %.3f ms identical=%s" % (n, D, metric, nq, k, res[0][3], res[0][0], res[1][3], res[1][0], same), flush=True)
        ix.close()
amd.set_tuning("flat_f32_tfilter_bigk", 1)
