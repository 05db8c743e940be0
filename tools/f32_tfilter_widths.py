"""fp32 flat search over row widths: the threshold filter (flat_f32_tfilter) against the other paths -- results compared, times per batch."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, cvt_amd
dev = torch.device("cuda", 0)
k = 100
GB = float(os.environ.get("GB", 0.5))
for D in [int(v) for v in os.environ.get("DS", "32,64,96,128,160,192,256,384,512").split(",")]:
    n = int(GB * 2**30 / (4 * D))
    g = torch.Generator(device=dev); g.manual_seed(D)
    cen = torch.randn((3000, D), generator=g, device=dev)
    x = cen[torch.randint(0, 3000, (n,), generator=g, device=dev)] + 0.6 * torch.randn((n, D), generator=g, device=dev)
    x = x / x.norm(dim=1, keepdim=True)
    for metric in (1, 0):
        ix = cvt_amd.FlatIndex(metric, D); ix.add(x)
        for nq in [int(v) for v in os.environ.get("NQS", "64,128,1000").split(",")]:
            q = x[torch.randint(0, n, (nq,), generator=g, device=dev)] + 0.2 * torch.randn((nq, D), generator=g, device=dev)
            q = (q / q.norm(dim=1, keepdim=True)).contiguous()
            res = {}
            for tf in (0, 4):
                cvt_amd.set_tuning("flat_f32_tfilter", tf)
                for _ in range(2): ix.search(q, k)
                torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 3
                for _ in range(reps): d, i = ix.search(q, k)
                torch.cuda.synchronize(); res[tf] = ((time.perf_counter() - t0) / reps * 1e3, d, i, ix.last_search()[0])
            same = bool(torch.equal(res[0][2], res[4][2]) and torch.equal(res[0][1].view(torch.int32), res[4][1].view(torch.int32)))
            print("D %d rows %d metric %d nq %d: path %d %.3f ms -> path %d %.3f ms identical=%s" % (D, n, metric, nq, res[0][3], res[0][0], res[4][3], res[4][0], same), flush=True)
        cvt_amd.set_tuning("flat_f32_tfilter", 4)
        ix.close()
    del x
