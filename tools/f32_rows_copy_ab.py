"""fp32 threshold filter: the exact finish out of the row-major copy ("flat_f32_rows_copy" 4, default) against the blocked rows (0) -- ms per search, lists and
bits compared; RootSIFT-shaped rows (tight at wide widths) and zero-mean clustered rows"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, cvt_amd as amd
from cvt_amd import synth
dev = torch.device("cuda", 0)
def ms(f, reps=10):
    f(); f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
def clustered(n, D, seed):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    cen = torch.randn((2000, D), generator=g, device=dev)
    x = cen[torch.randint(0, 2000, (n,), generator=g, device=dev)] + 0.7 * torch.randn((n, D), generator=g, device=dev)
    x = x / x.norm(dim=1, keepdim=True)
    q = x[torch.randint(0, n, (1000,), generator=g, device=dev)] + 0.2 * torch.randn((1000, D), generator=g, device=dev)
    return x, (q / q.norm(dim=1, keepdim=True)).contiguous()
for name, n, D in (("sift", 1_000_000, 128), ("sift", 4_000_000, 128), ("sift", 524_288, 512), ("sift", 262_144, 1024), ("clustered", 524_288, 512),
                   ("clustered", 262_144, 1024), ("clustered", 262_144, 2048), ("sift", 1_000_000, 100)):
    if name == "sift":
        x = synth.sift_like(n, D, device=dev); qa = synth.sift_like(1000, D, seed=0xBEEF, device=dev)
    else:
        x, qa = clustered(n, D, D)
    for metric in (0, 1):
        res = {}
        for copy in (4, 0):
            amd.set_tuning("flat_f32_rows_copy", copy)
            ix = amd.FlatIndex(metric, D); ix.add(x)
            row = []
            for nq, k in ((1000, 10), (1000, 100), (128, 100), (1000, 1000)):
                q = qa[:nq].contiguous()
                t = ms(lambda: ix.search(q, k), 5); d, i = ix.search(q, k)
                row.append((nq, k, t, d.clone(), i.clone(), ix.last_search()[0]))
            res[copy] = row
            ix.close()
        out = []
        for a, b in zip(res[4], res[0]):
            same = bool(torch.equal(a[4], b[4]) and torch.equal(a[3].view(torch.int32), b[3].view(torch.int32)))
            out.append("nq=%d k=%d: %.3f(%d) / %.3f(%d)%s" % (a[0], a[1], a[2], a[5], b[2], b[5], "" if same else " DIFFERENT"))
        print("%s n=%d D=%d metric=%d (copy / blocked)  " % (name, n, D, metric) + "   ".join(out), flush=True)
    del x
amd.set_tuning("flat_f32_rows_copy", 4)
