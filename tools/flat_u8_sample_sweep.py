"""uint8 threshold filter: ms per 1000-query search against the sample divisor ("flat_u8_tfilter_sample": one tile group in v feeds the thresholds;
about k x v rows per query pass them) -- python tools/flat_u8_sample_sweep.py rows D"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, cvt_amd as amd
n, D = int(sys.argv[1]), int(sys.argv[2])
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randint(0, 256, (n, D), dtype=torch.uint8, device="cuda", generator=g)
ix = amd.FlatIndex(2, D); ix.add(x)
q = x[torch.randint(0, n, (1000,), device="cuda", generator=g)].clone(); q[:, :5] ^= 3
for k in (10, 32, 64, 128, 256, 512, 1024):
    row = []
    for v in (0, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64):
        amd.set_tuning("flat_u8_tfilter_sample", v)
        for _ in range(2): ix.search(q, k)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): ix.search(q, k)
        torch.cuda.synchronize()
        row.append("%d:%.3f%s" % (v, (time.perf_counter() - t) / 5 * 1e3, "" if ix.last_search()[0] == 4 else "!"))
    print("n=%d D=%d k=%d  " % (n, D, k) + "  ".join(row), flush=True)
amd.set_tuning("flat_u8_tfilter_sample", 0)
