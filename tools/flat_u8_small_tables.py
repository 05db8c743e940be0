"""uint8 threshold filter on tables under 262 144 rows, k <= 128: ms per search against the streaming passes, by batch size -- where does it pay?
python tools/flat_u8_small_tables.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, cvt_amd as amd
g = torch.Generator(device="cuda").manual_seed(1)
def ms(f, reps=10):
    f(); f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
for n, D in ((65_536, 128), (65_536, 512), (100_000, 256), (131_072, 128), (200_000, 512), (250_000, 64)):
    x = torch.randint(0, 256, (n, D), dtype=torch.uint8, device="cuda", generator=g)
    ix = amd.FlatIndex(2, D); ix.add(x)
    for k in (10, 100):
        row = []
        for nq in (129, 192, 256, 384, 512, 768, 1000, 2048, 4096):
            q = x[torch.randint(0, n, (nq,), device="cuda", generator=g)].clone(); q[:, :5] ^= 3
            amd.set_tuning("flat_u8_tfilter_small_min_nq", 1)
            t1 = ms(lambda: ix.search(q, k)); a = ix.search(q, k); how = ix.last_search()[0]
            amd.set_tuning("flat_u8_tfilter_small_min_nq", 1 << 30)
            t0 = ms(lambda: ix.search(q, k)); b = ix.search(q, k)
            same = bool(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]))
            row.append("%d:%.3f/%.3f%s%s" % (nq, t1, t0, "" if how == 4 else "?", "" if same else " DIFFERENT"))
        print("n=%d D=%d k=%d (filter/stream)  " % (n, D, k) + "  ".join(row), flush=True)
    ix.close()
# widths the streaming kernel does not take (it exists at 128 / 256 / 512-d): small batches through the filter against the row-tile kernels
for n, D in ((250_000, 64), (1_000_000, 96), (1_000_000, 192), (500_000, 384)):
    x = torch.randint(0, 256, (n, D), dtype=torch.uint8, device="cuda", generator=g)
    ix = amd.FlatIndex(2, D); ix.add(x)
    for k in (10, 100):
        row = []
        for nq in (1, 4, 8, 16, 32, 64, 128):
            q = x[torch.randint(0, n, (nq,), device="cuda", generator=g)].clone(); q[:, :5] ^= 3
            for key in ("flat_u8_tfilter_small_min_nq", "flat_u8_tfilter_min_nq", "flat_u8_tfilter_min_nq_k65"): amd.set_tuning(key, 1)
            t1 = ms(lambda: ix.search(q, k)); a = ix.search(q, k); how = ix.last_search()[0]
            for key in ("flat_u8_tfilter_small_min_nq", "flat_u8_tfilter_min_nq", "flat_u8_tfilter_min_nq_k65"): amd.set_tuning(key, 1 << 30)
            t0 = ms(lambda: ix.search(q, k)); b = ix.search(q, k)
            same = bool(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]))
            row.append("%d:%.3f/%.3f%s%s" % (nq, t1, t0, "" if how == 4 else "?", "" if same else " DIFFERENT"))
        print("n=%d D=%d k=%d (filter/row-tile kernels)  " % (n, D, k) + "  ".join(row), flush=True)
    ix.close()
