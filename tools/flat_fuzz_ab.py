"""random shapes around the threshold filters' dispatch bounds (rows, width, batch, k, duplicates, appends), uint8 and fp32: the default routes against
the round-5 kernels ("flat_u8_tfilter" 0 / "flat_variant" 1) -- lists and distance bits compared.  python tools/flat_fuzz_ab.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, cvt_amd as amd


def run(cases, seed, verbose=True):
    """-> (number of cases whose lists or bits differ, {path: count})"""
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda", 0)
    bad = 0
    paths = {}
    for c in range(cases):
        u8 = bool(rng.integers(0, 2))
        D = int(rng.choice([32, 64, 96, 128, 160, 192, 224, 256, 288, 320, 352, 384, 416, 448, 480, 512]) if u8 else rng.choice([32, 64, 100, 128, 160, 256, 384, 512, 768, 1024]))
        n = int(rng.choice([65_536, 65_537, 70_001, 100_000, 131_072, 200_003, 262_143, 262_144, 270_001, 400_000]))
        nq = int(rng.choice([1, 2, 3, 15, 16, 17, 64, 65, 96, 97, 128, 129, 130, 255, 256, 257, 300, 511, 513, 1000, 1025, 1100]))
        k = int(rng.choice([1, 5, 10, 32, 33, 64, 65, 100, 128, 129, 200, 512, 513, 1000, 1024, 1025, 2048]))
        if n * D * (1 if u8 else 4) > 1.2e9: n = 100_000
        g = torch.Generator(device=dev); g.manual_seed(c)
        if u8:
            hi = int(rng.choice([256, 256, 256, 16, 4]))
            x = torch.randint(0, hi, (n, D), dtype=torch.uint8, device=dev, generator=g)
        else:
            cen = torch.randn((500, D), generator=g, device=dev)
            x = cen[torch.randint(0, 500, (n,), generator=g, device=dev)] + 0.5 * torch.randn((n, D), generator=g, device=dev)
            if rng.integers(0, 2): x = x.clamp_(min=0)
            x = (x / x.norm(dim=1, keepdim=True).clamp_min(1e-6)).contiguous()
        x[1000:1000 + int(rng.integers(1, 400))] = x[7]
        q = x[torch.randint(0, n, (nq,), device=dev, generator=g)].clone()
        if u8: q[:, :3] ^= 1
        else: q = (q + 0.05 * torch.randn(q.shape, generator=g, device=dev)).contiguous()
        q[0] = x[7]
        metric = 2 if u8 else int(rng.integers(0, 2))
        cut = int(rng.integers(n // 2, n))
        ix = amd.FlatIndex(metric, D); ix.add(x[:cut]); ix.search(q[:min(nq, 3)], min(k, 10)); ix.add(x[cut:])
        d1, i1 = ix.search(q, k); how = ix.last_search()
        if u8: amd.set_tuning("flat_u8_tfilter", 0)
        else: amd.set_tuning("flat_variant", 1)
        d0, i0 = ix.search(q, k)
        amd.set_tuning("flat_u8_tfilter", 1); amd.set_tuning("flat_variant", 0)
        ix.close()
        same = bool(torch.equal(i1, i0) and torch.equal(d1.view(torch.int32), d0.view(torch.int32)))
        paths[how[0]] = paths.get(how[0], 0) + 1
        if not same:
            bad += 1
            if verbose: print("DIFFERENT: case %d u8=%s D=%d n=%d nq=%d k=%d path=%s" % (c, u8, D, n, nq, k, how), flush=True)
    return bad, paths


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    bad, paths = run(n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("cases %d, different %d, paths %s" % (n_cases, bad, paths))
