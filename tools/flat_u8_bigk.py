"""uint8 flat search with k > 128 (round 6): the threshold filter of flat_u8_tfilter.hip against the exact kernels ("flat_u8_tfilter" 0)
-- ms per call, identical lists.  python tools/flat_u8_bigk.py [rows] [D]"""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import cvt_amd as amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 512
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randint(0, 256, (n, D), dtype=torch.uint8, device="cuda", generator=g)
ix = amd.FlatIndex(2, D); ix.add(x)
out = {"rows": n, "d": D, "cases": {}}


def ms(f, reps):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


amd.set_tuning("flat_u8_tfilter_min_k", 1); amd.set_tuning("flat_u8_tfilter_min_nq", 1); amd.set_tuning("flat_u8_tfilter_min_nq_k65", 1)   # (every k through the pipeline: where does it pay below 129?)
for nq, k in ((4096, 10), (1000, 10), (512, 10), (256, 10), (128, 10), (96, 10), (64, 10), (32, 10), (1000, 64), (1000, 100), (512, 100), (256, 100), (128, 100), (96, 100), (64, 100), (32, 100), (1000, 128), (1000, 129), (1000, 512), (1000, 2048), (128, 129), (16, 1000), (1, 129)):
    q = x[torch.randint(0, n, (nq,), device="cuda", generator=g)].clone()
    q[:, :5] ^= 3
    amd.set_tuning("flat_u8_tfilter", 1)
    t1 = ms(lambda: ix.search(q, k), 5)
    d1, i1 = ix.search(q, k)
    how = ix.last_search()[0]
    amd.set_tuning("flat_u8_tfilter", 0)
    t0 = ms(lambda: ix.search(q, k), 1 if k > 128 else 5)
    how0 = ix.last_search()[0]
    d, i = ix.search(q, k)
    out["cases"]["nq=%d k=%d" % (nq, k)] = {"ms": round(t1, 3), "path": how, "without_ms": round(t0, 3), "without_path": how0,
                                           "identical": bool(torch.equal(d1, d) and torch.equal(i1, i))}
    print(nq, k, out["cases"]["nq=%d k=%d" % (nq, k)], flush=True)
amd.set_tuning("flat_u8_tfilter", 1); amd.set_tuning("flat_u8_tfilter_min_k", 1); amd.set_tuning("flat_u8_tfilter_min_nq", 129); amd.set_tuning("flat_u8_tfilter_min_nq_k65", 97)
print(json.dumps(out))
