"""one uint8 flat search configuration for the profiler: python tools/flat_u8_one.py rows D nq k [min_k]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, cvt_amd as amd
n, D, nq, k = (int(v) for v in sys.argv[1:5])
amd.set_tuning("flat_u8_tfilter_min_k", int(sys.argv[5]) if len(sys.argv) > 5 else 1); amd.set_tuning("flat_u8_tfilter_min_nq", 1); amd.set_tuning("flat_u8_tfilter_min_nq_k65", 1)
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randint(0, 256, (n, D), dtype=torch.uint8, device="cuda", generator=g)
ix = amd.FlatIndex(2, D); ix.add(x)
q = x[torch.randint(0, n, (nq,), device="cuda", generator=g)].clone(); q[:, :5] ^= 3
for _ in range(6): ix.search(q, k)
torch.cuda.synchronize()
print("path", ix.last_search())
