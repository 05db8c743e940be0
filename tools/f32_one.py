"""one fp32 flat search configuration on RootSIFT-shaped rows for the profiler: python tools/f32_one.py rows D nq k [metric]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, cvt_amd as amd
from cvt_amd import synth
n, D, nq, k = (int(v) for v in sys.argv[1:5])
metric = int(sys.argv[5]) if len(sys.argv) > 5 else 0
x = synth.sift_like(n, D, device=torch.device("cuda", 0))
q = synth.sift_like(nq, D, seed=0xBEEF, device=torch.device("cuda", 0))
ix = amd.FlatIndex(metric, D); ix.add(x)
for _ in range(6): ix.search(q, k)
torch.cuda.synchronize()
print("path", ix.last_search())
