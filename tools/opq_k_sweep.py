import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, cvt_amd
from cvt_amd import synth
dev = torch.device("cuda", 0)
D, M, K = 128, 16, 256
rng = np.random.default_rng(0)
books = (rng.normal(size=(M, K, D // M)) * 0.05).astype(np.float32)
g = torch.Generator(device=dev); g.manual_seed(1)
rows = 1_000_000
idx = cvt_amd.OpqIndex(np.zeros((1, D), np.float32), books, R=synth.random_rotation(D))
idx.add_codes(torch.randint(0, 256, (rows, M), generator=g, device=dev, dtype=torch.uint8))
for nq in [int(v) for v in os.environ.get('NQS', '10000,1000,8').split(',')]:
    q = torch.randn((nq, D), generator=g, device=dev) * 0.1
    for k in [int(v) for v in os.environ.get('KS', '1,10,50,100,128,129,200,500,1000,2048').split(',')]:
        for _ in range(2): idx.search(q, k)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); reps = 3
        for _ in range(reps): idx.search(q, k)
        torch.cuda.synchronize()
        print("rows=%d nq=%d k=%d: %.3f ms" % (rows, nq, k, (time.perf_counter() - t0) / reps * 1e3), flush=True)
