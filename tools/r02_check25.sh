#!/bin/bash
cd /root/repo
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys
sys.path.insert(0, '.')
import torch, cvt_amd
dev = torch.device("cuda", 0)
n, D = 10_000_000, 512
g = torch.Generator(device=dev); g.manual_seed(5)
ix = cvt_amd.FlatIndex(2, D)
for a in range(0, n, 1 << 21):
    ix.add(torch.randint(0, 256, (min(n, a + (1 << 21)) - a, D), generator=g, device=dev, dtype=torch.uint8))
cvt_amd.set_tuning("flat_variant", 1)
for nq in (64, 65, 96, 128, 129, 160, 192, 255, 256, 257, 384, 500, 512, 1000, 1024):
    q = torch.randint(0, 256, (nq, D), generator=g, device=dev, dtype=torch.uint8)
    for _ in range(2): ix.search(q, 10)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): ix.search(q, 10)
    e1.record(); torch.cuda.synchronize()
    print("row-tile only: nq=%d %.3f ms" % (nq, e0.elapsed_time(e1) / 3), flush=True)
PY
