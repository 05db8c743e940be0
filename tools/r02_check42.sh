#!/bin/bash
cd /root/repo
timeout 150 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys; sys.path.insert(0, '.')
import torch, cvt_amd, bench
dev = torch.device("cuda", 0)
n = 10_000_000
for D in (128, 256):
    g = torch.Generator(device=dev); g.manual_seed(5)
    ix = cvt_amd.FlatIndex(2, D)
    for a in range(0, n, 1 << 21):
        ix.add(torch.randint(0, 256, (min(n, a + (1 << 21)) - a, D), generator=g, device=dev, dtype=torch.uint8))
    for nq in (256, 384, 512, 768, 1000, 1024):
        q = torch.randint(0, 256, (nq, D), generator=g, device=dev, dtype=torch.uint8)
        res = []
        for v in (0, 2):
            cvt_amd.set_tuning("flat_variant", v)
            ms = bench._ev_ms(torch, lambda: ix.search(q, 10), reps=4, warm=2)
            res.append("variant %d: %.3f ms (filtered=%s)" % (v, ms, ix.last_search()[0]))
        print("D=%d nq=%d  " % (D, nq) + "  ".join(res), flush=True)
    cvt_amd.set_tuning("flat_variant", 0)
    ix.close()
PY
