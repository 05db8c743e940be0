#!/bin/bash
cd /root/repo
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys; sys.path.insert(0, '.')
import torch, cvt_amd, bench
dev = torch.device("cuda", 0)
for D in (512, 128):
    n = 10_000_000
    g = torch.Generator(device=dev); g.manual_seed(5)
    ix = cvt_amd.FlatIndex(2, D)
    for a in range(0, n, 1 << 21):
        ix.add(torch.randint(0, 256, (min(n, a + (1 << 21)) - a, D), generator=g, device=dev, dtype=torch.uint8))
    for nq in (1, 32):
        q = torch.randint(0, 256, (nq, D), generator=g, device=dev, dtype=torch.uint8)
        for b in (256, 255, 240, 192, 252, 256, 255):
            cvt_amd.set_tuning("flat_u8_mstream_blocks", b)
            ms = bench._ev_ms(torch, lambda: ix.search(q, 10), reps=10, warm=3)
            print("D=%d nq=%d blocks=%d: %.3f ms  %.2f TB/s" % (D, nq, b, ms, n * D / ms / 1e9))
    ix.close()
PY
