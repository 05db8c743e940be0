#!/bin/bash
cd /root/repo
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys; sys.path.insert(0, '.')
import torch, cvt_amd, bench
dev = torch.device("cuda", 0)
for d3 in (512, 256):
    g = torch.Generator(device=dev); g.manual_seed(3)
    feats = torch.randn((1 << 21, d3), generator=g, device=dev).clamp_(min=0)
    nb = feats.numel() * 4
    for b in (3, 8, 5, 12, 3, 16, 24, 32, 6, 3, 8, 16):
        cvt_amd.set_tuning("sq8_wave_blocks", b)
        ms = bench._ev_ms(torch, lambda: cvt_amd.sq8_train(feats, l2norm=True), reps=5, warm=2)
        print("d=%d blocks/CU=%d: %.3f ms  %.2f TB/s" % (d3, b, ms, nb / ms / 1e9))
PY
