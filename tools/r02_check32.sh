#!/bin/bash
cd /root/repo
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys; sys.path.insert(0, '.')
import torch, cvt_amd, bench
dev = torch.device("cuda", 0)
d3 = 512
g = torch.Generator(device=dev); g.manual_seed(3)
feats = torch.randn((1 << 21, d3), generator=g, device=dev).clamp_(min=0)
nb = feats.numel() * 4
vmin, vdiff = cvt_amd.sq8_train(feats, l2norm=True)
codes = cvt_amd.sq8_encode(vmin, vdiff, feats, l2norm=False)
for b in (0, 255, 240, 192, 252, 384, 510, 0, 768, 250):
    cvt_amd.set_tuning("sq8_tile_blocks", b)
    ms_e = bench._ev_ms(torch, lambda: cvt_amd.sq8_encode(vmin, vdiff, feats, l2norm=False), reps=5, warm=2)
    ms_d = bench._ev_ms(torch, lambda: cvt_amd.sq8_decode(vmin, vdiff, codes), reps=5, warm=2)
    ms_t = bench._ev_ms(torch, lambda: cvt_amd.sq8_train(feats, l2norm=False), reps=5, warm=2)
    print("blocks=%d: encode %.3f ms %.2f TB/s; decode %.3f ms %.2f TB/s; train(no norm) %.3f ms %.2f TB/s" % (b, ms_e, nb * 1.25 / ms_e / 1e9, ms_d, nb * 1.25 / ms_d / 1e9, ms_t, nb / ms_t / 1e9))
PY
