#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02n
timeout 900 python -m pytest tests/test_gpu_flat_sq8.py -x -q -m gpu -k "sq8" > gpurun_out/r02n/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r02n/pytest.log
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys; sys.path.insert(0, '.')
import torch, cvt_amd, bench
dev = torch.device("cuda", 0)
for d3 in (512, 256):
    g = torch.Generator(device=dev); g.manual_seed(3)
    feats = torch.randn((1 << 21, d3), generator=g, device=dev).clamp_(min=0)
    nb = feats.numel() * 4
    vmin, vdiff = cvt_amd.sq8_train(feats, l2norm=True)
    for wave in (0, 1, 0, 1):
        cvt_amd.set_tuning("sq8_encode_wave", wave)
        ms0 = bench._ev_ms(torch, lambda: cvt_amd.sq8_encode(vmin, vdiff, feats, l2norm=False), reps=5, warm=2)
        f2 = feats.clone()
        ms1 = bench._ev_ms(torch, lambda: cvt_amd.sq8_encode(vmin, vdiff, f2, l2norm=True), reps=5, warm=2)
        print("d=%d wave=%d: encode %.3f ms %.2f TB/s (4d in + d out); with normalisation written back %.3f ms, traffic %.2f TB/s (9d)" % (d3, wave, ms0, nb * 1.25 / ms0 / 1e9, ms1, nb * 2.25 / ms1 / 1e9))
PY
