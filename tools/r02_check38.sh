#!/bin/bash
cd /root/repo
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, time
sys.path.insert(0, '.')
os.environ["NQS"] = "1"
src = open("tools/opq_nq_sweep.py").read()
pre = src[:src.index("for nq in [int(v)")]
exec(pre)
for nq in (1, 4, 8, 32, 64, 256):
    q = qs[:nq].contiguous()
    for sp in (0, 64, 128, 256, 512, 1024):
        idx.set_param("splits", sp)
        try:
            for _ in range(3): idx.search(q, k)
        except Exception as e:
            print("nq", nq, "splits", sp, "error", str(e)[:80]); continue
        torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps): idx.search(q, k)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / reps * 1e3
        s = idx.last_scan()
        print("nq=%d splits(req)=%d -> qtile=%d splits=%d: scan %.3f ms, wall %.3f ms" % (nq, sp, s["qtile"], s["splits"], s["ms"], wall), flush=True)
PY
