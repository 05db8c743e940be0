import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, cvt_amd
from oracle import binding as ob
ob.build(); orc = ob.Oracle()
rng = np.random.default_rng(3)
bad = 0
for (D, n, nq, k) in [(128, 52, 104, 80), (128, 52, 104, 40), (128, 52, 4, 80), (128, 52, 64, 80), (128, 52, 65, 80), (128, 100, 104, 80), (128, 30, 33, 24), (512, 20, 40, 80), (128, 52, 104, 10)]:
    xu = rng.integers(0, 256, size=(n, D), dtype=np.uint8); qu = rng.integers(0, 256, size=(nq, D), dtype=np.uint8)
    xu[n // 2] = xu[0]; qu[0] = xu[0]
    for v in (0, 1):
        cvt_amd.set_tuning("flat_variant", v)
        fi = cvt_amd.FlatIndex(2, D); fi.add(xu)
        d, i = fi.search(qu, k)
        _, odi, oi = orc.flat_search(2, xu, qu, k)
        ok = np.array_equal(i, oi) and np.array_equal(d, odi)
        if not ok:
            bad += 1
            w = np.argwhere((i != oi) | (d != odi))
            a, b = int(w[0][0]), int(w[0][1])
            print("MISMATCH", (D, n, nq, k), "variant", v, "diffs", len(w), "first", (a, b), "gpu", i[a, max(0, b - 2):b + 3].tolist(), "orc", oi[a, max(0, b - 2):b + 3].tolist(),
                  "gpu d", d[a, max(0, b - 2):b + 3].tolist(), "orc d", odi[a, max(0, b - 2):b + 3].tolist(), d.dtype, odi.dtype, flush=True)
        else:
            print("ok", (D, n, nq, k), "variant", v, flush=True)
        fi.close()
cvt_amd.set_tuning("flat_variant", 0)
print("bad", bad)
