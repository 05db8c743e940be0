"""SpaceInterface::get_dist_func() of the host mirror (cvt_amd/host/hnswlib): the host distance functions of the three built-in
spaces against the checker's distances (which the goldens pin to the reference's own builds), bit for bit.  CPU only."""
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "cvt_amd", "bin", "dist_func_check")


def test_host_dist_funcs_match_the_checker(orc, tmp_path):
    assert os.path.exists(EXE), "host CLIs not built: __graft_entry__.build()"
    out = str(tmp_path / "d.bin")
    subprocess.run([EXE, out], check=True, stdout=subprocess.PIPE)
    blob = open(out, "rb").read()
    off, n = 0, 0
    while off < len(blob):
        metric, d = struct.unpack_from("<ii", blob, off); off += 8
        if metric == 2:
            a = np.frombuffer(blob, np.uint8, d, off); off += d
            b = np.frombuffer(blob, np.uint8, d, off); off += d
            (r,) = struct.unpack_from("<i", blob, off); off += 4
            assert r == int(orc.dist(2, 0, a, b)), (metric, d)
        else:
            a = np.frombuffer(blob, np.float32, d, off); off += 4 * d
            b = np.frombuffer(blob, np.float32, d, off); off += 4 * d
            r = np.frombuffer(blob, np.uint32, 1, off)[0]; off += 4
            if metric == 0:
                flavour = 4 if d % 4 == 0 else 0
            else:
                flavour = 8 if d % 16 == 0 else (4 if d % 4 == 0 else 0)
            exp = np.float32(orc.dist(metric, flavour, a, b))
            assert r == exp.view(np.uint32), (metric, d, r, exp.view(np.uint32))
        n += 1
    assert n == 3 * 9 * 4
