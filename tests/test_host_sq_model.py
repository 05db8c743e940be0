"""CPU: the Int8Quan mirror reads the faiss "IxSQ" container -- what the reference's trainer writes (sq_train.cpp:103 write_index of
an IndexScalarQuantizer(d, QT_8bit, METRIC_L2)) and what its Int8Quan(model_path) loads (int8_quan.cc:14 faiss::read_index) -- so
that an existing model file drops in.  faiss is not installed here: the files below are assembled by hand from faiss 1.5.3's
published layout (index header, scalar-quantiser block, trained vector, stored codes)."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "cvt_amd", "bin")


def ixsq_bytes(vmin, vdiff, ntotal=0, qtype=0, metric=1, trained=1, codes=None):
    d = len(vmin)
    b = b"IxSQ" + struct.pack("<iqqqBi", d, ntotal, 1 << 20, 1 << 20, trained, metric)
    b += struct.pack("<iifQQ", qtype, 0, 0.0, d, d)
    tr = np.concatenate([vmin, vdiff]).astype("<f4")
    b += struct.pack("<Q", tr.size) + tr.tobytes()
    codes = np.zeros((ntotal, d), np.uint8) if codes is None else codes
    b += struct.pack("<Q", codes.size) + codes.tobytes()
    return b


def info(path):
    exe = os.path.join(BIN, "sq_model_info")
    assert os.path.exists(exe), "host CLIs not built: __graft_entry__.build()"
    r = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=60)
    if r.returncode != 0:
        return None, r.stdout + r.stderr
    out = {}
    for line in r.stdout.splitlines():
        key, _, val = line.partition(":")
        out[key] = val.split()
    return out, r.stdout


def as_bits(a):
    return ["%08x" % v for v in np.asarray(a, np.float32).view(np.uint32)]


@pytest.mark.parametrize("d,ntotal", [(64, 0), (512, 0), (7, 3), (64, 1000)])
def test_faiss_ixsq_container_is_read(tmp_path, d, ntotal):
    rng = np.random.default_rng(d + ntotal)
    vmin = rng.normal(size=d).astype(np.float32) * 0.1
    vdiff = np.abs(rng.normal(size=d)).astype(np.float32); vdiff[d // 2] = 0.0
    codes = rng.integers(0, 256, size=(ntotal, d), dtype=np.uint8)      # an index that also stores vectors: skipped
    p = tmp_path / "model.bin"
    p.write_bytes(ixsq_bytes(vmin, vdiff, ntotal=ntotal, codes=codes))
    got, txt = info(p)
    assert got is not None, txt
    assert got["format"] == ["faiss", "IxSQ"] and got["d"] == [str(d)]
    assert got["vmin_bits"] == as_bits(vmin) and got["vdiff_bits"] == as_bits(vdiff)
    # the plain form of the same model reads back the same arrays
    q = tmp_path / "plain.bin"
    q.write_bytes(struct.pack("<i", d) + vmin.tobytes() + vdiff.tobytes())
    got2, _ = info(q)
    assert got2["format"] == ["plain"] and got2["vmin_bits"] == got["vmin_bits"] and got2["vdiff_bits"] == got["vdiff_bits"]


def test_faiss_ixsq_rejections(tmp_path):
    """status() == false, as for a missing file (int8_quan.cc:8-12): other quantiser types (their codecs are not Int8Encode's
    in-tree formula), an untrained index, a truncated file; nothing is guessed"""
    d = 16
    vmin, vdiff = np.zeros(d, np.float32), np.ones(d, np.float32)
    good = ixsq_bytes(vmin, vdiff)
    for name, blob in (("qt_4bit", ixsq_bytes(vmin, vdiff, qtype=1)), ("untrained", ixsq_bytes(vmin, vdiff, trained=0)),
                       ("truncated", good[:len(good) - 40]), ("header_only", good[:20])):
        p = tmp_path / (name + ".bin")
        p.write_bytes(blob)
        got, txt = info(p)
        assert got is None, name
    got, txt = info(tmp_path / "does_not_exist.bin")
    assert got is None and "model file is not exists" in txt
