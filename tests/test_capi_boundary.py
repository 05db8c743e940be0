"""CPU: the C-ABI library builds, loads, exports every symbol include/cvtmi.h declares, and fails
loudly (no CPU fallback) when no HIP device is present.  No compute calls here."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "cvtmi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cvtmi_[a-z0-9_]+)\s*\(", text)))


def test_header_is_plain_c():
    src = os.path.join(ROOT, "tests", "_hdr_check.c")
    with open(src, "w") as f:
        f.write('#include "cvtmi.h"\nint main(void){ return CVTMI_OK; }\n')
    try:
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-fsyntax-only", src],
                       check=True)
    finally:
        os.remove(src)


def test_library_exports_every_declared_symbol():
    import cvt_amd
    lib = cvt_amd.lib()
    names = declared_symbols()
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.cvtmi_version() == 200


def test_no_oracle_in_product():
    """The product path may not import, link or call anything under oracle/."""
    for base, _, files in os.walk(os.path.join(ROOT, "cvt_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".hpp")) or fn == "Makefile":
                txt = open(os.path.join(base, fn), errors="replace").read()
                assert "cvt_oracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, fn
    out = subprocess.run(["ldd", os.path.join(ROOT, "cvt_amd", "lib", "libcvtmi.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out and "torch" not in out


def _has_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.mark.skipif(_has_gpu(), reason="checks the no-device behaviour")
def test_fails_loudly_without_device():
    import cvt_amd
    coarse = np.zeros((1, 8), dtype=np.float32)
    books = np.zeros((2, 256, 4), dtype=np.float32)
    with pytest.raises(cvt_amd.CvtmiError) as e:
        cvt_amd.OpqIndex(coarse, books)
    assert "hip" in str(e.value).lower()
    with pytest.raises(cvt_amd.CvtmiError):
        cvt_amd.sq8_train(np.zeros((4, 8), dtype=np.float32))


def test_argument_validation_is_reported():
    import cvt_amd
    lib = cvt_amd.lib()
    h = ctypes.c_void_p(0)
    rc = lib.cvtmi_opq_create(8, 1, 3, 256, None, None, None, None, ctypes.byref(h))
    assert rc == -1 and b"cvtmi_opq_create" in lib.cvtmi_last_error()
    rc = lib.cvtmi_flat_create(7, 128, ctypes.byref(h))
    assert rc == -1
    assert lib.cvtmi_opq_destroy(None) == 0


def test_tuning_hooks_validate_their_arguments():
    """cvtmi_set_tuning needs no device: unknown names and out-of-range values are refused with a message"""
    import cvt_amd
    lib = cvt_amd.lib()
    lib.cvtmi_last_error.restype = ctypes.c_char_p
    for name in (b"assign_variant", b"flat_variant"):
        assert lib.cvtmi_set_tuning(name, ctypes.c_int64(2)) == 0
        assert lib.cvtmi_set_tuning(name, ctypes.c_int64(0)) == 0
        assert lib.cvtmi_set_tuning(name, ctypes.c_int64(7)) != 0 and name in lib.cvtmi_last_error()
    assert lib.cvtmi_set_tuning(b"flat_u8_dbg", ctypes.c_int64(1)) != 0 and b"CVTMI_GF_DBG" in lib.cvtmi_last_error()   # shipping build: refused
    assert lib.cvtmi_set_tuning(b"sq8_wave_blocks", ctypes.c_int64(0)) != 0 and lib.cvtmi_set_tuning(b"sq8_wave_blocks", ctypes.c_int64(3)) == 0
    assert lib.cvtmi_set_tuning(b"flat_u8_mstream_min", ctypes.c_int64(0)) != 0 and b"flat_u8_mstream_min" in lib.cvtmi_last_error()
    assert lib.cvtmi_set_tuning(b"flat_u8_mstream_min", ctypes.c_int64(130)) != 0
    assert lib.cvtmi_set_tuning(b"flat_u8_mstream_min", ctypes.c_int64(1)) == 0
    assert lib.cvtmi_set_tuning(b"no_such_knob", ctypes.c_int64(1)) != 0 and b"no_such_knob" in lib.cvtmi_last_error()
    assert lib.cvtmi_set_tuning(None, ctypes.c_int64(1)) != 0


def test_hnsw_load_rejects_hostile_headers(golden):
    """cvtmi_hnsw_load validates the file before it touches the device (so this runs without one): header products that would
    wrap, element counts that would exhaust host memory, and upper-level links into nodes that do not have that level must all
    come back as CVTMI_EINVAL / ENOMEM -- never a crash, never an exception across the C ABI."""
    import cvt_amd
    lib = cvt_amd.lib()
    g = golden.hnsw
    metric, D = int(g["ip32_meta"][0]), int(g["ip32_meta"][1])
    blob = bytearray(g["ip32_index"].tobytes())

    def load(b):
        h = ctypes.c_void_p()
        buf = (ctypes.c_char * len(b)).from_buffer_copy(bytes(b))
        rc = lib.cvtmi_hnsw_load(buf, ctypes.c_int64(len(b)), ctypes.c_int(metric), ctypes.c_int(D), ctypes.byref(h))
        if rc == 0:
            lib.cvtmi_hnsw_destroy(h)
        return rc

    hdr = np.frombuffer(bytes(blob[:48]), np.uint64).copy()   # offsetLevel0, max_elements, cur_count, size_per, label_off, offsetData
    size_per = int(hdr[3])
    bad = bytearray(blob); bad[8:16] = np.uint64((1 << 64) // size_per + 1).tobytes()           # max_elements * size_per wraps to ~0
    bad[16:24] = np.uint64(1).tobytes()
    assert load(bad) == -1
    bad = bytearray(blob); bad[8:16] = np.uint64(1 << 54).tobytes(); bad[16:24] = np.uint64(1 << 54).tobytes()  # huge counts
    assert load(bad) in (-1, -2)
    bad = bytearray(blob); bad[48:52] = np.int32(-3).tobytes()                                   # negative maxlevel
    assert load(bad) == -1
    # an upper-level link that points at a node living on level 0 only
    max_elements, cur = int(hdr[1]), int(hdr[2])
    maxM = int(np.frombuffer(bytes(blob[56:64]), np.uint64)[0])
    p = 96 + max_elements * size_per
    levels = []
    for i in range(max_elements):
        sz = int(np.frombuffer(bytes(blob[p:p + 4]), np.uint32)[0]); levels.append((p + 4, sz)); p += 4 + sz
    flat = [i for i, (_, sz) in enumerate(levels[:cur]) if sz == 0]
    tall = [(i, off) for i, (off, sz) in enumerate(levels[:cur]) if sz > 0]
    if flat and tall:
        i, off = tall[0]
        cnt = int(np.frombuffer(bytes(blob[off:off + 4]), np.uint32)[0])
        if cnt > 0:
            bad = bytearray(blob); bad[off + 4:off + 8] = np.uint32(flat[0]).tobytes()
            assert load(bad) == -1
    # and the untouched file still fails only for lack of a device here (or loads, on a GPU box)
    assert load(blob) in (0, -3)
