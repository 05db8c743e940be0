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
    assert lib.cvtmi_set_tuning(b"no_such_knob", ctypes.c_int64(1)) != 0 and b"no_such_knob" in lib.cvtmi_last_error()
    assert lib.cvtmi_set_tuning(None, ctypes.c_int64(1)) != 0
