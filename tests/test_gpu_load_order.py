"""The library loaded BEFORE torch in a fresh interpreter (cvt_amd.capi._torch_first): one HIP runtime, the device is seen,
and a device-pointer call on torch tensors works afterwards."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import sys
import numpy as np
import cvt_amd
assert "torch" not in sys.modules
cvt_amd.lib()
assert cvt_amd.capi.device_count() >= 1
import torch
assert torch.cuda.is_available()
rng = np.random.default_rng(3)
D, M, K, n = 32, 4, 256, 5000
books = rng.normal(size=(M, K, D // M)).astype(np.float32)
idx = cvt_amd.OpqIndex(np.zeros((1, D), np.float32), books, perm=np.arange(D, dtype=np.int32))
x = torch.from_numpy(rng.normal(size=(n, D)).astype(np.float32)).cuda()
_, codes = idx.encode(idx.rotate(x))
idx.add_codes(codes)
d, i = idx.search(x[:4].contiguous(), 5, rotate=True)
torch.cuda.synchronize()
assert i.shape == (4, 5) and int(i.min()) >= 0
print("LOAD_ORDER_OK")
"""


@pytest.mark.gpu
def test_library_before_torch():
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, "-c", CODE], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "LOAD_ORDER_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
