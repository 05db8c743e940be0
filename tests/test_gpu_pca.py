"""GPU: PCA projection + L2 normalisation (cvtk::PCAUtils::reduceDim, pca_utils.cc:25-35) through
the C ABI against the CPU checker.  The reference's projection is an OpenCV call (absent here): PARITY UNPINNED.
Two comparisons: bit for bit against the kernel's specification (k-ascending fp32 fmaf chain, checker flavour 1),
and within 2e-6 absolute on unit-norm rows of OpenCV's own gemm arithmetic (double accumulators, flavour 0)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "cvt_amd", "bin")
TOL = 2e-6  # absolute, on rows of unit norm


@pytest.fixture(scope="module")
def amd():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import cvt_amd
    cvt_amd.lib()  # raises if the HIP library is missing: there is no fallback
    return cvt_amd


def cnn_like(rng, n, d):
    x = np.maximum(rng.normal(size=(n, d)), 0).astype(np.float32) * rng.gamma(2.0, 1.0, size=(1, d)).astype(np.float32)
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def ulp_diff(a, b):
    ia = a.view(np.int32).astype(np.int64); ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia); ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


def test_reference_model_projection(amd, golden, orc):
    """the reference's own 1024 -> 128 model (tests/golden/pca_model.npz)"""
    g = golden.pca
    rng = np.random.default_rng(0x9CA)
    x = cnn_like(rng, 300, 1024)
    x[7] = g["mean"][0]            # projects to exactly zero: the 1e-12 clamp
    raw = amd.pca_project(g["mean"], g["vectors"], x, l2norm=False)
    spec = orc.pca_project(g["mean"], g["vectors"], x, False, flavour=1)
    assert np.array_equal(raw.view(np.uint32), spec.view(np.uint32))
    assert np.all(raw[7] == 0)
    y = amd.pca_project(g["mean"], g["vectors"], x, l2norm=True)
    ys = orc.pca_project(g["mean"], g["vectors"], x, True, flavour=1)
    assert ulp_diff(y, ys).max() <= 1 and np.mean(y.view(np.uint32) == ys.view(np.uint32)) > 0.999
    ycv = orc.pca_project(g["mean"], g["vectors"], x, True, flavour=0)
    assert np.abs(y.astype(np.float64) - ycv).max() <= TOL
    assert np.all(y[7] == 0) and np.all(np.abs(np.linalg.norm(np.delete(y, 7, 0).astype(np.float64), axis=1) - 1) < 1e-6)


@pytest.mark.parametrize("din,dout,n", [(20, 7, 129), (64, 32, 1), (100, 33, 257), (2048, 256, 300), (512, 200, 128), (36, 1, 50)])
def test_shapes(amd, orc, din, dout, n):
    rng = np.random.default_rng(din * 1000 + dout)
    e = np.linalg.qr(rng.normal(size=(din, din)))[0][:dout].astype(np.float32)
    mean = rng.normal(0.02, 0.01, size=din).astype(np.float32)
    x = cnn_like(rng, n, din)
    for l2 in (False, True):
        y = amd.pca_project(mean, e, x, l2norm=l2)
        spec = orc.pca_project(mean, e, x, l2, flavour=1)
        if l2:
            assert ulp_diff(y, spec).max() <= 1
        else:
            assert np.array_equal(y.view(np.uint32), spec.view(np.uint32))
        assert np.abs(y.astype(np.float64) - orc.pca_project(mean, e, x, l2, flavour=0)).max() <= TOL


def test_device_entry_and_empty(amd, golden, orc):
    import torch
    g = golden.pca
    x = cnn_like(np.random.default_rng(5), 1000, 1024)
    dev = "cuda:0"
    y = amd.pca_project(torch.from_numpy(g["mean"]).to(dev), torch.from_numpy(g["vectors"]).to(dev), torch.from_numpy(x).to(dev), l2norm=True)
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy().view(np.uint32), amd.pca_project(g["mean"], g["vectors"], x, l2norm=True).view(np.uint32))
    assert amd.pca_project(g["mean"], g["vectors"], np.zeros((0, 1024), np.float32)).shape == (0, 128)
    with pytest.raises(amd.CvtmiError):
        amd.pca_project(np.zeros(10, np.float32), np.zeros((4, 10), np.float32), np.zeros((3, 10), np.float32))  # din % 4 != 0


def test_pca_project_cli(tmp_path, amd, golden):
    """PCAUtils mirror: OpenCV FileStorage YAML model -> loadModel -> reduceDim, through the CLI"""
    assert os.path.exists(os.path.join(BIN, "pca_project")), "host CLIs not built: __graft_entry__.build()"
    g = golden.pca
    write_opencv_yaml(tmp_path / "model.yml", g)
    x = cnn_like(np.random.default_rng(11), 9, 1024)
    with open(tmp_path / "feats.txt", "w") as f:
        for i, row in enumerate(x):
            f.write("frame_%d," % i + ",".join("%.9g" % v for v in row) + "\n")
    r = subprocess.run([os.path.join(BIN, "pca_project"), "model.yml", "feats.txt", "out.txt"], cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    want = amd.pca_project(g["mean"], g["vectors"], x, l2norm=True)
    lines = (tmp_path / "out.txt").read_text().splitlines()
    assert [ln.split()[0] for ln in lines] == ["frame_%d" % i for i in range(9)]
    got = np.array([[float(v) for v in ln.split()[1:]] for ln in lines], np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def write_opencv_yaml(path, g):
    """what cv::FileStorage writes for a cv::PCA (the layout of pca_train_project/model/*.yml)"""
    def node(name, a):
        vals = ["%.8e" % v for v in a.ravel()]
        body = ",\n       ".join(", ".join(vals[i:i + 4]) for i in range(0, len(vals), 4))
        return "%s: !!opencv-matrix\n   rows: %d\n   cols: %d\n   dt: f\n   data: [ %s ]\n" % (name, a.shape[0], a.shape[1], body)
    with open(path, "w") as f:
        f.write("%YAML:1.0\n---\nname: PCA\n" + node("vectors", g["vectors"]) + node("values", g["values"]) + node("mean", g["mean"]))
