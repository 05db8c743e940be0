"""CPU: the PCA checker against numpy float64, and the host mirror's OpenCV-YAML model reader (cvtk::PCAUtils::
loadModel, pca_utils.cc:16-23) through `pca_project --info` (no GPU involved)."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "cvt_amd", "bin")


def test_checker_flavours(orc, golden):
    g = golden.pca
    rng = np.random.default_rng(3)
    x = np.maximum(rng.normal(size=(40, 1024)), 0).astype(np.float32)
    t = (x - g["mean"]).astype(np.float32).astype(np.float64)
    want = (t @ g["vectors"].astype(np.float64).T)
    y0 = orc.pca_project(g["mean"], g["vectors"], x, False, flavour=0)
    y1 = orc.pca_project(g["mean"], g["vectors"], x, False, flavour=1)
    assert np.abs(y0 - want).max() <= np.abs(want).max() * 1e-7      # double accumulation, one rounding
    assert np.abs(y1 - want).max() <= np.abs(want).max() * 2e-5      # fp32 chain of 1024 terms
    n0 = orc.pca_project(g["mean"], g["vectors"], x, True, flavour=0)
    assert np.all(np.abs(np.linalg.norm(n0.astype(np.float64), axis=1) - 1) < 1e-6)
    assert np.abs(n0 - want / np.linalg.norm(want, axis=1, keepdims=True)).max() < 1e-6


def test_model_reader(tmp_path, golden):
    assert os.path.exists(os.path.join(BIN, "pca_project")), "host CLIs not built: __graft_entry__.build()"
    from test_gpu_pca import write_opencv_yaml
    g = golden.pca
    write_opencv_yaml(tmp_path / "model.yml", g)
    r = subprocess.run([os.path.join(BIN, "pca_project"), str(tmp_path / "model.yml"), "--info"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    sv = float(g["vectors"].astype(np.float64).sum()); sm = float(g["mean"].astype(np.float64).sum())
    assert r.stdout.strip() == "vectors 128 x 1024 sum %.9g; values 128 x 1; mean 1 x 1024 sum %.9g" % (sv, sm)
    # malformed files fail loudly
    (tmp_path / "bad.yml").write_text("%YAML:1.0\n---\nvectors: !!opencv-matrix\n   rows: 2\n   cols: 2\n   dt: f\n   data: [ 1., 2., 3. ]\n")
    r = subprocess.run([os.path.join(BIN, "pca_project"), str(tmp_path / "bad.yml"), "--info"], capture_output=True, text=True)
    assert r.returncode == 1 and "too few values" in r.stdout
