import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _tuning_from_env():
    """CVTMI_TEST_TUNE="name=value,name=value": run the GPU suite under non-default tuning keys (they never change results)."""
    spec = os.environ.get("CVTMI_TEST_TUNE", "")
    if spec:
        import torch  # (first: the library must meet the HIP runtime torch has loaded, as in every GPU test)
        torch.cuda.is_available()
        import cvt_amd as amd
        for kv in spec.split(","):
            name, value = kv.split("=")
            amd.set_tuning(name.strip(), float(value))
    yield


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure, oracle/cvt_oracle.c)."""
    from oracle import binding
    binding.build()
    return binding.Oracle()


class Golden:
    """tests/golden/*.npz -- outputs of the reference itself (tests/golden/make_golden.py)."""

    def __init__(self):
        g = os.path.join(ROOT, "tests", "golden")
        self.dir = g
        z = np.load(os.path.join(g, "opq_golden.npz"))
        self.opq = {}
        for key in z.files:
            case, name = key.split("/")
            self.opq.setdefault(case, {})[name] = z[key]
        # the two real-data cases keep their raw inputs as the reference's own .bin files
        names = ["6231519245", "6231075428", "6230951284", "6230880830", "6231307582"]  # 5_feats_list.txt order
        vids = [np.fromfile(os.path.join(g, "opq_data", "db", n + "_feat.bin"), dtype=np.float32).reshape(-1, 128)
                for n in names]
        for tag, f in (("opq_real_q1", "6231519245_6_feat.bin"), ("opq_real_q9", "6231519245_feat.bin")):
            self.opq[tag]["db"] = np.concatenate(vids)
            self.opq[tag]["queries"] = np.fromfile(os.path.join(g, "opq_data", "query", f), dtype=np.float32).reshape(-1, 128)
        self.flat = dict(np.load(os.path.join(g, "flat_golden.npz")))
        self.sq8 = dict(np.load(os.path.join(g, "sq8_inputs.npz")))
        self.sq8_norm = dict(np.load(os.path.join(g, "sq8_norm_golden.npz")))   # MathUtil::L2NormArray / L2NormVec outputs
        self.hnsw = dict(np.load(os.path.join(g, "hnsw_golden.npz")))
        self.pca = dict(np.load(os.path.join(g, "pca_model.npz")))

    def video_of_row(self, case):
        rows = self.opq[case]["video_rows"]
        return np.concatenate([np.full(int(n), i, dtype=np.int32) for i, n in enumerate(rows)])


@pytest.fixture(scope="session")
def golden():
    return Golden()


SQ8_NORM_GROUPS = ["demo64", "cnn64", "cnn512", "cnn2048", "cnn37", "corners64", "clamp16"]
OPQ_CASES = ["opq_exh_m8", "opq_vec_m16", "opq_ivf", "opq_m1", "opq_real_q1", "opq_real_q9"]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
