"""HNSW search on the GPU: graphs built and saved by the reference, searched through the C ABI;
labels and distance bits must equal the reference's own searchKnn answers (golden) and the oracle's."""
import os
import tempfile

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu
CASES = ["ip32", "l2f16", "ip20", "l2f7", "ip128"]


@pytest.fixture(scope="module")
def amd():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import cvt_amd
    cvt_amd.lib()  # raises if the HIP library is missing: there is no fallback
    return cvt_amd


@pytest.mark.parametrize("case", CASES)
def test_hnsw_golden(amd, orc, golden, case):
    g = golden.hnsw
    metric, D, n, M, efc, k, ef = (int(v) for v in g[case + "_meta"])
    ix = amd.HnswIndex(g[case + "_index"].tobytes(), metric, D)
    assert ix.ntotal == n
    d, lab = ix.search(g[case + "_q"], k, ef)
    assert np.array_equal(lab, g[case + "_l"]), case
    assert np.array_equal(bits(d), bits(g[case + "_d"])), case
    # other (k, ef) pairs against the oracle: ef < k, ef = 1, k = 1, large ef
    for k2, ef2 in ((1, 1), (20, 5), (3, 64), (50, 300), (64, 1024)):
        od, ol = orc.hnsw_search(g[case + "_index"].tobytes(), metric, D, g[case + "_q"], k2, ef2)
        d2, l2 = ix.search(g[case + "_q"], k2, ef2)
        assert np.array_equal(l2, ol) and np.array_equal(bits(d2), bits(od)), (case, k2, ef2)
    ix.close()


def test_hnsw_load_rejects_garbage(amd, golden):
    g = golden.hnsw
    blob = g["ip32_index"].tobytes()
    with pytest.raises(amd.CvtmiError):
        amd.HnswIndex(blob[:50], 0, 32)
    with pytest.raises(amd.CvtmiError):
        amd.HnswIndex(blob, 0, 33)            # header does not describe 33-d vectors
    with pytest.raises(amd.CvtmiError):
        amd.HnswIndex(blob[:len(blob) // 2], 0, 32)
    ix = amd.HnswIndex(blob, 0, 32)
    with pytest.raises(amd.CvtmiError):
        ix.search(g["ip32_q"], 2000, 10)      # k out of range is refused, not truncated


def test_hnsw_larger_graph_device_pointers(amd, orc):
    """A 30 K-node, M = 16 graph built on the spot by the reference (oracle/_ref/libref_hnsw.so travels with the
    snapshot); 2000 queries on device pointers, ef = 200: every answer equals the oracle's."""
    import torch
    from oracle import binding as ob
    if not os.path.exists(os.path.join(os.path.dirname(ob.__file__), "_ref", "libref_hnsw.so")):
        pytest.skip("oracle/_ref/libref_hnsw.so not built")
    rng = np.random.default_rng(77)
    n, D = 30_000, 64
    cen = rng.normal(size=(200, D)).astype(np.float32)
    x = cen[rng.integers(0, 200, n)] + 0.5 * rng.normal(size=(n, D)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    q = x[rng.integers(0, n, 2000)] + 0.1 * rng.normal(size=(2000, D)).astype(np.float32)
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    path = os.path.join(tempfile.gettempdir(), "cvt_test_big.hnsw")
    ob.RefHnsw().build(0, x, path, 16, 100)
    blob = open(path, "rb").read()
    os.remove(path)
    ix = amd.HnswIndex(blob, 0, D)
    qd = torch.from_numpy(q).cuda()
    d, lab = ix.search(qd, 10, 200)
    od, ol = orc.hnsw_search(blob, 0, D, q, 10, 200)
    assert np.array_equal(lab.cpu().numpy(), ol) and np.array_equal(bits(d.cpu().numpy()), bits(od))
    # recall against the exact answer, for the record (a property of the graph, identical on CPU and GPU)
    exact = np.argmax(q @ x.T, axis=1)
    assert (lab[:, 0].cpu().numpy() == exact).mean() > 0.9


def vectors_of(blob, D):
    """the fp32 vectors stored in a saveIndex file, internal-id order"""
    hdr = np.frombuffer(blob, np.uint64, 6, 0)
    max_elements, cur, size_per, off_data = int(hdr[1]), int(hdr[2]), int(hdr[3]), int(hdr[5])
    raw = np.frombuffer(blob, np.uint8, max_elements * size_per, 96).reshape(max_elements, size_per)
    return np.ascontiguousarray(raw[:cur, off_data:off_data + 4 * D]).view(np.float32).reshape(cur, D)


@pytest.mark.parametrize("case,M,K", [("ip128", 16, 256), ("ip32", 8, 64), ("l2f16", 4, 256)])
def test_hnsw_over_opq_codes(amd, orc, golden, case, M, K):
    """BASELINE config 5: the reference-built graph traversed with ADC distances over PQ codes (dense rotation,
    codes from the GPU encoder).  Specification = the oracle's traversal with the oracle's ADC arithmetic."""
    from cvt_amd import synth
    g = golden.hnsw
    metric, D, n, _, _, k, ef = (int(v) for v in g[case + "_meta"])
    blob = g[case + "_index"].tobytes()
    x = vectors_of(blob, D)
    if D % 32 == 0:
        R = synth.random_rotation(D, seed=3)                  # dense rotation: MFMA GEMM (built for D = 32 .. 128)
        mk = lambda books: amd.OpqIndex(np.zeros((1, D), np.float32), books, R=R)
        rot = lambda v: orc.rotate_fma(R, v)
    else:
        perm = synth.random_permutation(D, seed=3)            # the reference's own "rotation"
        mk = lambda books: amd.OpqIndex(np.zeros((1, D), np.float32), books, perm=perm)
        rot = lambda v: orc.reorder(perm, v)
    xr = rot(x)
    rng = np.random.default_rng(1)
    step = D // M
    books = np.ascontiguousarray(np.stack([xr[rng.integers(0, n, K), m * step:(m + 1) * step] for m in range(M)]))
    opq = mk(books)
    _, codes = opq.encode(opq.rotate(x))
    opq.add_codes(codes)
    _, ocodes = orc.pq_encode(xr, np.zeros((1, D), np.float32), books)
    assert np.array_equal(codes, ocodes)
    ix = amd.HnswIndex(blob, metric, D)
    q = g[case + "_q"]
    for k2, ef2 in ((k, ef), (10, 40), (1, 1)):
        d, lab = ix.search_adc(opq, q, k2, ef2)
        od, ol = orc.hnsw_search_adc(blob, books, ocodes, rot(q), k2, ef2)
        assert np.array_equal(lab, ol), (case, k2, ef2)
        assert np.array_equal(bits(d), bits(od)), (case, k2, ef2)
    # mismatched handles are refused
    small = mk(books)
    small.add_codes(codes[:10])
    with pytest.raises(amd.CvtmiError):
        ix.search_adc(small, q, 5, 10)
