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
        od, ol = orc.hnsw_search_adc(blob, books, ocodes, rot(q), k2, ef2)
        for tables_in_lds in (0, 1):   # round 5: the tables are read from the scratch (default) or copied into LDS -- the same traversal
            amd.set_tuning("hnsw_adc_tables", tables_in_lds)
            try:
                d, lab = ix.search_adc(opq, q, k2, ef2)
            finally:
                amd.set_tuning("hnsw_adc_tables", 0)
            assert np.array_equal(lab, ol), (case, k2, ef2, tables_in_lds)
            assert np.array_equal(bits(d), bits(od)), (case, k2, ef2, tables_in_lds)
    # mismatched handles are refused
    small = mk(books)
    small.add_codes(codes[:10])
    with pytest.raises(amd.CvtmiError):
        ix.search_adc(small, q, 5, 10)
    # exact re-rank of the ADC result list (cvtmi_hnsw_search_adc_rerank): the ADC traversal's `rerank` best nodes, their fp32
    # distances in the reference's summation order (the oracle's orc_dist, default flavour), the k smallest, ties in ADC order
    labels_of = np.frombuffer(blob, np.uint8)  # labels of the golden graphs are read back through a plain search
    for k2, ef2, rr in ((5, 40, 40), (10, 64, 30), (1, 16, 8)):
        d, lab = ix.search_adc_rerank(opq, q, k2, ef2, rr)
        _, cand = orc.hnsw_search_adc(blob, books, ocodes, rot(q), rr, ef2)       # labels of the ADC list, ADC order
        lab2row = {int(l): r for r, l in enumerate(_graph_labels(blob, D))}
        for qi in range(q.shape[0]):
            rows = [lab2row[int(l)] for l in cand[qi] if l >= 0]
            ex = np.array([orc.dist(metric, 4, q[qi], x[r]) for r in rows], dtype=np.float32)
            order = np.argsort(ex, kind="stable")[:k2]
            want_l = [int(cand[qi][j]) for j in order]
            assert lab[qi, :len(want_l)].tolist() == want_l, (case, k2, ef2, rr, qi)
            assert np.array_equal(bits(d[qi, :len(want_l)]), bits(ex[order])), (case, k2, ef2, rr, qi)
    with pytest.raises(amd.CvtmiError):
        ix.search_adc_rerank(opq, q, 10, 40, 5)                                   # rerank < k is refused


def test_hnsw_concurrent_searches_on_one_handle(amd, orc, golden):
    """searchKnn is a read in the reference (hnswalg.h:688-728): six host threads search one graph handle at once -- plain, ADC and
    ADC + re-rank, host arrays in, each call on a scratch set leased from the handle -- while the OPQ handle the ADC calls borrow is
    searched too; every call returns exactly what it returns alone."""
    import threading
    from cvt_amd import synth
    g = golden.hnsw
    case = "ip128"
    metric, D, n, _, _, k, ef = (int(v) for v in g[case + "_meta"])
    blob = g[case + "_index"].tobytes()
    x = vectors_of(blob, D)
    R = synth.random_rotation(D, seed=3)
    xr = orc.rotate_fma(R, x)
    rng = np.random.default_rng(2)
    books = np.ascontiguousarray(np.stack([xr[rng.integers(0, n, 256), m * 8:(m + 1) * 8] for m in range(16)]))
    opq = amd.OpqIndex(np.zeros((1, D), np.float32), books, R=R)
    _, codes = opq.encode(opq.rotate(x)); opq.add_codes(codes)
    ix = amd.HnswIndex(blob, metric, D)
    q = g[case + "_q"]
    jobs = [("plain", lambda: ix.search(q, 10, 64)), ("adc", lambda: ix.search_adc(opq, q, 10, 64)),
            ("rerank", lambda: ix.search_adc_rerank(opq, q, 5, 64, 32)), ("opq", lambda: opq.search(q, 10)),
            ("plain1", lambda: ix.search(q[:3], 1, 16)), ("adc1", lambda: ix.search_adc(opq, q[:5], 3, 20))]
    want = {name: fn() for name, fn in jobs}
    bad, start = [], threading.Barrier(len(jobs))

    def worker(name, fn):
        start.wait()
        for _ in range(25):
            d, l = fn()
            if not (np.array_equal(np.asarray(l), np.asarray(want[name][1])) and np.array_equal(bits(np.asarray(d)), bits(np.asarray(want[name][0])))):
                bad.append(name)
                return
    ts = [threading.Thread(target=worker, args=j) for j in jobs]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not bad, bad
    ix.close(); opq.close()


def _graph_labels(blob, D):
    hdr = np.frombuffer(blob, np.uint64, 6, 0)
    max_elements, cur, size_per, label_off = int(hdr[1]), int(hdr[2]), int(hdr[3]), int(hdr[4])
    raw = np.frombuffer(blob, np.uint8, max_elements * size_per, 96).reshape(max_elements, size_per)
    return np.ascontiguousarray(raw[:cur, label_off:label_off + 8]).view(np.uint64).reshape(cur).astype(np.int64)


def test_hnsw_config5_scale(amd, orc):
    """BASELINE config 5 at scale: a 1 M-node graph (M = 16, efC = 40; built by the reference's own addPoint on all host
    cores -- its per-node locks make that legal, hnswalg.h:178,386,594), a batch of 10 000 queries.  fp32 traversal:
    labels and distance bits equal the reference's searchKnn on a sample; ADC traversal equals the oracle's on a sample;
    the exact re-rank recovers the fp32 graph's recall."""
    import torch
    from oracle import binding as ob
    if not os.path.exists(os.path.join(os.path.dirname(ob.__file__), "_ref", "libref_hnsw.so")):
        pytest.skip("oracle/_ref/libref_hnsw.so not built")
    from cvt_amd import synth
    n, D, nq = int(os.environ.get("CVT_C5_NODES", 1_000_000)), 128, 10_000
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(55)
    cen = torch.randn((4000, D), generator=g, device=dev)
    xd = cen[torch.randint(0, 4000, (n,), generator=g, device=dev)] + 0.3 * torch.randn((n, D), generator=g, device=dev)
    xd = xd / xd.norm(dim=1, keepdim=True)
    qd = xd[torch.randint(0, n, (nq,), generator=g, device=dev)] + 0.05 * torch.randn((nq, D), generator=g, device=dev)
    qd = (qd / qd.norm(dim=1, keepdim=True)).contiguous()
    x, q = xd.cpu().numpy(), qd.cpu().numpy()
    path = os.path.join(tempfile.gettempdir(), "cvt_test_c5.hnsw")
    rh = ob.RefHnsw()
    rh.build(0, x, path, 16, 40, threads=os.cpu_count() or 8)
    blob = open(path, "rb").read()
    ix = amd.HnswIndex(blob, 0, D)
    assert ix.ntotal == n
    d, lab = ix.search(qd, 10, 100)
    cs = 300
    rd, rl = rh.search(0, D, path, q[:cs], 10, 100)                                # the reference itself, same file
    os.remove(path)
    assert np.array_equal(lab[:cs].cpu().numpy(), rl) and np.array_equal(bits(d[:cs].cpu().numpy()), bits(rd))
    exact = torch.empty(nq, dtype=torch.int64, device=dev)
    for a in range(0, nq, 1000):
        exact[a:a + 1000] = torch.argmax(qd[a:a + 1000] @ xd.T, dim=1)
    labels = torch.from_numpy(_graph_labels(blob, D)).to(dev)                       # parallel build: label != internal id
    rec_fp32 = float((lab[:, 0] == exact).float().mean().item())
    assert rec_fp32 > 0.5, rec_fp32                                               # a property of this quickly built graph, not of the search
    # over OPQ codes (16 bytes per node), internal-id order
    R = synth.random_rotation(D, seed=3)
    xi = torch.from_numpy(vectors_of(blob, D)).to(dev)
    tmp = amd.OpqIndex(np.zeros((1, D), np.float32), np.zeros((16, 256, D // 16), np.float32), R=R)
    xr = tmp.rotate(xi)
    _, books = amd.opq_train(xr[:100_000].contiguous(), 1, 16, 256, 6, 1)
    books = books.cpu().numpy()
    opq = amd.OpqIndex(np.zeros((1, D), np.float32), books, R=R)
    _, codes = opq.encode(xr)
    opq.add_codes(codes)
    da, la = ix.search_adc(opq, qd, 10, 100)
    cs2 = 60
    od, ol = orc.hnsw_search_adc(blob, books, codes.cpu().numpy(), orc.rotate_fma(R, q[:cs2]), 10, 100)
    assert np.array_equal(la[:cs2].cpu().numpy(), ol) and np.array_equal(bits(da[:cs2].cpu().numpy()), bits(od))
    dr, lr = ix.search_adc_rerank(opq, qd, 10, 100, 100)
    rec_adc = float((la[:, 0] == exact).float().mean().item())
    rec_rr = float((lr[:, 0] == exact).float().mean().item())
    assert rec_rr > rec_adc, (rec_fp32, rec_adc, rec_rr)          # how far it recovers is a property of this quickly built graph
    print("config 5, %d nodes, %d queries: recall@1 fp32 %.3f, ADC %.3f, ADC + re-rank %.3f" % (n, nq, rec_fp32, rec_adc, rec_rr))
    del labels
