"""GPU parity tests of the OPQ path through the C ABI: HIP kernels vs the golden vectors (outputs of
the reference itself) and vs the CPU oracle on seeded inputs.  Codes / list ids / LUT entries / ADC
distances / top-k ids are all required to be BIT-EXACT (stronger than the 1e-4 relative bound of the
north star)."""
import numpy as np
import pytest

from conftest import OPQ_CASES, bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import cvt_amd
    cvt_amd.lib()  # raises if the HIP library is missing: there is no fallback
    return cvt_amd


def make_index(amd, g, rotation="perm"):
    D = g["coarse"].shape[1]
    if rotation == "perm":
        return amd.OpqIndex(g["coarse"], g["books"], perm=g["perm"])
    R = np.zeros((D, D), dtype=np.float32)
    R[np.arange(D), g["perm"]] = 1.0
    return amd.OpqIndex(g["coarse"], g["books"], R=R)


@pytest.mark.parametrize("case", OPQ_CASES)
def test_golden_rotate_encode_query(amd, golden, case):
    g = golden.opq[case]
    idx = make_index(amd, g)
    # rotation (IVFOPQ::reorder)
    assert np.array_equal(idx.rotate(g["queries"]), g["q_rot"])
    db_rot = idx.rotate(g["db"])
    assert np.array_equal(db_rot, g["db_rot"])
    # encode (IVFOPQ::Add)
    lists, codes = idx.encode(db_rot)
    idx.add_codes(codes, list_id=lists, video_id=golden.video_of_row(case))
    off, vid, ccodes = idx.get_entries()
    assert np.array_equal(off, g["list_off"])
    assert np.array_equal(ccodes, g["codes"]), "PQ codes differ from the reference"
    assert np.array_equal(vid, g["video_id"])
    # query (IVFOPQ::QueryThrehold) incl. rotation of the raw query file contents
    ms = idx.query_video(g["queries"], int(g["nk"]), len(g["video_rows"]), rotate=True)
    assert np.array_equal(bits(ms), bits(g["match_score"])), "match scores differ from the reference"


@pytest.mark.parametrize("case", ["opq_exh_m8", "opq_vec_m16", "opq_real_q9"])
def test_golden_rotation_as_mfma_gemm(amd, golden, case):
    """The permutation fed to the MFMA fp32 GEMM as a 0/1 matrix reproduces the reference's gather."""
    g = golden.opq[case]
    idx = make_index(amd, g, rotation="R")
    assert np.array_equal(idx.rotate(g["queries"]), g["q_rot"])
    assert np.array_equal(idx.rotate(g["db"]), g["db_rot"])


def test_golden_exhaustive_topk(amd, golden, orc):
    """North-star form on the reference's numbers: one vector per video, no clamp."""
    g = golden.opq["opq_vec_m16"]
    idx = make_index(amd, g)
    idx.add_codes(g["codes"])
    ms = g["match_score"]
    for variant in (6, 5, 4, 3, 1, 2, 0):
        for qt in (0, 1, 2, 4, 8):
            for splits in (0, 1, 3, 8):
                idx.set_param("scan_variant", variant); idx.set_param("qtile", qt); idx.set_param("splits", splits)
                d, i = idx.search(g["queries"], 100, rotate=True)
                for f in range(ms.shape[0]):
                    order = np.lexsort((np.arange(ms.shape[1]), ms[f]))[:100]
                    assert np.array_equal(i[f], order), (variant, qt, splits, f)
                    assert np.array_equal(bits(d[f]), bits(ms[f][order])), (variant, qt, splits, f)
    assert list(i[0][:3]) == [7, 100, 200]  # exact duplicates: ties resolved by id


def test_golden_lut(amd, golden, orc):
    for case in ("opq_m1", "opq_vec_m16", "opq_ivf"):
        g = golden.opq[case]
        idx = make_index(amd, g)
        nq = g["q_rot"].shape[0]
        lists = (np.arange(nq) % g["coarse"].shape[0]).astype(np.int32)
        lut = idx.lut(g["q_rot"], lists)
        for f in range(nq):
            ref = orc.lut(g["q_rot"][f], g["coarse"][lists[f]], g["books"])
            assert np.array_equal(bits(lut[f]), bits(ref)), (case, f)
    g = golden.opq["opq_m1"]  # M = 1: the reference's scores are single LUT entries
    idx = make_index(amd, g)
    lut = idx.lut(g["q_rot"])
    for f in range(g["q_rot"].shape[0]):
        assert np.array_equal(bits(lut[f, 0][g["codes"][:, 0]]), bits(g["match_score"][f]))


def synth_model(rng, D, M, K, n_train=4000, scale=1.0):
    x = (rng.normal(size=(n_train, D)) * scale).astype(np.float32)
    step = D // M
    books = np.stack([x[rng.integers(0, n_train, K), m * step:(m + 1) * step] for m in range(M)]).astype(np.float32)
    return np.ascontiguousarray(books)


@pytest.mark.parametrize("D,M,K,coarseK", [(128, 16, 256, 1), (128, 8, 256, 1), (64, 16, 256, 5), (32, 4, 200, 40),
                                           (128, 4, 256, 1), (96, 2, 17, 3), (24, 3, 256, 1)])
def test_encode_parity_seeded(amd, orc, D, M, K, coarseK):
    rng = np.random.default_rng(D * 1000 + M)
    books = synth_model(rng, D, M, K)
    coarse = np.zeros((1, D), np.float32) if coarseK == 1 else rng.normal(size=(coarseK, D)).astype(np.float32)
    n = 3000 + 37
    x = rng.normal(size=(n, D)).astype(np.float32)
    x[5] = np.nan          # no centroid can claim it: list -1, codes 255 (IVFOPQ.cpp:144,161)
    x[6, 3] = np.inf
    x[7] = books[:, 9, :].reshape(-1) + (coarse[0] if coarseK == 1 else 0)  # exact codeword
    idx = amd.OpqIndex(coarse, books)
    lists, codes = idx.encode(x)
    ol, oc = orc.pq_encode(x, coarse, books)
    assert np.array_equal(lists, ol)
    assert np.array_equal(codes, oc)
    assert lists[5] == -1 and np.all(codes[5] == 255)
    if K == 256 and D // M in (8, 16) and M <= 16:   # the matrix-core filter kernel is chosen from 8192 rows up: force it
        idx.set_param("encode_variant", 2)
        l2, c2 = idx.encode(x)
        assert np.array_equal(l2, ol) and np.array_equal(c2, oc)
        for nn in (1, 31, 33, 1025):
            assert np.array_equal(idx.encode(x[:nn])[1], oc[:nn])
        idx.set_param("encode_variant", 0)
    # ragged sizes around the tile width, and the empty input
    for nn in (0, 1, 255, 1025):
        l2, c2 = idx.encode(x[:nn])
        assert np.array_equal(c2, oc[:nn]) and np.array_equal(l2, ol[:nn])


@pytest.mark.parametrize("D,M", [(128, 16), (64, 8)])
def test_encode_one_video_at_a_time(amd, orc, D, M):
    """The reference builds its index one video at a time: IVFOPQ::Add of a few hundred frames over 8192 lists (opq/src/IVFOPQ.cpp:135-163).
    Round 5: such calls take the matrix-core PQ encode at every row count (the VALU kernel cost 0.7 ms per call) and a coarse assignment
    whose centroid range is cut over workgroups (one thread per row walking 8192 centroids cost 7.6 ms per call).  Lists and codes against
    the oracle for 1 ... 4095 frames, device and host pointers, with duplicate centroids (first minimum) and a NaN frame."""
    import torch
    K, L = 256, 1500
    rng = np.random.default_rng(D + M)
    books = synth_model(rng, D, M, K)
    coarse = rng.normal(size=(L, D)).astype(np.float32)
    coarse[700] = coarse[20]; coarse[1499] = coarse[20]          # strict '<': list 20 wins every tie
    idx = amd.OpqIndex(coarse, books)
    for n in (1, 9, 300, 1000, 4095):
        x = (coarse[rng.integers(0, L, n)] + 0.3 * rng.normal(size=(n, D))).astype(np.float32)
        x[0] = coarse[20]
        if n > 5:
            x[5] = np.nan
        ol, oc = orc.pq_encode(x, coarse, books)
        for dev in (False, True):
            lists, codes = idx.encode(torch.from_numpy(x).cuda() if dev else x)
            if dev:
                lists, codes = lists.cpu().numpy(), codes.cpu().numpy()
            assert np.array_equal(lists, ol) and np.array_equal(codes, oc), (n, dev)
        assert ol[0] == 20


def test_encode_tie_takes_first_minimum(amd, orc):
    D, M, K = 32, 4, 256
    rng = np.random.default_rng(1)
    books = synth_model(rng, D, M, K)
    books[:, 200] = books[:, 40]   # duplicate codewords: strict '<' keeps index 40
    x = np.tile(books[:, 40].reshape(1, -1), (300, 1)).astype(np.float32)
    idx = amd.OpqIndex(np.zeros((1, D), np.float32), books)
    _, codes = idx.encode(x)
    assert np.all(codes == 40)


@pytest.mark.parametrize("M", [16, 8])
def test_encode_matrix_core_filter_hard_cases(amd, orc, M):
    """the fp32 matrix-core filter must hand every pair it cannot separate to the exact chain: duplicate codewords,
    rows equidistant from two codewords, codewords one ulp apart, huge / tiny / non-finite magnitudes, a codebook
    with an infinite entry -- and agree with the reference's codes on clustered data at its natural tie rate"""
    D, K = 128, 256
    step = D // M
    rng = np.random.default_rng(77 + M)
    cen = rng.normal(size=(K, D)).astype(np.float32)
    books = np.ascontiguousarray(np.stack([cen[:, m * step:(m + 1) * step] for m in range(M)]))
    books[:, 200] = books[:, 40]                                         # exact duplicates
    books[:, 201] = np.nextafter(books[:, 41], np.float32(np.inf))       # one ulp apart
    books[0, 17] = books[0, 3] + np.float32(2.0)                         # for the midpoint rows
    n = 20000
    x = (cen[rng.integers(0, K, n)] + 0.3 * rng.normal(size=(n, D))).astype(np.float32)
    x[:300] = cen[40]                                                    # on a duplicated codeword
    x[300:600] = cen[41]
    x[600:700, :step] = (books[0, 3] + books[0, 17]) / 2                 # equidistant in sub-space 0
    x[700:720] *= np.float32(3e4); x[720:740] *= np.float32(1e-30)       # |d| past the start value / denormal products
    x[740, 5] = np.inf; x[741, 77] = -np.inf; x[742] = np.nan; x[743, 9] = np.nan
    x[744:760] = 0
    for i, f in enumerate((1 - 1e-3, 1 - 1e-6, 1 - 1e-7, 1.0, 1 + 1e-7, 1 + 1e-6, 1 + 1e-3, 0.4, 0.6)):   # |x|^2 around the start value 2^32
        v = rng.normal(size=D); x[760 + i] = (v / np.linalg.norm(v) * np.sqrt(2.0 ** 32 * f)).astype(np.float32)
    zero = np.zeros((1, D), np.float32)
    for bk in (books, np.where(np.arange(K)[None, :, None] == 99, np.float32(np.inf), books).astype(np.float32)):
        bk = np.ascontiguousarray(bk)
        idx = amd.OpqIndex(zero, bk)
        idx.set_param("encode_variant", 2)
        lists, codes = idx.encode(x)
        ol, oc = orc.pq_encode(x, zero, bk)
        bad = np.argwhere(codes != oc)
        assert bad.size == 0, (bad[:10], codes[bad[:10, 0], bad[:10, 1]], oc[bad[:10, 0], bad[:10, 1]])
        assert np.array_equal(lists, ol) and set(np.unique(ol)) == {-1, 0}   # the kernel's own single-list assignment
        idx.set_param("encode_variant", 1)
        l1, c1 = idx.encode(x)
        assert np.array_equal(c1, oc) and np.array_equal(l1, ol)


@pytest.mark.parametrize("M", [16, 8, 4])
@pytest.mark.parametrize("k", [1, 10, 100, 128])
def test_search_parity_seeded(amd, orc, M, k):
    D, K = 128, 256
    rng = np.random.default_rng(M * 7 + k)
    books = synth_model(rng, D, M, K, scale=0.1)
    n = 20000 + 13
    codes = rng.integers(0, K, size=(n, M), dtype=np.uint8)
    codes[100] = codes[50]; codes[15000] = codes[50]           # exact ties straddling splits
    q = (rng.normal(size=(9, D)) * 0.1).astype(np.float32)
    R = None
    idx = amd.OpqIndex(np.zeros((1, D), np.float32), books, R=R)
    idx.add_codes(codes[:7000]); idx.add_codes(codes[7000:])   # two appends
    assert idx.ntotal == n
    od, oi = orc.adc_search(q, books, codes, k)
    for variant, qt, splits in ((6, 0, 0), (6, 0, 1), (6, 0, 8), (6, 0, 3), (5, 0, 0), (5, 0, 1), (5, 0, 8), (5, 0, 3), (4, 0, 0), (4, 0, 1), (4, 0, 8), (3, 0, 3), (3, 0, 1), (1, 0, 0), (1, 4, 1), (1, 4, 8), (2, 4, 3), (2, 0, 0), (0, 0, 0), (0, 1, 1), (0, 2, 5), (0, 4, 8),
                                (0, 4, 16), (0, 1, 64)):
        idx.set_param("scan_variant", variant); idx.set_param("qtile", qt); idx.set_param("splits", splits)
        d, i = idx.search(q, k, rotate=False)
        assert np.array_equal(i, oi), (variant, qt, splits)
        assert np.array_equal(bits(d), bits(od)), (variant, qt, splits)


def test_search_two_region_tail(amd, orc):
    """Round 5: when the query groups past the last full round of workgroups would leave most CUs empty, adc_scan16q cuts THOSE groups
    into row splits (plan_scan's two-region plan: 4096 + 160 queries -> 20 groups in 4 splits here).  Same lists as whole groups and as
    the oracle, ties across the split boundaries included."""
    import torch
    D, M, K, k = 128, 16, 256, 20
    rng = np.random.default_rng(55)
    books = synth_model(rng, D, M, K, scale=0.1)
    n = 70_000 + 5
    codes = rng.integers(0, K, size=(n, M), dtype=np.uint8)
    codes[17_500:17_520] = codes[3]; codes[35_001] = codes[3]; codes[69_999] = codes[3]    # exact ties on both sides of every boundary
    nq = 4096 + 160
    q = (rng.normal(size=(nq, D)) * 0.1).astype(np.float32)
    q[4100] = q[7]
    idx = amd.OpqIndex(np.zeros((1, D), np.float32), books)
    idx.add_codes(codes)
    idx.set_param("scan_variant", 3)
    qd = torch.from_numpy(q).cuda()
    res = {}
    try:
        for tail in (0, -1, 2):
            amd.set_tuning("scan_tail_splits", tail)
            d, i = idx.search(qd, k, rotate=False)
            res[tail] = (d.cpu().numpy(), i.cpu().numpy())
    finally:
        amd.set_tuning("scan_tail_splits", 0)
    for tail in (0, 2):
        assert np.array_equal(res[tail][1], res[-1][1]) and np.array_equal(bits(res[tail][0]), bits(res[-1][0])), tail
    sel = np.r_[0:8, 4090:4110, nq - 8:nq]
    od, oi = orc.adc_search(q[sel], books, codes, k)
    assert np.array_equal(res[0][1][sel], oi) and np.array_equal(bits(res[0][0][sel]), bits(od))
    # the host-pointer entry on the same batch: pageable arrays (pipelined pieces), page-locked result arrays through the copy engines
    # (opq_host_zero_copy 0) and written by the kernels themselves in one launch chain (1, the default) -- the first region's groups
    # store their lists straight into the caller's memory, the merge fills in the tail's
    outs = {"pageable": (np.zeros((nq, k), np.float32), np.zeros((nq, k), np.int64)),
            "page-locked": (amd.pinned_empty((nq, k), np.float32), amd.pinned_empty((nq, k), np.int64))}
    qp = amd.pinned_empty((nq, D), np.float32); qp[:] = q
    try:
        for zc in (0, 1):
            amd.set_tuning("opq_host_zero_copy", zc)
            for name, out in outs.items():
                for qa in (q, qp):
                    out[0][:] = 0; out[1][:] = -7
                    idx.search(qa, k, rotate=False, out=out)
                    assert np.array_equal(out[1], res[-1][1]) and np.array_equal(bits(out[0]), bits(res[-1][0])), (zc, name)
    finally:
        amd.set_tuning("opq_host_zero_copy", 1)


@pytest.mark.parametrize("M,step", [(8, 16), (4, 32), (12, 8), (5, 8), (1, 32), (15, 4)])
def test_search_any_m_through_padded_rows(amd, orc, M, step):
    """Round 5: an index with M < 16 is searched by the M = 16 kernels over rows padded to 16 code bytes with zeros and all-zero tables
    behind the model's own (M = 8 at C2's shape: 1.0 -> 3.1 M queries/s).  Same lists as the oracle for every M below 16 -- 12, 5, 1 and
    15 had no scan kernel at all before --, for one query and for batches, appended rows included; M = 4 / 8 also against their old
    row-per-lane kernels (scan_pad_m 0)."""
    import torch
    D, K = M * step, 256
    rng = np.random.default_rng(M * 31 + step)
    books = (rng.normal(size=(M, K, step)) * 0.1).astype(np.float32)
    n = 70_000 + 7
    codes = rng.integers(0, K, size=(n, M), dtype=np.uint8)
    codes[100] = codes[50]; codes[40_000] = codes[50]; codes[n - 1] = codes[50]
    idx = amd.OpqIndex(np.zeros((1, D), np.float32), books)
    idx.add_codes(codes[:30_000])
    try:
        for nq, k in ((1, 10), (3, 100), (9, 128), (300, 10), (4200, 20)):
            q = (rng.normal(size=(nq, D)) * 0.1).astype(np.float32)
            q[0] = books[np.arange(M), codes[50]].reshape(-1)            # the tied rows' own reconstruction: distance 0 three times
            if nq == 9:
                idx.add_codes(codes[30_000:])                            # the padded / rotated copies are extended, not rebuilt
            rows = codes[:idx.ntotal]
            sel = np.unique(np.r_[0:min(nq, 4), rng.integers(0, nq, size=min(nq, 12))])
            od, oi = orc.adc_search(q[sel], books, rows, k)
            # M = 8 / 4: the native packed scan (round 6, default), the padded rows (round 5) and the old row-per-lane kernels
            for pad, packed in (((1, 1), (1, 0), (0, 0)) if M in (4, 8) else ((1, 1),)):
                amd.set_tuning("scan_pad_m", pad); amd.set_tuning("scan_packed_m", packed)
                for dev in (False, True):
                    d, i = idx.search(torch.from_numpy(q).cuda() if dev else q, k, rotate=False)
                    if dev:
                        d, i = d.cpu().numpy(), i.cpu().numpy()
                    assert np.array_equal(i[sel], oi) and np.array_equal(bits(d[sel]), bits(od)), (M, nq, k, pad, packed, dev)
    finally:
        amd.set_tuning("scan_pad_m", 1); amd.set_tuning("scan_packed_m", 1)
    idx.close()


@pytest.mark.parametrize("M,step", [(8, 16), (4, 8), (8, 4)])
def test_search_packed_rows_m8_m4(amd, orc, M, step):
    """Round 6: adc_scan16p -- an M = 8 / M = 4 index scanned as it lies in memory, 16 / M rows per 16-byte load, table
    copies per lane group, no padded rows.  Against the oracle: ragged row counts (last group / last load partly filled), rows appended at
    odd counts (the packed rotation is extended from the middle of a group), forced row splits (shared thresholds, ties across the
    boundaries), an id base, all-identical rows (every distance ties), descending distances (every row beats the threshold: the
    full-buffer retry path), fewer rows than k; and the same lists from the padded-row form (opq/src/IVFOPQ.h:24-29: any M <= 16)."""
    import torch
    D, K = M * step, 256
    rng = np.random.default_rng(M * 131 + step)
    books = (rng.normal(size=(M, K, step)) * 0.1).astype(np.float32)
    idx = amd.OpqIndex(np.zeros((1, D), np.float32), books)
    try:
        # growing index: every size is searched right after an append at an odd row count
        total = 0
        allc = np.zeros((0, M), np.uint8)
        for add in (3, 60, 2048 + 1, 40_000 + 3, 150_000 + 5):
            c = rng.integers(0, K, size=(add, M), dtype=np.uint8)
            if total:
                c[0] = allc[total // 2]; c[add - 1] = allc[total // 2]      # exact duplicates of an older row: ties
            idx.add_codes(c); allc = np.concatenate([allc, c]); total += add
            for nq, k in ((8, 100), (37, 10)) if total < 100_000 else ((8, 100), (37, 10), (2500, 100)):
                q = (rng.normal(size=(nq, D)) * 0.1).astype(np.float32)
                q[0] = books[np.arange(M), allc[total // 2]].reshape(-1)
                sel = np.unique(np.r_[0:min(nq, 4), rng.integers(0, nq, size=min(nq, 12))])
                od, oi = orc.adc_search(q[sel], books, allc, k)
                d, i = idx.search(torch.from_numpy(q).cuda(), k, rotate=False)
                d, i = d.cpu().numpy(), i.cpu().numpy()
                assert np.array_equal(i[sel], oi) and np.array_equal(bits(d[sel]), bits(od)), (M, total, nq, k)
        # forced row splits + an id base; the padded form must agree bit for bit
        q = (rng.normal(size=(64, D)) * 0.1).astype(np.float32)
        q[1] = books[np.arange(M), allc[7]].reshape(-1)
        od, oi = orc.adc_search(q, books, allc, 100)
        idx.set_id_base(1 << 34)
        for splits in (1, 2, 3, 8):
            idx.set_param("splits", splits)
            for packed in (1, 0):
                amd.set_tuning("scan_packed_m", packed)
                d, i = idx.search(torch.from_numpy(q).cuda(), 100, rotate=False)
                assert np.array_equal(i.cpu().numpy(), oi + (1 << 34)) and np.array_equal(bits(d.cpu().numpy()), bits(od)), (M, splits, packed)
        amd.set_tuning("scan_packed_m", 1)
        idx.set_param("splits", 0); idx.set_id_base(0)
        # all rows identical; descending distances
        idx.reset()
        idx.add_codes(np.tile(allc[:1], (5001, 1)))
        d, i = idx.search(q[:9], 100, rotate=False)
        assert np.array_equal(i, np.tile(np.arange(100), (9, 1)))
        idx.reset()
        lut = orc.lut(q[0], np.zeros(D, np.float32), books)
        ranks = np.argsort(-lut[0], kind="stable")
        n = 6001
        desc = np.zeros((n, M), dtype=np.uint8)
        desc[:, 0] = ranks[(np.arange(n) * 256 // n)]
        idx.add_codes(desc)
        od, oi = orc.adc_search(q[:9], books, desc, 100)
        for splits in (1, 2):
            idx.set_param("splits", splits)
            d, i = idx.search(q[:9], 100, rotate=False)
            assert np.array_equal(i, oi) and np.array_equal(bits(d), bits(od)), (M, "descending", splits)
    finally:
        amd.set_tuning("scan_packed_m", 1)
    idx.close()


def test_search_edge_cases(amd, orc):
    D, M, K = 128, 16, 256
    rng = np.random.default_rng(2)
    books = synth_model(rng, D, M, K, scale=0.1)
    idx = amd.OpqIndex(np.zeros((1, D), np.float32), books)
    q = (rng.normal(size=(3, D)) * 0.1).astype(np.float32)
    # empty index: every slot padded with (+inf, -1)
    d, i = idx.search(q, 10, rotate=False)
    assert np.all(np.isinf(d)) and np.all(i == -1)
    # fewer rows than k
    codes = rng.integers(0, K, size=(7, M), dtype=np.uint8)
    idx.add_codes(codes)
    d, i = idx.search(q, 10, rotate=False)
    od, oi = orc.adc_search(q, books, codes, 10)
    assert np.array_equal(i, oi) and np.array_equal(bits(d), bits(od))
    assert np.all(i[:, 7:] == -1)
    # all rows identical (worst case for the selection: every distance ties) and an id base
    idx.reset()
    same = np.tile(codes[:1], (5000, 1))
    idx.add_codes(same)
    idx.set_id_base(1 << 33)
    for variant in (3, 6):
        idx.set_param("scan_variant", variant)
        d, i = idx.search(q, 100, rotate=False)
        assert np.array_equal(i, np.tile((1 << 33) + np.arange(100), (3, 1))), variant
    idx.set_param("scan_variant", 3)
    # descending distances: every new row beats the threshold (stresses buffer overflow + retry path)
    idx.reset(); idx.set_id_base(0)
    order_books = books.copy()
    lut = orc.lut(q[0], np.zeros(D, np.float32), order_books)
    ranks = np.argsort(-lut[0], kind="stable")          # code values by descending LUT[0] entry
    n = 6000
    desc = np.zeros((n, M), dtype=np.uint8)
    desc[:, 0] = ranks[(np.arange(n) * 256 // n)]
    idx.add_codes(desc)
    od, oi = orc.adc_search(q, order_books, desc, 100)
    for variant in (6, 5, 4, 3, 1, 2, 0):
        for splits in (1, 2):
            idx.set_param("scan_variant", variant); idx.set_param("splits", splits)
            d, i = idx.search(q, 100, rotate=False)
            assert np.array_equal(i, oi) and np.array_equal(bits(d), bits(od)), (variant, splits)
    # error path: k out of range is refused, not truncated
    with pytest.raises(amd.CvtmiError):
        idx.search(q, 2049, rotate=False)
    with pytest.raises(amd.CvtmiError):
        idx.search(q, 0, rotate=False)


def test_two_region_scan_plan(amd, orc):
    """Tail groups split finer than the leading ones (kernels.h ScanPlan::groups_a / splits_b): same results."""
    D, M, K = 128, 16, 256
    rng = np.random.default_rng(21)
    books = synth_model(rng, D, M, K, scale=0.1)
    codes = rng.integers(0, K, size=(9000, M), dtype=np.uint8)
    q = (rng.normal(size=(77, D)) * 0.1).astype(np.float32)
    od, oi = orc.adc_search(q, books, codes, 100)
    idx = amd.OpqIndex(np.zeros((1, D), np.float32), books)
    idx.add_codes(codes)
    for variant in (5, 4, 3):
        for splits, ga, sb in ((1, 3, 2), (1, 9, 3), (2, 1, 5), (8, 4, 16), (1, 10, 4), (1, 3, 8)):
            idx.set_param("scan_variant", variant); idx.set_param("splits", splits)
            idx.set_param("groups_a", ga); idx.set_param("splits_b", sb)
            d, i = idx.search(q, 100, rotate=False)
            assert np.array_equal(i, oi) and np.array_equal(bits(d), bits(od)), (variant, splits, ga, sb)
    idx.set_param("groups_a", 0); idx.set_param("splits_b", 0); idx.set_param("splits", 0)
    # rows rotated in registers instead of streamed from the pre-rotated copy; appended rows extend the copy
    for pre in (0, 1):
        idx.set_param("prerotate", pre)
        for variant in (5, 4, 3):
            idx.set_param("scan_variant", variant)
            d, i = idx.search(q, 100, rotate=False)
            assert np.array_equal(i, oi) and np.array_equal(bits(d), bits(od)), (pre, variant)
    more = rng.integers(0, K, size=(777, M), dtype=np.uint8)
    idx.add_codes(more)
    od2, oi2 = orc.adc_search(q, books, np.concatenate([codes, more]), 100)
    d, i = idx.search(q, 100, rotate=False)
    assert np.array_equal(i, oi2) and np.array_equal(bits(d), bits(od2))
    idx.reset(); idx.add_codes(more)
    od3, oi3 = orc.adc_search(q, books, more, 100)
    d, i = idx.search(q, 100, rotate=False)
    assert np.array_equal(i, oi3) and np.array_equal(bits(d), bits(od3))


def test_search_device_pointers_and_rotation(amd, orc):
    """_dev entry points on torch tensors, rotation by a dense orthonormal R through the MFMA GEMM."""
    import torch
    from cvt_amd import synth
    D, M, K = 128, 16, 256
    R = synth.random_rotation(D, seed=7)
    x = synth.sift_like(6000, D, device="cuda")
    qs = synth.sift_like(33, D, seed=0xBEEF, device="cuda")
    xr_ref = orc.rotate_fma(R, x.cpu().numpy())
    books = synth_model(np.random.default_rng(4), D, M, K, scale=0.1)
    books = np.ascontiguousarray(np.stack([xr_ref[np.random.default_rng(m).integers(0, 6000, K), m * 8:(m + 1) * 8] for m in range(M)]))
    idx = amd.OpqIndex(np.zeros((1, D), np.float32), books, R=R)
    xr = idx.rotate(x)
    assert np.array_equal(bits(xr.cpu().numpy()), bits(xr_ref)), "MFMA rotation != k-ordered fmaf chain"
    lists, codes = idx.encode(xr)
    ol, oc = orc.pq_encode(xr_ref, np.zeros((1, D), np.float32), books)
    assert np.array_equal(codes.cpu().numpy(), oc)
    idx.add_codes(codes)
    d, i = idx.search(qs, 100, rotate=True)
    torch.cuda.synchronize()
    od, oi = orc.adc_search(orc.rotate_fma(R, qs.cpu().numpy()), books, oc, 100)
    assert np.array_equal(i.cpu().numpy(), oi)
    assert np.array_equal(bits(d.cpu().numpy()), bits(od))


def test_full_size_properties(amd, orc):
    """BASELINE config sizes (SIFT-1M, M=16, top-100) through size-independent properties:
    split-invariance, merge-of-shards == whole, sortedness, and an oracle check on a query sample."""
    import torch
    from cvt_amd import synth
    D, M, K, n, nq, k = 128, 16, 256, 1_000_000, 64, 100
    R = synth.random_rotation(D)
    x = synth.sift_like(n, D, device="cuda")
    q = synth.sift_like(nq, D, seed=0xBEEF, device="cuda")
    idx0 = amd.OpqIndex(np.zeros((1, D), np.float32), np.zeros((M, K, D // M), np.float32), R=R)
    books = synth.train_books(idx0.rotate(x[:50000]), M, K, iters=3)
    idx = amd.OpqIndex(np.zeros((1, D), np.float32), books, R=R)
    _, codes = idx.encode(idx.rotate(x))
    idx.add_codes(codes)
    d, i = idx.search(q, k)
    dn, inn = d.cpu().numpy(), i.cpu().numpy()
    assert np.all(np.diff(dn, axis=1) >= 0)                                   # ascending
    tie = np.diff(dn, axis=1) == 0
    assert np.all(np.diff(inn, axis=1)[tie] > 0)                              # ties in id order
    for variant, qt, splits in ((0, 1, 8), (0, 2, 16), (0, 4, 1), (1, 4, 24), (2, 4, 1), (2, 4, 16), (1, 4, 1), (3, 0, 1), (3, 0, 8), (4, 0, 1), (4, 0, 2), (5, 0, 1), (5, 0, 2), (5, 0, 8), (6, 0, 0), (6, 0, 1), (6, 0, 2), (6, 0, 8)):
        idx.set_param("scan_variant", variant); idx.set_param("qtile", qt); idx.set_param("splits", splits)
        d2, i2 = idx.search(q, k)
        assert torch.equal(i2, i) and torch.equal(d2.view(torch.int32), d.view(torch.int32)), (variant, qt, splits)
    idx.set_param("scan_variant", 1); idx.set_param("qtile", 0); idx.set_param("splits", 0)
    # two row shards searched separately and merged == the whole index
    half = n // 2
    parts_d, parts_i = [], []
    for s, (a, b) in enumerate(((0, half), (half, n))):
        sh = amd.OpqIndex(np.zeros((1, D), np.float32), books, R=R)
        sh.add_codes(codes[a:b].contiguous()); sh.set_id_base(a)
        dd, ii = sh.search(q, k)
        parts_d.append(dd); parts_i.append(ii)
    md, mi = amd.topk_merge(torch.stack(parts_d, 1).contiguous(), torch.stack(parts_i, 1).contiguous(), k)
    assert torch.equal(mi, i) and torch.equal(md.view(torch.int32), d.view(torch.int32))
    # oracle on a sample of queries at full N
    qr = idx.rotate(q[:4]).cpu().numpy()
    od, oi = orc.adc_search(qr, books, codes.cpu().numpy(), k)
    assert np.array_equal(inn[:4], oi) and np.array_equal(bits(dn[:4]), bits(od))


def test_scan_row_offsets_beyond_2pow28(amd, orc):
    """300 M code rows on one GPU (4.8 GB + the rotated copy): rows planted past 2^28 and at the very end with the
    query's best code per sub-quantiser must come back first, in id order (32-bit byte offsets inside a split,
    64-bit everywhere else)."""
    import torch
    D, M, K = 128, 16, 256
    n = 300_000_000
    free_b, _ = torch.cuda.mem_get_info()
    if free_b < n * M * 2.5:
        pytest.skip("not enough free HBM")
    rng = np.random.default_rng(5)
    books = synth_model(rng, D, M, K, scale=0.1)
    q = (rng.normal(size=(9, D)) * 0.1).astype(np.float32)
    idx = amd.OpqIndex(np.zeros((1, D), np.float32), books)
    idx.reserve(n)
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    planted = [123, (1 << 28) + 5, 290_000_007, n - 1]
    lut0 = orc.lut(q[0], np.zeros(D, np.float32), books)            # [M][K]
    best = lut0.argmin(axis=1).astype(np.uint8)
    done = 0
    while done < n:
        m = min(1 << 25, n - done)
        chunk = torch.randint(0, 256, (m, M), generator=g, device="cuda", dtype=torch.uint8)
        for p in planted:
            if done <= p < done + m:
                chunk[p - done] = torch.from_numpy(best).cuda()
        idx.add_codes(chunk)
        done += m
    d, i = idx.search(q, 10, rotate=False)
    assert list(i[0, :4]) == planted, i[0]
    dmin = np.float32(0)
    for m_ in range(M):
        dmin = np.float32(dmin + lut0[m_, best[m_]])
    assert np.all(bits(d[0, :4]) == bits(np.array([dmin], np.float32))[0])
    assert np.all(d[:, 1:] >= d[:, :-1]) and i.min() >= 0 and i.max() < n


def test_scan_lazy_selection_and_shared_thresholds(amd, orc):
    """adc_scan16q between checkpoints: selection on the integer lower bounds (exact sums once, at the end) and row splits
    that publish their thresholds to each other.  Same answers with either switch on or off, on the cases that stress them:
    a crowded band around the k-th row (masses of equal / near-equal rows: the query must fall back to exact keys),
    tables with non-finite entries (K < 256: the +inf padding a stray code reaches; lazy must not start), large common
    offsets (bias >> range: the slack grows with bias / scale) and many row splits."""
    D, M, K = 128, 16, 256
    rng = np.random.default_rng(77)
    books = synth_model(rng, D, M, K, scale=0.1)
    n = 60_000
    codes = rng.integers(0, K, size=(n, M), dtype=np.uint8)
    near = codes[123].copy()
    crowd = np.tile(near, (4000, 1))                      # 4000 rows that differ from one another in one byte only
    crowd[:, 5] = rng.integers(0, K, size=4000)
    crowd[:700] = near                                    # ... 700 of them exact duplicates
    codes[20_000:24_000] = crowd
    q = (rng.normal(size=(19, D)) * 0.1).astype(np.float32)
    # query 0 sits on the crowd's codewords: the k-th best is deep inside the duplicates
    q[0] = np.concatenate([books[m, near[m]] for m in range(M)])
    shifted = books + 3.0                                 # every table entry ~ 8 * 9 = 72 with a tiny spread: bias / scale is large
    for bk, tag in ((books, "plain"), (shifted.astype(np.float32), "shifted")):
        idx = amd.OpqIndex(np.zeros((1, D), np.float32), bk)
        idx.add_codes(codes)
        for k in (1, 100, 128):
            od, oi = orc.adc_search(q, bk, codes, k)
            for variant in (3, 4, 5, 6):
                for lazy, share, splits in ((1, 1, 0), (1, 1, 1), (1, 1, 3), (1, 0, 3), (0, 1, 3), (0, 0, 1), (1, 1, 16)):
                    idx.set_param("scan_variant", variant); idx.set_param("scan_lazy", lazy); idx.set_param("scan_share", share)
                    idx.set_param("splits", splits)
                    d, i = idx.search(q, k, rotate=False)
                    assert np.array_equal(i, oi), (tag, k, variant, lazy, share, splits)
                    assert np.array_equal(bits(d), bits(od)), (tag, k, variant, lazy, share, splits)
        idx.close()
    # non-finite tables: K = 200 (codes >= 200 read the +inf padding) and a NaN / inf query
    K2 = 200
    books2 = synth_model(rng, D, M, K2, scale=0.1)
    codes2 = rng.integers(0, K2, size=(30_000, M), dtype=np.uint8)
    codes2[777, 3] = 250                                   # stray code past K: its distance is +inf, it must rank last
    q2 = (rng.normal(size=(9, D)) * 0.1).astype(np.float32)
    idx = amd.OpqIndex(np.zeros((1, D), np.float32), books2)
    idx.add_codes(codes2)
    od, oi = orc.adc_search(q2, books2, codes2, 100)
    assert np.all(np.isfinite(od)) and not np.any(oi == 777)
    for variant in (3, 6):
        for splits in (0, 1, 4):
            idx.set_param("scan_variant", variant); idx.set_param("splits", splits)
            d, i = idx.search(q2, 100, rotate=False)
            assert np.array_equal(i, oi) and np.array_equal(bits(d), bits(od)), (variant, splits)
    # a NaN and an inf query beside ordinary ones (their sums bound nothing: the spill area overflows, exact reductions take over)
    q3 = q2.copy(); q3[2, 5] = np.nan; q3[6, 77] = np.inf
    od, oi = orc.adc_search(q3, books2, codes2, 100)
    for variant in (3, 6):
        idx.set_param("scan_variant", variant); idx.set_param("splits", 0)
        d, i = idx.search(q3, 100, rotate=False)
        ok = np.array([f not in (2, 6) for f in range(len(q3))])   # the finite queries of the batch must not notice
        assert np.array_equal(i[ok], oi[ok]) and np.array_equal(bits(d)[ok], bits(od)[ok]), variant
    idx.close()


def test_rotate_encode_one_call(amd, orc):
    """cvtmi_opq_rotate_encode = rotate then encode, chunked through the handle's scratch (several chunks, a ragged tail,
    permutation and dense rotation, coarse lists, host and device entry points)."""
    import torch
    from cvt_amd import synth
    D, M, K = 128, 16, 256
    rng = np.random.default_rng(321)
    books = synth_model(rng, D, M, K)
    for coarseK, rot in ((1, "R"), (7, "perm"), (1, "perm")):   # (1, perm): the encode kernel gathers through the permutation itself
        coarse = np.zeros((1, D), np.float32) if coarseK == 1 else rng.normal(size=(coarseK, D)).astype(np.float32)
        kw = {"R": synth.random_rotation(D, seed=5)} if rot == "R" else {"perm": synth.random_permutation(D, seed=5)}
        idx = amd.OpqIndex(coarse, books, **kw)
        n = 131072 * 2 + 777
        x = torch.from_numpy(rng.normal(size=(n, D)).astype(np.float32)).cuda()
        x[17, 5] = float("nan"); x[18] = float("inf"); x[19] = 0.0; x[20] = x[21]      # hard rows: same answer on either path
        l0, c0 = idx.encode(idx.rotate(x))
        l1, c1 = idx.rotate_encode(x)
        assert torch.equal(c0, c1) and torch.equal(l0, l1)
        l2, c2 = idx.rotate_encode(x[:5000].cpu().numpy())
        assert np.array_equal(c2, c0[:5000].cpu().numpy()) and np.array_equal(l2, l0[:5000].cpu().numpy())
        if rot == "perm" and coarseK == 1:   # and the checker's encode of the permuted rows (rows without non-finite values)
            xp = x[100:400].cpu().numpy()[:, kw["perm"]]
            _, oc = orc.pq_encode(np.ascontiguousarray(xp), coarse, books)
            assert np.array_equal(oc, c1[100:400].cpu().numpy())
        idx.close()


@pytest.mark.gpu
def test_scan_seed_thresholds(amd, orc):
    """scan16q_seed: the first thresholds come from a histogram of a split's first 2048 rows.  Seed regions made of duplicates of
    the best row, of the worst rows only (the true neighbours all come later), of rows far apart (one per bin), k = 1 / 100 / 128,
    every split count, seed on == seed off == the checker."""
    D, M, K = 128, 16, 256
    rng = np.random.default_rng(77)
    books = synth_model(rng, D, M, K, scale=0.1)
    n = 40_000
    base = rng.integers(0, K, size=(n, M), dtype=np.uint8)
    q = (rng.normal(size=(17, D)) * 0.1).astype(np.float32)
    lut0 = orc.lut(q[0], np.zeros(D, np.float32), books)                 # [M][K] table of query 0
    best = np.argmin(lut0, axis=1).astype(np.uint8)                       # its nearest code word per sub-quantiser
    worst = np.argmax(lut0, axis=1).astype(np.uint8)
    cases = {}
    c = base.copy(); c[:2048] = best; cases["seed region = 2048 copies of query 0's best row"] = c
    c = base.copy(); c[:2048] = worst; c[30_000:30_300] = best; cases["seed region = the worst row, neighbours much later"] = c
    c = base.copy()
    order = np.argsort(lut0, axis=1)                                      # rows walking from best to worst: sums spread over every bin
    for r in range(2048):
        c[r] = order[np.arange(M), (r * K // 2048)]
    cases["seed region spread over the whole range"] = c
    try:
        for tag, codes in cases.items():
            idx = amd.OpqIndex(np.zeros((1, D), np.float32), books)
            idx.add_codes(codes)
            for k in (1, 100, 128):
                od, oi = orc.adc_search(q, books, codes, k)
                for variant in (3, 4, 5):
                    for splits in (0, 1, 2, 4):
                        for seed in (1, 0):
                            amd.set_tuning("scan_seed", seed)
                            idx.set_param("scan_variant", variant); idx.set_param("splits", splits)
                            d, i = idx.search(q, k, rotate=False)
                            assert np.array_equal(i, oi), (tag, k, variant, splits, seed)
                            assert np.array_equal(bits(d), bits(od)), (tag, k, variant, splits, seed)
            idx.close()
    finally:
        amd.set_tuning("scan_seed", 1)


@pytest.mark.gpu
def test_scan_h_item_tables(amd, orc):
    """adc_scan16h (scan_variant 6): every shape of its item table gives the oracle's answer -- equal shares of the flat
    (query group x row) space (groups cut at arbitrary tiles, several items per workgroup, shares smaller than a group and
    larger than several), (group, split) blocks taken round-robin, one query, a ragged last group, k larger than a segment,
    and an index whose candidates overflow the spill areas (more rows below the bound than 4096: the mid-scan reduction)."""
    D, M, K = 128, 16, 256
    rng = np.random.default_rng(606)
    books = synth_model(rng, D, M, K, scale=0.1)
    n = 70_000 + 5
    codes = rng.integers(0, K, size=(n, M), dtype=np.uint8)
    codes[100] = codes[50]; codes[40_000] = codes[50]; codes[69_999] = codes[50]
    idx = amd.OpqIndex(np.zeros((1, D), np.float32), books)
    idx.add_codes(codes)
    idx.set_param("scan_variant", 6)
    try:
        for nq in (1, 3, 8, 9, 77, 530):
            q = (rng.normal(size=(nq, D)) * 0.1).astype(np.float32)
            for k in (1, 100, 128):
                od, oi = orc.adc_search(q, books, codes, k)
                for balance, min_rows, splits in ((1, 2048, 0), (1, 8192, 0), (1, 16384, 0), (1, 1 << 20, 0), (2, 0, 0), (2, 0, 5), (2, 0, 16), (0, 0, 0)):
                    amd.set_tuning("scanh_balance", balance)
                    if min_rows: amd.set_tuning("scanh_min_rows", min_rows)
                    idx.set_param("splits", splits)
                    d, i = idx.search(q, k, rotate=False)
                    assert np.array_equal(i, oi), (nq, k, balance, min_rows, splits)
                    assert np.array_equal(bits(d), bits(od)), (nq, k, balance, min_rows, splits)
        # 9000 rows of which 6000 are one and the same vector, the query on it: every one of them is a candidate
        same = codes[:9000].copy(); same[1000:7000] = codes[7]
        idx.reset(); idx.add_codes(same)
        q = (rng.normal(size=(5, D)) * 0.1).astype(np.float32)
        q[1] = np.concatenate([books[m, codes[7][m]] for m in range(M)])
        for k in (10, 128):
            od, oi = orc.adc_search(q, books, same, k)
            for balance, splits in ((1, 0), (2, 1), (2, 2)):
                amd.set_tuning("scanh_balance", balance); idx.set_param("splits", splits)
                d, i = idx.search(q, k, rotate=False)
                assert np.array_equal(i, oi) and np.array_equal(bits(d), bits(od)), (k, balance, splits)
    finally:
        amd.set_tuning("scanh_balance", 0); amd.set_tuning("scanh_min_rows", 16384)
        idx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("M", [16, 8])
def test_search_any_k(amd, orc, M):
    """k beyond 128 (get_sort_results(score, num_show) takes any, opq/src/common.h:25-37): 129 ... 2048 through the exact kernel with
    the large selection buffer -- row splits, ties straddling them, k larger than a split and larger than the index"""
    D, K = 128, 256
    rng = np.random.default_rng(900 + M)
    books = synth_model(rng, D, M, K, scale=0.1)
    n = 30000 + 7
    codes = rng.integers(0, K, size=(n, M), dtype=np.uint8)
    codes[100] = codes[50]; codes[29000] = codes[50]; codes[12000:12300] = codes[77]
    q = (rng.normal(size=(6, D)) * 0.1).astype(np.float32)
    q[1] = np.concatenate([books[m, codes[77][m]] for m in range(M)])   # 300 exact ties at the top
    idx = amd.OpqIndex(np.zeros((1, D), np.float32), books)
    idx.add_codes(codes)
    for k in (129, 500, 1000, 2048):
        od, oi = orc.adc_search(q, books, codes, k)
        for variant, splits in ((7, 0), (3, 1), (6, 3), (0, 7), (0, 64)):
            idx.set_param("scan_variant", variant); idx.set_param("splits", splits)
            d, i = idx.search(q, k, rotate=False)
            assert np.array_equal(i, oi), (k, variant, splits)
            assert np.array_equal(bits(d), bits(od)), (k, variant, splits)
    idx.reset(); idx.add_codes(codes[:300])                                   # k > rows: padded with (+inf, -1)
    idx.set_param("scan_variant", 7); idx.set_param("splits", 0)
    d, i = idx.search(q, 1000, rotate=False)
    od, oi = orc.adc_search(q, books, codes[:300], 300)
    assert np.array_equal(i[:, :300], oi) and np.all(i[:, 300:] == -1) and np.all(np.isinf(d[:, 300:]))
    idx.close()


@pytest.mark.gpu
def test_search_big_k_through_the_filter_scan(amd, orc):
    """Round 6: k = 129 .. 2048 at M = 16 on >= 65 536 rows through the bound-first filter pipeline (adc_scan_h.hip:
    sampled histogram bound -> candidate lists -> one selection workgroup per query) instead of the exact kernel with one query per
    workgroup.  Same lists as the oracle and as the exact kernel ("scan_bigk" 0), bit for bit: exact ties at the top and across the k-th
    place, 5000 copies of one query's nearest row (its band holds more than the selection sorts: that query is flagged and answered by
    the exact kernel behind the pipeline), a NaN query (tables that bound nothing: flagged), appended rows, an id base, rotation on,
    host and device pointers; get_sort_results takes any num_show (opq/src/common.h:25-37)."""
    import torch
    from cvt_amd import synth
    D, M, K = 128, 16, 256
    rng = np.random.default_rng(77)
    books = synth_model(rng, D, M, K, scale=0.1)
    R = synth.random_rotation(D, seed=3)
    n = 150_000 + 11
    codes = rng.integers(0, K, size=(n, M), dtype=np.uint8)
    codes[100] = codes[50]; codes[140_000] = codes[50]; codes[12000:12300] = codes[77]
    codes[60_000:65_000] = codes[99]                                     # 5000 equal rows: a crowded band for the query that hits them
    idx = amd.OpqIndex(np.zeros((1, D), np.float32), books, R=R)
    idx.add_codes(codes[:100_000])
    idx.add_codes(codes[100_000:])
    idx.set_id_base(1 << 35)
    nq = 41
    qr = (rng.normal(size=(nq, D)) * 0.1).astype(np.float32)            # queries in the ROTATED space ...
    qr[1] = np.concatenate([books[m, codes[77][m]] for m in range(M)])  # 300 exact ties at distance 0
    qr[2] = np.concatenate([books[m, codes[99][m]] for m in range(M)])  # 5000 exact ties at distance 0
    qr[3] = np.concatenate([books[m, codes[50][m]] for m in range(M)])
    q = (qr.astype(np.float64) @ R.astype(np.float64)).astype(np.float32)   # ... brought back: the library rotates them (y = R x)
    q[5, 7] = np.nan
    q_rot = orc.rotate_fma(R, q)
    try:
        for k in (129, 300, 1000, 2048):
            od, oi = orc.adc_search(q_rot, books, codes, k)
            ok = ~np.isnan(q_rot).any(axis=1)                            # (a NaN query: every distance is NaN; the order is the kernels' own)
            ref = None
            for bigk in (1, 0):
                amd.set_tuning("scan_bigk", bigk)
                for dev in (True, False):
                    d, i = idx.search(torch.from_numpy(q).cuda() if dev else q, k, rotate=True)
                    if dev:
                        d, i = d.cpu().numpy(), i.cpu().numpy()
                    assert np.array_equal(i[ok], oi[ok] + (1 << 35)), (k, bigk, dev)
                    assert np.array_equal(bits(d[ok]), bits(od[ok])), (k, bigk, dev)
                    if ref is None:
                        ref = (d, i)
                    assert np.array_equal(i, ref[1]) and np.array_equal(bits(d), bits(ref[0])), (k, bigk, dev, "NaN query included")
    finally:
        amd.set_tuning("scan_bigk", 1)
    idx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("rotation", ["dense", "perm", "none"])
def test_small_batch_path(amd, orc, rotation):
    """1 .. 128 queries (the reference's own call pattern: 1-9 query frames per Query) through the small-batch path -- rotation folded
    into the table kernel, a global bound from a histogram pass, candidate lists, selection by the last workgroup -- against the oracle
    and against the ordinary path: k = 1 .. 128, exact ties, a list that overflows (7000 copies of the query's nearest row: the exact
    fall-back inside the kernel), a NaN query beside ordinary ones, appended rows, an id base, device and host pointers."""
    import torch
    from cvt_amd import synth
    D, M, K = 128, 16, 256
    rng = np.random.default_rng({"dense": 1, "perm": 2, "none": 3}[rotation])
    books = synth_model(rng, D, M, K, scale=0.1)
    kw = {}
    if rotation == "dense":
        kw["R"] = synth.random_rotation(D, seed=5)
    elif rotation == "perm":
        kw["perm"] = rng.permutation(D).astype(np.int32)
    idx = amd.OpqIndex(np.zeros((1, D), np.float32), books, **kw)
    n = 150_000 + 11
    codes = rng.integers(0, K, size=(n, M), dtype=np.uint8)
    codes[100] = codes[50]; codes[140_000] = codes[50]
    idx.add_codes(codes[:90_000]); idx.add_codes(codes[90_000:])
    idx.set_id_base(1 << 34)
    q = (rng.normal(size=(128, D)) * 0.1).astype(np.float32)

    def rot(x):
        if rotation == "dense":
            return orc.rotate_fma(kw["R"], x)
        if rotation == "perm":
            return x[:, kw["perm"]]
        return x

    for nq in (1, 2, 5, 8, 9, 17, 32, 100, 128):   # up to sixteen query groups; 9 and 17: a last group with one real query and seven copies
        for k in ((1, 10, 100, 128) if nq <= 32 else (100,)):
            od, oi = orc.adc_search(rot(q[:nq]), books, codes, k)
            for small in (1, 0):
                idx.set_param("scan_small", small)
                d, i = idx.search(q[:nq], k, rotate=True)
                assert np.array_equal(i, oi + (1 << 34)), (nq, k, small)
                assert np.array_equal(bits(d), bits(od)), (nq, k, small)
            idx.set_param("scan_small", 1)
            dd, ii = idx.search(torch.from_numpy(q[:nq]).cuda(), k, rotate=True)
            assert np.array_equal(ii.cpu().numpy(), oi + (1 << 34)) and np.array_equal(bits(dd.cpu().numpy()), bits(od))
    # a candidate list that overflows, and sums that bound nothing
    idx.set_id_base(0)
    crowd = codes.copy(); crowd[20_000:27_000] = codes[7]
    idx.reset(); idx.add_codes(crowd)
    q2 = q[:8].copy()
    raw7 = np.concatenate([books[m, codes[7][m]] for m in range(M)])            # the rotated point that sits on row 7's codewords
    if rotation == "dense":
        q2[3] = (kw["R"].T.astype(np.float64) @ raw7.astype(np.float64)).astype(np.float32)
    elif rotation == "perm":
        q2[3][kw["perm"]] = raw7
    else:
        q2[3] = raw7
    q2[5, 17] = np.nan
    for k in (10, 128):
        od, oi = orc.adc_search(rot(q2), books, crowd, k)
        d, i = idx.search(q2, k, rotate=True)
        ok = np.array([f != 5 for f in range(8)])
        assert np.array_equal(i[ok], oi[ok]) and np.array_equal(bits(d)[ok], bits(od)[ok]), k
        assert np.array_equal(i[3], np.sort(i[3])) and i[3][0] == 7 or rotation == "dense"   # the crowd's ties come in id order
    # the same crowd in the SECOND group of a 12-query call (group-local indices in the selection kernel)
    q3 = np.concatenate([q[8:16], q2[:4]])
    od, oi = orc.adc_search(rot(q3), books, crowd, 100)
    d, i = idx.search(q3, 100, rotate=True)
    assert np.array_equal(i, oi) and np.array_equal(bits(d), bits(od))
    idx.close()
