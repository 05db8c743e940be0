"""Codebook training: GPU Lloyd iteration == the oracle's, bit for bit.
The reference delegates to yael (not vendored): parity unpinned beyond the structure restated in
oracle/cvt_oracle.c (orc_kmeans / orc_opq_train, citing train_PQ_codebook.cpp:150-244)."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import cvt_amd
    cvt_amd.lib()  # raises if the HIP library is missing: there is no fallback
    return cvt_amd


def clustered(rng, n, d, k, spread=0.3):
    cen = rng.normal(size=(k, d)).astype(np.float32) * 3
    return (cen[rng.integers(0, k, n)] + rng.normal(size=(n, d)).astype(np.float32) * spread).astype(np.float32)


@pytest.mark.parametrize("n,d,k,niter", [(5000, 8, 256, 0), (3000, 128, 40, 0), (777, 3, 5, 0), (4000, 16, 64, 3),
                                         (2000, 200, 7, 0), (300, 8, 300, 0)])
def test_kmeans_parity(amd, orc, n, d, k, niter):
    rng = np.random.default_rng(n + d)
    x = clustered(rng, n, d, max(2, k // 2))
    x[5] = x[6]                     # exact duplicate rows: distance ties between rows
    if n > 1000:
        x[17] = np.nan              # a row no centroid can claim: assignment -1, excluded from every mean
    for seed in (1, 12345):
        oc, oa, oit = orc.kmeans(x, k, niter, seed)
        gc, ga, git = amd.kmeans(x, k, niter, seed)
        assert git == oit, (git, oit)
        assert np.array_equal(ga, oa)
        assert np.array_equal(bits(gc), bits(oc))


def test_kmeans_device_pointers_and_properties(amd):
    import torch
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    cen = torch.randn((256, 8), generator=g, device="cuda") * 4
    x = (cen[torch.randint(0, 256, (200_000,), generator=g, device="cuda")] + 0.2 * torch.randn((200_000, 8), generator=g, device="cuda")).contiguous()
    c, a, it = amd.kmeans(x, 256, 0, 1)
    assert it >= 1 and int(a.min()) >= 0 and int(a.max()) < 256
    # fixed point: every row is assigned to its nearest centroid, every non-empty centroid is its members' mean
    d2 = torch.cdist(x, c)
    assert torch.equal(d2.argmin(dim=1).to(torch.int32), a) or float((d2.gather(1, a.long()[:, None])[:, 0] - d2.min(dim=1).values).abs().max()) < 1e-5
    means = torch.zeros_like(c, dtype=torch.float64).index_add_(0, a.long(), x.double())
    cnt = torch.bincount(a.long(), minlength=256).clamp(min=1)[:, None]
    nonempty = torch.bincount(a.long(), minlength=256) > 0
    assert torch.allclose((means / cnt)[nonempty].float(), c[nonempty], atol=1e-5)


def test_opq_train_parity_and_use(amd, orc):
    """TrainPQ::IFVPQ structure: coarse k-means, residuals, per-sub-space k-means; the trained model drives
    encode + search end to end and matches the oracle's model bit for bit."""
    rng = np.random.default_rng(99)
    D, M, K, coarseK, n = 32, 4, 64, 6, 6000
    x = clustered(rng, n, D, 24)
    oc, ob = orc.opq_train(x, coarseK, M, K, 0, 1)
    gc, gb = amd.opq_train(x, coarseK, M, K, 0, 1)
    assert np.array_equal(bits(gc), bits(oc)) and np.array_equal(bits(gb), bits(ob))
    idx = amd.OpqIndex(gc, gb)
    lists, codes = idx.encode(x[:500])
    olists, ocodes = orc.pq_encode(x[:500], oc, ob)
    assert np.array_equal(lists, olists) and np.array_equal(codes, ocodes)


@pytest.mark.parametrize("n,d,k", [(6000, 128, 200), (5000, 64, 1000), (4200, 32, 64), (7000, 96, 77), (9000, 128, 2048)])
def test_assignment_matrix_core_filter(amd, orc, n, d, k):
    """nearest-centroid assignment through the bf16 matrix-core filter (assign_variant 2): coarse lists of encode and a
    whole k-means run equal the checker's, on clustered rows with duplicate / one-ulp-apart centroids, rows sitting
    exactly on centroids and exactly between two, huge, tiny and non-finite rows"""
    rng = np.random.default_rng(n + d + k)
    cen = (rng.normal(size=(k, d)) * 2).astype(np.float32)
    cen[k // 2] = cen[3]                                              # duplicate centroid: the lower index wins
    cen[k // 2 + 1] = np.nextafter(cen[4], np.float32(np.inf))        # one ulp apart
    x = (cen[rng.integers(0, k, n)] + 0.4 * rng.normal(size=(n, d))).astype(np.float32)
    x[:50] = cen[3]; x[50:100] = cen[4]
    x[100:150] = (cen[7] + cen[8]) / 2                                # equidistant up to rounding
    x[150:160] *= np.float32(1e5); x[160:170] *= np.float32(1e-25)
    x[170, 1] = np.inf; x[171] = np.nan; x[172, d - 1] = np.nan
    M = d // 8
    books = (rng.normal(size=(M, 16, 8))).astype(np.float32)
    try:
        for variant in (2, 1):
            amd.set_tuning("assign_variant", variant)
            idx = amd.OpqIndex(cen, books)
            lists, codes = idx.encode(x)
            ol, oc = orc.pq_encode(x, cen, books)
            assert np.array_equal(lists, ol), (variant, np.argwhere(lists != ol)[:10].ravel())
            assert np.array_equal(codes, oc)
        amd.set_tuning("assign_variant", 2)
        kk = min(k, 96)
        xs = x[np.isfinite(x).all(axis=1)][:5000]
        xs[9] = np.nan
        oc_, oa, oit = orc.kmeans(xs, kk, 4, 7)
        gc, ga, git = amd.kmeans(xs, kk, 4, 7)
        assert git == oit and np.array_equal(ga, oa) and np.array_equal(bits(gc), bits(oc_))
    finally:
        amd.set_tuning("assign_variant", 0)


def _correlated(rng, n, D):
    """rows whose dimensions are mixed and whose variances decay: what a learned rotation can untangle and balance"""
    A = rng.normal(size=(D, D))
    z = rng.normal(size=(n, D)) * np.exp(-np.arange(D) / (D / 5.0))
    return (z @ A.T).astype(np.float32)


def _distortion(orc, R, books, x):
    """mean squared quantisation error of the rows under (R, books), in the oracle's arithmetic"""
    M = books.shape[0]
    xr = orc.rotate_fma(R, x)
    _, codes = orc.pq_encode(xr, np.zeros((1, x.shape[1]), np.float32), books)
    y = np.concatenate([books[m][codes[:, m]] for m in range(M)], axis=1)
    return float(((xr.astype(np.float64) - y) ** 2).sum() / x.shape[0])


@pytest.mark.parametrize("n,D,M,K,outer,niter", [(3000, 32, 4, 16, 3, 4), (5000, 64, 8, 32, 2, 3), (2500, 128, 16, 64, 2, 2), (1100, 96, 4, 8, 4, 0)])
def test_opq_rotation_learning_parity(amd, orc, n, D, M, K, outer, niter):
    """the dense rotation learned by alternating k-means and orthogonal Procrustes.  Not in the reference (it only
    permutes): the specification is the oracle's orc_opq_learn_rotation, and the device path -- MFMA rotation, Lloyd iterations,
    reconstruction, X^T Y in the specified blocked order, host Jacobi -- has to reproduce R and the codebooks bit for bit; host-pointer and
    device-pointer entries alike; outer = 0 is plain PQ training under the identity."""
    import torch
    rng = np.random.default_rng(n + D)
    x = _correlated(rng, n, D)
    oR, ob_ = orc.opq_learn_rotation(x, M, K, outer, niter, 7)
    gR, gb = amd.opq_learn_rotation(x, M, K, outer, niter, 7)
    assert np.array_equal(bits(gR), bits(oR)), np.abs(gR - oR).max()
    assert np.array_equal(bits(gb), bits(ob_))
    tR, tb = amd.opq_learn_rotation(torch.from_numpy(x).cuda(), M, K, outer, niter, 7)
    assert np.array_equal(bits(tR.cpu().numpy()), bits(oR)) and np.array_equal(bits(tb.cpu().numpy()), bits(ob_))
    assert np.abs(gR.astype(np.float64) @ gR.astype(np.float64).T - np.eye(D)).max() < 1e-5     # a rotation
    iR, ib = amd.opq_learn_rotation(x, M, K, 0, niter, 7)
    assert np.array_equal(iR, np.eye(D, dtype=np.float32))
    # the learned rotation quantises these rows better than the identity (it is what the alternation minimises)
    assert _distortion(orc, gR, gb, x) < 0.9 * _distortion(orc, iR, ib, x)


def test_opq_rotation_learning_feeds_the_index(amd, orc):
    """R and books go straight into cvtmi_opq_create: rotate + encode + ADC search under the learned model equal the oracle's, and the
    search finds more exact neighbours than under the identity with the same budget of bytes"""
    rng = np.random.default_rng(3)
    n, D, M, K, k = 20_000, 64, 8, 256, 10
    both = _correlated(rng, n + 300, D)        # database and queries from the same (mixed, unbalanced) distribution
    x, q = both[:n].copy(), both[n:].copy()
    xd, qd = x.astype(np.float64), q.astype(np.float64)
    exact = np.argmin((qd ** 2).sum(1)[:, None] - 2 * qd @ xd.T + (xd ** 2).sum(1)[None, :], axis=1)
    hits = {}
    for outer in (0, 4):
        R, books = amd.opq_learn_rotation(x[:8000].copy(), M, K, outer, 4, 1)
        idx = amd.OpqIndex(np.zeros((1, D), np.float32), books, R=R)
        xr = idx.rotate(x)
        _, codes = idx.encode(xr)
        idx.add_codes(codes)
        d, i = idx.search(q, k, rotate=True)
        oxr = orc.rotate_fma(R, x)
        _, ocodes = orc.pq_encode(oxr, np.zeros((1, D), np.float32), books)
        assert np.array_equal(codes, ocodes)
        od, oi = orc.adc_search(orc.rotate_fma(R, q), books, ocodes, k)
        assert np.array_equal(i, oi) and np.array_equal(bits(d), bits(od))
        hits[outer] = float((i == exact[:, None]).any(axis=1).mean())
        idx.close()
    assert hits[4] > hits[0], hits
