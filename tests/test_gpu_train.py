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
