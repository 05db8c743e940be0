"""GPU parity: exhaustive (flat) search, int8 scalar quantisation and the top-k merge, through the C ABI."""
import numpy as np
import pytest

from conftest import SQ8_NORM_GROUPS, bits

pytestmark = pytest.mark.gpu
IP, L2F, L2U8 = 0, 1, 2


@pytest.fixture(scope="module")
def amd():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import cvt_amd
    cvt_amd.lib()
    return cvt_amd


def test_flat_golden(amd, golden):
    f = golden.flat
    ix = amd.FlatIndex(IP, 128); ix.add(f["ip_db"])
    d, i = ix.search(f["ip_q"], 100)
    assert np.array_equal(i, f["ip_i"])
    assert np.array_equal(bits(d), bits(f["ip_d"]))       # bit-exact vs the reference's SSE build
    # labels that are not row numbers (ascending order is the documented device-side requirement)
    lab = f["ipl_labels"]
    order = np.argsort(lab)
    ix = amd.FlatIndex(IP, 128); ix.add(f["ip_db"][order], labels=lab[order])
    d, i = ix.search(f["ip_q"], 10)
    assert np.array_equal(i, f["ipl_i"]) and np.array_equal(bits(d), bits(f["ipl_d"]))
    ix = amd.FlatIndex(L2F, 64); ix.add(f["l2_db"])
    d, i = ix.search(f["l2_q"], 10)
    assert np.array_equal(i, f["l2_i"]) and np.array_equal(bits(d), bits(f["l2_d"]))
    for tag in ("u8a", "u8b", "u8c"):
        db = f[tag + "_db"]
        ix = amd.FlatIndex(L2U8, db.shape[1]); ix.add(db[:100]); ix.add(db[100:])
        d, i = ix.search(f[tag + "_q"], 10)
        assert d.dtype == np.int32
        assert np.array_equal(d, f[tag + "_d"]), tag
        assert np.array_equal(i, f[tag + "_i"]), tag


@pytest.mark.parametrize("metric,D", [(IP, 128), (IP, 20), (IP, 7), (L2F, 128), (L2F, 36), (L2F, 5), (L2U8, 512), (L2U8, 21)])
def test_flat_seeded(amd, orc, metric, D):
    rng = np.random.default_rng(metric * 100 + D)
    n, nq, k = 9000 + 5, 11, 37
    if metric == L2U8:
        db = rng.integers(0, 256, size=(n, D), dtype=np.uint8); q = rng.integers(0, 256, size=(nq, D), dtype=np.uint8)
    else:
        db = rng.normal(size=(n, D)).astype(np.float32); q = rng.normal(size=(nq, D)).astype(np.float32)
    db[4000] = db[10]; q[0] = db[10]
    ix = amd.FlatIndex(metric, D); ix.add(db)
    d, i = ix.search(q, k)
    od, odi, oi = orc.flat_search(metric, db, q, k)
    assert np.array_equal(i, oi)
    if metric == L2U8:
        assert np.array_equal(d, odi)
    else:
        assert np.array_equal(bits(d), bits(od))
    # single query (qtile 1 path) and k > n
    d1, i1 = ix.search(q[:1], k)
    assert np.array_equal(i1, oi[:1])
    small = amd.FlatIndex(metric, D); small.add(db[:5])
    d5, i5 = small.search(q[:2], 8)
    assert np.all(i5[:, 5:] == -1) and np.array_equal(i5[:, :5], orc.flat_search(metric, db[:5], q[:2], 5)[2])


@pytest.mark.parametrize("metric,D", [(IP, 128), (L2F, 96), (L2F, 33), (L2U8, 512), (L2U8, 50)])
def test_flat_any_k(amd, orc, metric, D):
    """searchKnn(query, k) takes any k (brutoforce.hpp:73-93): 129 ... 2048 through the exact kernels with the large selection buffer,
    one query and a batch, duplicate rows, k larger than the index"""
    rng = np.random.default_rng(metric * 1000 + D)
    n, nq = 12000 + 3, 9
    if metric == L2U8:
        db = rng.integers(0, 256, size=(n, D), dtype=np.uint8); q = rng.integers(0, 256, size=(nq, D), dtype=np.uint8)
    else:
        db = rng.normal(size=(n, D)).astype(np.float32); q = rng.normal(size=(nq, D)).astype(np.float32)
    db[4000:4200] = db[10]; q[0] = db[10]
    ix = amd.FlatIndex(metric, D); ix.add(db)
    for k in (129, 700, 2048):
        od, odi, oi = orc.flat_search(metric, db, q, k)
        for qq, sl in ((q, slice(None)), (q[:1], slice(0, 1))):
            d, i = ix.search(qq, k)
            assert np.array_equal(i, oi[sl]), (k, len(qq))
            if metric == L2U8:
                assert np.array_equal(d, odi[sl])
            else:
                assert np.array_equal(bits(d), bits(od[sl]))
    small = amd.FlatIndex(metric, D); small.add(db[:150])
    d, i = small.search(q[:2], 400)
    assert np.all(i[:, 150:] == -1) and np.array_equal(i[:, :150], orc.flat_search(metric, db[:150], q[:2], 150)[2])
    # k larger than the index at small k too (every kernel family pads alike): label -1, the +inf bit pattern in the distance field
    tiny = amd.FlatIndex(metric, D); tiny.add(db[:52])
    for nq2, k2 in ((9, 80), (1, 60), (9, 128)):
        d, i = tiny.search(q[:nq2], k2)
        od, odi, oi = orc.flat_search(metric, db[:52], q[:nq2], k2)
        assert np.array_equal(i, oi) and np.all(i[:, 52:] == -1), (nq2, k2)
        assert np.array_equal(d, odi) if metric == L2U8 else np.array_equal(bits(d), bits(od)), (nq2, k2)
        assert np.all(np.asarray(d[:, 52:]).view(np.uint32) == 0x7f800000)
    with pytest.raises(amd.CvtmiError):
        ix.search(q, 2049)


@pytest.mark.parametrize("D,nq,k", [(512, 200, 10), (512, 70, 16), (256, 100, 40), (128, 40, 48), (512, 150, 100),
                                     (64, 300, 1), (512, 33, 5), (96, 129, 12), (512, 5, 10), (128, 7, 128)])
def test_flat_u8_mfma_query_tiles(amd, orc, D, nq, k):
    """i8 matrix-core path with 32 / 64 / 128 queries per workgroup (chosen by k and nq): bit-exact distances
    and ids incl. duplicate rows (id tie-break) and rows sorted by decreasing distance to a query (every row
    beats the running threshold: buffer-overflow + retry path)."""
    rng = np.random.default_rng(D * 7 + nq)
    n = 20000 + 3
    db = rng.integers(0, 256, size=(n, D), dtype=np.uint8)
    q = rng.integers(0, 256, size=(nq, D), dtype=np.uint8)
    db[7000] = db[11]; db[15000] = db[11]; q[1] = db[11]
    # rows 0..4999 ordered by decreasing distance to q[0]
    dist0 = ((db[:5000].astype(np.int64) - q[0].astype(np.int64)) ** 2).sum(axis=1)
    db[:5000] = db[:5000][np.argsort(-dist0, kind="stable")]
    ix = amd.FlatIndex(L2U8, D); ix.add(db)
    d, i = ix.search(q, k)
    _, odi, oi = orc.flat_search(L2U8, db, q, k)
    assert np.array_equal(d, odi), (D, nq, k)
    assert np.array_equal(i, oi), (D, nq, k)


def test_flat_u8_many_splits_parity(amd, orc):
    """200 K rows -> 24 row splits of one 128-query workgroup column: the splits exchange their k-th best and, for
    k <= 16, their minima (k slots per query) to tighten each other's filter; the merged result must still be
    the oracle's, bit for bit, incl. duplicate rows in different splits."""
    rng = np.random.default_rng(31)
    n, D, nq = 200_000, 128, 100
    db = rng.integers(0, 256, size=(n, D), dtype=np.uint8)
    q = rng.integers(0, 256, size=(nq, D), dtype=np.uint8)
    db[150_000] = db[11]; db[199_999] = db[11]; q[1] = db[11]      # ties across splits: ids decide
    q[2] = db[77_777]
    ix = amd.FlatIndex(L2U8, D); ix.add(db)
    for k in (10, 3, 16):
        d, i = ix.search(q, k)
        _, odi, oi = orc.flat_search(L2U8, db, q, k)
        assert np.array_equal(d, odi), k
        assert np.array_equal(i, oi), k


def test_flat_u8_row_tile_empty_last_split(amd, orc):
    """1 M rows x 1000 queries on the row-tile kernels: the planner asks for 120 splits, whole tiles make them 8448 rows each, and
    the last split starts PAST the index (119 x 8448 > 1 M).  It used to fetch its first tiles from there (a memory fault at 512-d);
    now an empty split reads row 0 and reports nothing.  Against the default route on all queries and the checker on a few."""
    rng = np.random.default_rng(77)
    n, D, nq, k = 1_000_000, 64, 1000, 10
    db = rng.integers(0, 256, size=(n, D), dtype=np.uint8)
    q = rng.integers(0, 256, size=(nq, D), dtype=np.uint8)
    ix = amd.FlatIndex(L2U8, D); ix.add(db)
    try:
        amd.set_tuning("flat_variant", 1)
        d1, i1 = ix.search(q, k)
    finally:
        amd.set_tuning("flat_variant", 0)
    d0, i0 = ix.search(q, k)
    assert np.array_equal(d1, d0) and np.array_equal(i1, i0)
    _, odi, oi = orc.flat_search(L2U8, db, q[:4], k)
    assert np.array_equal(d1[:4], odi) and np.array_equal(i1[:4], oi)


def test_flat_full_size_u8_property(amd):
    """Config 3 shape (512-d uint8) at a size that needs row splits: self-queries come back first with
    distance 0 and the result is invariant to how the rows were appended."""
    import torch
    n, D = 400_000, 512
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    db = torch.randint(0, 256, (n, D), generator=g, device="cuda", dtype=torch.uint8)
    ix = amd.FlatIndex(L2U8, D); ix.add(db)
    rows = torch.tensor([0, 12345, n - 1], device="cuda")
    d, i = ix.search(db[rows].contiguous(), 10)
    assert torch.equal(i[:, 0], rows) and torch.all(d[:, 0] == 0)
    assert torch.all(d[:, 1:] > 0) and torch.all(d[:, 1:] >= d[:, :-1])
    ix2 = amd.FlatIndex(L2U8, D)
    for a in range(0, n, 150_000):
        ix2.add(db[a:a + 150_000].contiguous())
    d2, i2 = ix2.search(db[rows].contiguous(), 10)
    assert torch.equal(d2, d) and torch.equal(i2, i)


def test_config3_full_size(amd, orc):
    """BASELINE config 3 at its real size: 10 M x 512-d int8 codes (SQ8 of CNN-like features: ReLU'd Gaussians, L2-normalised,
    encoded on device), brute-force L2 top-10.  Checked against the oracle's L2SqrI loop over ALL 10 M rows for a sample of the
    queries (host threads split the queries), for nq = 1, a mid-size and a large batch -- the three kernel regimes -- and through
    size-independent properties: self-queries come back first at distance 0, distances ascend, every batch size agrees."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    n, D, k = 10_000_000, 512, 10
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(3)
    train = torch.randn((1 << 18, D), generator=g, device=dev).clamp_(min=0)
    vmin, vdiff = amd.sq8_train(train, l2norm=True)
    ix = amd.FlatIndex(L2U8, D)
    host = np.empty((n, D), dtype=np.uint8)
    chunk = 1 << 20
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        g.manual_seed(1000 + a // chunk)
        c = amd.sq8_encode(vmin, vdiff, torch.randn((b - a, D), generator=g, device=dev).clamp_(min=0), l2norm=True)
        if a == 0:
            c[77] = c[5]; c[900_000] = c[5]                       # exact duplicates: (distance, label) ties
        ix.add(c)
        host[a:b] = c.cpu().numpy()
    assert ix.ntotal == n
    g.manual_seed(77)
    q = amd.sq8_encode(vmin, vdiff, torch.randn((4096, D), generator=g, device=dev).clamp_(min=0), l2norm=True)
    q[0] = torch.from_numpy(host[5]).to(dev)                       # lands on the duplicates
    q[1] = torch.from_numpy(host[n - 1]).to(dev)
    d_big, i_big = ix.search(q, k)
    d_mid, i_mid = ix.search(q[:1000].contiguous(), k)
    assert torch.equal(d_mid, d_big[:1000]) and torch.equal(i_mid, i_big[:1000])
    for j in (0, 1, 2):
        d1, i1 = ix.search(q[j:j + 1].contiguous(), k)
        assert torch.equal(d1[0], d_big[j]) and torch.equal(i1[0], i_big[j])
    assert i_big[0, :3].tolist() == [5, 77, 900_000] and torch.all(d_big[0, :3] == 0)
    assert int(i_big[1, 0]) == n - 1 and int(d_big[1, 0]) == 0
    assert torch.all(d_big[:, 1:] >= d_big[:, :-1])
    cs = 16
    qh = q[:cs].cpu().numpy()
    with ThreadPoolExecutor(max_workers=cs) as ex:                 # ctypes releases the GIL: one query per host thread
        parts = list(ex.map(lambda j: orc.flat_search(L2U8, host, qh[j:j + 1], k), range(cs)))
    odi = np.concatenate([p[1] for p in parts]); oi = np.concatenate([p[2] for p in parts])
    assert np.array_equal(i_big[:cs].cpu().numpy(), oi)
    assert np.array_equal(d_big[:cs].cpu().numpy(), odi)
    ix.close()


@pytest.mark.parametrize("d", [512, 256])
def test_sq8_train_wave_kernel(amd, orc, d):
    """Training with normalisation at d = 256 / 512 and >= 4096 rows takes the wave-per-row kernel: min / max - min must equal
    the oracle's bit for bit, including rows whose norm needs the index-order sum (huge dynamic range), zero rows and a
    non-finite row."""
    import torch
    rng = np.random.default_rng(d)
    n = 20_000 + 37
    x = np.abs(rng.normal(size=(n, d))).astype(np.float32)
    x[5] = 0
    x[17] *= 1e-30; x[18] *= 1e18
    x[100:600] *= np.exp(rng.normal(size=(500, 1)) * 8).astype(np.float32)      # wide range of norms
    x[700:1200] = (x[700:1200] * np.exp(rng.normal(size=(500, d)) * 6)).astype(np.float32)   # wide range inside a row
    vmin, vdiff = amd.sq8_train(torch.from_numpy(x).cuda(), l2norm=True)
    ovmin, ovdiff = orc.sq8_train(x, l2norm=True)
    assert np.array_equal(bits(vmin.cpu().numpy()), bits(ovmin)) and np.array_equal(bits(vdiff.cpu().numpy()), bits(ovdiff))
    x[9, 3] = np.inf
    vmin, vdiff = amd.sq8_train(torch.from_numpy(x).cuda(), l2norm=True)
    ovmin, ovdiff = orc.sq8_train(x, l2norm=True)
    assert np.array_equal(bits(vmin.cpu().numpy()), bits(ovmin)) and np.array_equal(bits(vdiff.cpu().numpy()), bits(ovdiff))


@pytest.mark.parametrize("d", [512, 256])
def test_sq8_train_unnormalised_huge_magnitudes(amd, orc, d):
    """without normalisation the wave kernel's decision filter scales elements by 2^60 -- an extreme of 2^68 or more sent
    its own threshold (and the scaled elements) to infinity, and strictly larger elements were never compared.  Columns whose
    maximum / minimum climbs through 1e20 .. 3e38 in steps, early and late in the table, filter on and off, against the chain."""
    import torch
    rng = np.random.default_rng(d + 99)
    n = 12_000 + 5
    x = rng.normal(size=(n, d)).astype(np.float32)
    steps = np.float32([1e19, 3e20, 2e21, 5e24, 1e30, 3e38])
    for c in range(0, 24):   # rising maxima in columns 0..11, falling minima in 12..23, spread over the rows of many waves
        rows = rng.choice(n, size=len(steps), replace=False); rows.sort()
        x[rows, c] = steps if c < 12 else -steps
    x[n - 1, 30] = np.float32(2.5e38); x[0, 31] = np.float32(-2.5e38)   # the last / the first row holds the extreme
    ovm, ovd = orc.sq8_train(x.copy(), l2norm=False)
    for filt in (1, 0):
        amd.set_tuning("sq8_filter", filt)
        try:
            tv, td = amd.sq8_train(torch.from_numpy(x.copy()).cuda(), l2norm=False)
        finally:
            amd.set_tuning("sq8_filter", 1)
        assert np.array_equal(bits(tv.cpu().numpy()), bits(ovm)), (filt, np.flatnonzero(bits(tv.cpu().numpy()) != bits(ovm))[:8])
        assert np.array_equal(bits(td.cpu().numpy()), bits(ovd)), (filt, np.flatnonzero(bits(td.cpu().numpy()) != bits(ovd))[:8])


@pytest.mark.parametrize("d", [512, 256])
def test_sq8_encode_wave_kernel(amd, orc, d):
    """Encode at d = 256 / 512 and >= 4096 rows takes the wave-per-row kernel, with and without normalisation: codes and the
    normalised rows written back must equal the oracle's bit for bit (and the tile kernel's), including rows whose norm needs the
    index-order sum, zero rows, a non-finite row, a column with vdiff = 0 and a ragged tail."""
    import torch
    rng = np.random.default_rng(d + 1)
    n = 20_000 + 37
    x = np.abs(rng.normal(size=(n, d))).astype(np.float32)
    x[5] = 0
    x[17] *= 1e-30; x[18] *= 1e18
    x[100:600] *= np.exp(rng.normal(size=(500, 1)) * 8).astype(np.float32)
    x[700:1200] = (x[700:1200] * np.exp(rng.normal(size=(500, d)) * 6)).astype(np.float32)
    x[9, 3] = np.inf
    x[:, 7] = 0.25                                            # constant column
    for l2 in (True, False):
        base = x.copy() if l2 else (x / np.maximum(np.linalg.norm(np.where(np.isfinite(x), x, 0), axis=1, keepdims=True), 1e-12)).astype(np.float32)
        ovmin, ovdiff = orc.sq8_train(base[20:], l2norm=l2)   # train without the hard rows: codes clamp outside [vmin, vmin + vdiff]
        oc, ox = orc.sq8_encode(ovmin, ovdiff, base, l2norm=l2)
        out = {}
        try:
            for wave in (1, 0):
                amd.set_tuning("sq8_encode_wave", wave)
                xt = torch.from_numpy(base.copy()).cuda()
                codes = amd.sq8_encode(torch.from_numpy(ovmin).cuda(), torch.from_numpy(ovdiff).cuda(), xt, l2norm=l2)
                out[wave] = (codes.cpu().numpy(), xt.cpu().numpy())
        finally:
            amd.set_tuning("sq8_encode_wave", 1)
        assert np.array_equal(out[1][0], out[0][0]) and np.array_equal(bits(out[1][1]), bits(out[0][1]))
        assert np.array_equal(out[1][0], oc)
        assert np.array_equal(bits(out[1][1]), bits(ox))


@pytest.mark.parametrize("group", SQ8_NORM_GROUPS)
def test_sq8_normalisation_golden(amd, golden, group):
    """a-Q / a-T against the REFERENCE's own bits: rows normalised by MathUtil::L2NormArray (utils/math_util.h:29-39 == Int8Quan::
    L2NormalizeVector, int8_quan.cc:46-56; compiled in place, tests/golden/sq8_norm_golden.npz).  cvtmi_sq8_encode(l2norm = 1)
    writes the normalised rows back as the reference does to its caller's buffer: those rows must be the golden rows bit for bit --
    host-pointer entry, device entry, and (d = 256 / 512 at >= 4096 rows) the wave-per-row kernel; training = min / max - min of them."""
    import torch
    x, want = golden.sq8_norm[group + "_x"], golden.sq8_norm[group + "_array"]
    n, d = x.shape
    vmin, vdiff = np.zeros(d, np.float32), np.ones(d, np.float32)
    xg = x.copy()
    amd.sq8_encode(vmin, vdiff, xg, l2norm=True)
    assert np.array_equal(bits(xg), bits(want)), "host-pointer entry"
    reps = -(-4200 // n)                       # enough rows for the wave-per-row kernels where the width has one
    xt = torch.from_numpy(np.tile(x, (reps, 1))).cuda()
    codes = amd.sq8_encode(torch.from_numpy(vmin).cuda(), torch.from_numpy(vdiff).cuda(), xt, l2norm=True)
    assert np.array_equal(bits(xt.cpu().numpy()), bits(np.tile(want, (reps, 1)))), "device entry"
    with np.errstate(all="ignore"):   # codes = (int)(255 * clamp(row)) of the golden rows (vmin 0, vdiff 1: x / 1 is exact)
        expect = (np.float32(255) * np.clip(np.nan_to_num(want, nan=0.0), 0, 1)).astype(np.int32).astype(np.uint8)
    ok = np.all(np.isfinite(want), axis=1)
    assert np.array_equal(codes.cpu().numpy()[:n][ok], expect[ok])
    if ok.any():
        w = want[ok]
        for rows in (x[ok], np.tile(x[ok], (-(-4200 // int(ok.sum())), 1))):
            tv, td = amd.sq8_train(torch.from_numpy(rows.copy()).cuda(), l2norm=True)
            tv, td = tv.cpu().numpy(), td.cpu().numpy()
            # (a column whose minimum is zero and that holds zeros of both signs takes the sign of the FIRST zero in row order, as the
            # sequential loop does: test_sq8_train_sign_of_a_zero_minimum)
            first = np.array([w[np.argmax(w[:, c] == w[:, c].min()), c] for c in range(w.shape[1])], np.float32)   # first occurrence of the minimum
            assert np.array_equal(bits(tv), bits(first))
            assert np.array_equal(bits(td), bits(w.max(axis=0) - w.min(axis=0)))


def _same_min(a, b):
    """column minima, bit for bit -- the sign of a zero minimum included (test_sq8_train_sign_of_a_zero_minimum)"""
    return np.array_equal(bits(a), bits(b))


@pytest.mark.parametrize("d,n", [(512, 9000), (256, 70_000), (64, 500), (300, 2000), (12, 64)])
@pytest.mark.parametrize("l2", [True, False])
def test_sq8_train_sign_of_a_zero_minimum(amd, orc, d, n, l2):
    """A column whose minimum is zero and that holds zeros of both signs: the loop this restates (faiss train_NonUniform, RS_minmax:
    strict '<' in row order behind sq_train.cpp:100) keeps the FIRST zero it meets; the device reduction orders -0.0 below +0.0.  The
    training call therefore looks such columns up again (sq8_zero_first_kernel).  Every kernel family (wave-per-row with sample pass,
    tile, two-pass) and both first-zero signs, zeros that only appear through an infinite norm, and columns without any zero."""
    import torch
    rng = np.random.default_rng(d + n + int(l2))
    x = np.abs(rng.normal(size=(n, d))).astype(np.float32)          # non-negative: the minimum of a column with a zero IS zero
    z = rng.random(size=(n, d)) < 0.05
    x[z] = 0.0
    x[z & (rng.random(size=(n, d)) < 0.5)] = -0.0
    x[:, 0] = np.abs(x[:, 0]) + 0.1                                 # no zero at all
    x[:, 1] = 1.0; x[n // 2, 1] = -0.0; x[n // 2 + 1, 1] = 0.0      # first zero negative
    x[:, 2] = 1.0; x[n // 3, 2] = 0.0; x[n // 3 + 5, 2] = -0.0      # first zero positive, a negative one later
    x[:, 3] = 1.0; x[n - 1, 3] = -0.0                               # a single negative zero, in the last row
    if l2:
        x[7] = np.float32(1e30) * np.sign(rng.normal(size=d)).astype(np.float32)   # norm overflows to inf: every quotient of the row is a signed zero
        x[:7, 4] = 1.0; x[8:, 4] = 1.0                              # column 4's only zero is that row's
    ovm, ovd = orc.sq8_train(x.copy(), l2norm=l2)
    for filt in (1, 0):
        amd.set_tuning("sq8_filter", filt)
        try:
            tv, td = amd.sq8_train(torch.from_numpy(x.copy()).cuda(), l2norm=l2)
        finally:
            amd.set_tuning("sq8_filter", 1)
        assert np.array_equal(bits(tv.cpu().numpy()), bits(ovm)), (filt, np.flatnonzero(bits(tv.cpu().numpy()) != bits(ovm))[:8])
        assert np.array_equal(bits(td.cpu().numpy()), bits(ovd))
    hv, hd = amd.sq8_train(x.copy(), l2norm=l2)                     # host-pointer entry
    assert np.array_equal(bits(hv), bits(ovm)) and np.array_equal(bits(hd), bits(ovd))


@pytest.mark.parametrize("d,kind", [(d, k) for d in (512, 256) for k in ("relu", "signed", "wide")] +
                         [(768, "relu"), (1024, "wide"), (1536, "signed"), (2048, "relu"), (2048, "wide")] +
                         [(128, "relu"), (128, "wide"), (64, "signed"), (64, "relu"), (64, "wide")])   # round 6: several rows per wave (sq8_*_group_f_kernel)
def test_sq8_decision_filter_equals_the_chain(amd, orc, d, kind):
    """Round 5: the wave kernels decide most code bytes / column extremes from a bounded approximation and run the reference's chain
    (two correctly rounded divisions + the byte, int8_quan.cc:46-56, :79-92) only where that cannot decide.  Codes, written-back rows and
    trained ranges must equal the chain's (filter off) and the oracle's, on: half-zero rows (exact zeros sit ON an integer when vmin = 0),
    signed rows, rows spanning 40 binades, clamped values on both sides, and columns the bound does not cover (vdiff 0 / tiny / huge,
    |vmin| >> vdiff, a NaN range), zero rows, huge rows, a non-finite row, negative zeros."""
    import torch
    rng = np.random.default_rng(d + len(kind))
    n = (30_000 if d <= 512 else 9_000) + 41     # (round 5: the same kernels take rows of 768 ... 2048 floats, fewer rows in flight per wave)
    x = rng.normal(size=(n, d)).astype(np.float32)
    if kind == "relu":
        x = np.maximum(x, 0)
    elif kind == "wide":
        x = (x * np.exp2(rng.integers(-20, 20, size=(n, d)))).astype(np.float32)
        x[rng.random(size=(n, d)) < 0.2] = 0
    x[3] = 0; x[4, 5] = -0.0; x[11] *= 1e30; x[12] *= 1e-30; x[13, 7] = np.inf; x[14] = -0.0
    x[20:40, :8] = -0.0
    for l2 in (1, 2, 0):       # 1 = the reference's in-place normalisation, 2 = normalised codes, rows left alone, 0 = turn_off_l2norm
        xfin = x[50:][np.all(np.isfinite(x[50:]), axis=1)]
        ovmin, ovdiff = orc.sq8_train(xfin.copy(), l2norm=(l2 != 0))       # trained without the first rows: some of them clamp
        hv, hd = ovmin.copy(), ovdiff.copy()
        hd[1] = 0.0; hd[2] = 1e-41; hd[3] = 1e30; hv[4] = 50.0; hd[4] = 1e-3; hd[5] = np.nan; hv[6] = -0.0; hd[7] = -abs(hd[7])
        hv[8] = hv[8] + 0.3 * hd[8]; hd[9] *= 0.4                            # values below vmin / above vmin + vdiff: both clamps
        oc, ox = orc.sq8_encode(hv, hd, x, l2norm=(l2 != 0))
        got = {}
        try:
            for filt in (1, 2, 0):                                           # 2 = the filter kernel with the ds_bpermute butterfly for its wave sums
                amd.set_tuning("sq8_filter", 1 if filt else 0)
                amd.set_tuning("sq8_flags", 0 if filt == 2 else 1)
                xt = torch.from_numpy(x.copy()).cuda()
                codes = amd.sq8_encode(torch.from_numpy(hv).cuda(), torch.from_numpy(hd).cuda(), xt, l2norm=l2)
                got[filt] = (codes.cpu().numpy(), xt.cpu().numpy())
        finally:
            amd.set_tuning("sq8_filter", 1); amd.set_tuning("sq8_flags", 1)
        assert np.array_equal(got[2][0], got[1][0]) and np.array_equal(bits(got[2][1]), bits(got[1][1])), "wave-sum flavours"
        fin = np.all(np.isfinite(ox), axis=1)                                # (int) NaN is undefined in the reference itself
        fin_cols = np.isfinite(hd)
        assert np.array_equal(got[1][0], got[0][0]), (kind, d, l2, "filter vs chain")
        assert np.array_equal(got[1][0][fin][:, fin_cols], oc[fin][:, fin_cols]), (kind, d, l2, "vs oracle")
        if l2 == 1:
            assert np.array_equal(bits(got[1][1]), bits(ox)) and np.array_equal(bits(got[0][1]), bits(ox))
        else:
            assert np.array_equal(bits(got[1][1]), bits(x))
        if l2 == 0:   # (no normalisation: every finite ROW is comparable, whatever its norm would have been)
            fin = np.all(np.isfinite(x), axis=1)
            assert np.array_equal(got[1][0][fin][:, fin_cols], oc[fin][:, fin_cols])
    # training: extremes of the (normalised) rows, sample pass + seeded main pass; without normalisation the wave kernels take the rows
    # wider than the tile kernel's 512 floats
    xf = x[np.all(np.isfinite(x), axis=1)]
    times = 3 if d <= 512 else 8                                             # past the 8 x 8192 rows that switch the sample pass on
    for l2 in (True, False):
        ovmin, ovdiff = orc.sq8_train(xf.copy(), l2norm=l2)
        res = {}
        try:
            for filt in (1, 2, 0):
                amd.set_tuning("sq8_filter", 1 if filt else 0)
                amd.set_tuning("sq8_flags", 0 if filt == 2 else 1)
                for rows in (xf, np.tile(xf, (times, 1))):
                    tv, td = amd.sq8_train(torch.from_numpy(rows.copy()).cuda(), l2norm=l2)
                    res[(filt, len(rows))] = (tv.cpu().numpy(), td.cpu().numpy())
        finally:
            amd.set_tuning("sq8_filter", 1); amd.set_tuning("sq8_flags", 1)
        for key, (tv, td) in res.items():
            assert _same_min(tv, ovmin) and np.array_equal(bits(td), bits(ovdiff)), (kind, d, l2, key)


@pytest.mark.parametrize("d", [64, 512, 2048])
def test_sq8_one_vector_per_call(amd, orc, d):
    """The reference's call shape: Int8Encode / Int8Decode on ONE feature vector (int8_quan.cc:72-132).  Small host-pointer calls run out
    of a page-locked scratch area (round 5; sq8_host_small 0 = the allocate-copy-free form): same codes, rows and decoded values."""
    rng = np.random.default_rng(d)
    xs = np.abs(rng.normal(size=(300, d))).astype(np.float32)
    vmin, vdiff = orc.sq8_train(xs.copy(), l2norm=True)
    try:
        for small in (1, 0):
            amd.set_tuning("sq8_host_small", small)
            for n in (1, 3):
                for l2 in (True, False):
                    x = xs[7:7 + n].copy()
                    codes = amd.sq8_encode(vmin, vdiff, x, l2norm=l2)
                    oc, ox = orc.sq8_encode(vmin, vdiff, xs[7:7 + n], l2norm=l2)
                    assert np.array_equal(codes, oc) and np.array_equal(bits(x), bits(ox)), (d, small, n, l2)
                dec = amd.sq8_decode(vmin, vdiff, codes)
                assert np.array_equal(bits(dec), bits(orc.sq8_decode(vmin, vdiff, codes))), (d, small, n)
    finally:
        amd.set_tuning("sq8_host_small", 1)


def test_sq8_parity(amd, orc, golden):
    rng = np.random.default_rng(8)
    for d in (64, 512, 300):
        x = np.abs(rng.normal(size=(777, d))).astype(np.float32)
        x[0, :64] = golden.sq8["int8_quan_test_x"]
        x[5] = 0
        vmin, vdiff = amd.sq8_train(x, l2norm=True)
        ovmin, ovdiff = orc.sq8_train(x, l2norm=True)
        assert np.array_equal(bits(vmin), bits(ovmin)) and np.array_equal(bits(vdiff), bits(ovdiff))
        vdiff2 = vdiff.copy(); vdiff2[1] = 0
        for l2 in (True, False):
            xg = x.copy()
            codes = amd.sq8_encode(vmin, vdiff2, xg, l2norm=l2)
            ocodes, ox = orc.sq8_encode(vmin, vdiff2, x, l2norm=l2)
            assert np.array_equal(codes, ocodes), (d, l2)
            assert np.array_equal(bits(xg), bits(ox)), "in-place normalisation differs"
        dec = amd.sq8_decode(vmin, vdiff, codes)
        assert np.array_equal(bits(dec), bits(orc.sq8_decode(vmin, vdiff, codes)))
        v3, d3 = amd.sq8_train(x, l2norm=False)
        o3, od3 = orc.sq8_train(x, l2norm=False)
        assert np.array_equal(bits(v3), bits(o3)) and np.array_equal(bits(d3), bits(od3))


def test_sq8_parity_hard_values(amd, orc):
    """The fast correctly-rounded division (reciprocal + 2 fma) and its guards: signed values over 60 binades,
    zeros of both signs, denormals, huge values, divisors with an all-ones significand, tiny and zero ranges,
    widths on and off the single-pass kernel, row counts off the 64-row tile."""
    rng = np.random.default_rng(88)
    for d, n in ((512, 1000), (256, 130), (4, 65), (12, 64), (100, 333), (516, 70), (1024, 40), (7, 5)):
        x = (rng.normal(size=(n, d)) * np.exp2(rng.integers(-30, 30, size=(n, d)))).astype(np.float32)
        x[rng.random(size=(n, d)) < 0.3] = 0.0
        x[rng.random(size=(n, d)) < 0.05] = -0.0
        x[1] = 0.0                                   # zero norm: den = 1e-12
        x[2] = np.float32(1e-30) * rng.normal(size=d).astype(np.float32)   # products underflow: tiny den
        x[3, 0] = np.float32(3e18)                   # one huge element
        x[4] = np.float32(1e-42)                     # denormals
        if n > 9:
            x[9] = 0.0; x[9, d // 2] = np.frombuffer(np.uint32(0x3fffffff).tobytes(), np.float32)[0]  # norm = 1.9999999
        vmin, vdiff = amd.sq8_train(x, l2norm=True)
        ovmin, ovdiff = orc.sq8_train(x, l2norm=True)
        assert np.array_equal(bits(vmin), bits(ovmin)) and np.array_equal(bits(vdiff), bits(ovdiff)), (d, n)
        vdiff2 = vdiff.copy()
        vdiff2[0] = 0.0
        vdiff2[1 % d] = np.frombuffer(np.uint32(0x3d7fffff).tobytes(), np.float32)[0]   # all-ones significand
        vdiff2[2 % d] = np.float32(1e-41)
        vdiff2[3 % d] = np.float32(1e30)
        for l2 in (True, False):
            xg = x.copy()
            codes = amd.sq8_encode(vmin, vdiff2, xg, l2norm=l2)
            ocodes, ox = orc.sq8_encode(vmin, vdiff2, x, l2norm=l2)
            assert np.array_equal(bits(xg), bits(ox)), ("in-place normalisation differs", d, n, l2)
            assert np.array_equal(codes, ocodes), (d, n, l2)
        dec = amd.sq8_decode(vmin, vdiff2, codes)
        assert np.array_equal(bits(dec), bits(orc.sq8_decode(vmin, vdiff2, codes))), (d, n)


def test_sq8_device_roundtrip_full_width(amd):
    """Config 3 width (512-d) on device pointers: decode(encode(x)) lands within one bucket of x."""
    import torch
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    x = torch.randn((200_000, 512), generator=g, device="cuda").relu_()
    vmin, vdiff = amd.sq8_train(x, l2norm=True)
    xn = x.clone()
    codes = amd.sq8_encode(vmin, vdiff, xn, l2norm=True)
    torch.cuda.synchronize()
    assert torch.allclose(xn.norm(dim=1), torch.ones(x.shape[0], device="cuda"), atol=1e-5)
    dec = amd.sq8_decode(vmin, vdiff, codes)
    assert torch.all((dec - xn).abs() <= vdiff / 255 * 1.0001 + 1e-7)
    assert int(codes.max()) == 255


def test_merge_parity(amd, orc):
    rng = np.random.default_rng(21)
    for nq, L, k in ((5, 2, 100), (3, 8, 100), (4, 64, 10), (2, 300, 128), (1, 1, 1)):
        d = np.sort(rng.integers(0, 50, size=(nq, L, k)).astype(np.float32) - 10.0, axis=2)  # negative dists too, many ties
        ids = np.empty((nq, L, k), dtype=np.int64)
        for l in range(L):
            ids[:, l, :] = l * 100000 + np.sort(rng.choice(100000, k, replace=False))
        # restore the (dist,id) order inside each list and pad some tails
        for q in range(nq):
            for l in range(L):
                o = np.lexsort((ids[q, l], d[q, l])); d[q, l] = d[q, l][o]; ids[q, l] = ids[q, l][o]
        ids[:, L // 2, k - k // 3:] = -1
        md, mi = amd.topk_merge(d, ids, k)
        od, oi = orc.merge_topk(d, ids, k)
        assert np.array_equal(mi, oi), (nq, L, k)
        assert np.array_equal(bits(md), bits(od))


@pytest.mark.parametrize("metric,D,k", [(IP, 128, 100), (L2F, 128, 100), (IP, 64, 10), (L2F, 32, 1), (L2F, 96, 128), (IP, 128, 5)])
def test_flat_f32_matrix_core_filter(amd, orc, metric, D, k):
    """fp32 search through the bf16 matrix-core filter (flat_variant 2) against the exact kernels (flat_variant 1) on the
    whole batch, and against the checker on a few queries: clustered rows with exact duplicates of rows and of queries
    (ties broken by row), appends between searches, unit-norm and large-magnitude data"""
    rng = np.random.default_rng(D * 7 + k + metric)
    n, nq = 150_000, 200
    cen = rng.normal(size=(500, D)).astype(np.float32)
    x = (cen[rng.integers(0, 500, n)] + 0.5 * rng.normal(size=(n, D))).astype(np.float32)
    if metric == IP:
        x /= np.linalg.norm(x, axis=1, keepdims=True)
    else:
        x *= np.float32(37.0)
    x[100_000:100_300] = x[5]; x[140_000:140_050] = x[70_000]          # duplicates inside and outside the leading sample
    q = x[rng.integers(0, n, nq)] + (0.05 * rng.normal(size=(nq, D))).astype(np.float32)
    q[:20] = x[rng.integers(0, n, 20)]                                  # queries that are rows: zero distances, ties
    q[20] = x[5]
    q = np.ascontiguousarray(q, np.float32)
    try:
        res = {}
        for v in (2, 1):
            amd.set_tuning("flat_variant", v)
            ix = amd.FlatIndex(metric, D); ix.add(x[:130_000]); ix.add(x[130_000:])
            res[v] = ix.search(q, k)
            used, worst = ix.last_search()
            assert used == (1 if v == 2 else 0) and (v == 1 or 0 < worst < 24 * k + 1024), (used, worst)
            if v == 2:   # append after a search: the operand copy is rebuilt
                ix.add(x[:1000] * np.float32(0.5))
                d2, i2 = ix.search(q, k)
                assert ix.last_search()[0]
                amd.set_tuning("flat_variant", 1)
                d1, i1 = ix.search(q, k)
                assert np.array_equal(i2, i1) and np.array_equal(bits(d2), bits(d1))
        assert np.array_equal(res[2][1], res[1][1])
        assert np.array_equal(bits(res[2][0]), bits(res[1][0]))
        od, _, oi = orc.flat_search(metric, x, q[:24], k, flavour=4 if metric == IP else 8)
        assert np.array_equal(res[2][1][:24], oi) and np.array_equal(bits(res[2][0][:24]), bits(od))
    finally:
        amd.set_tuning("flat_variant", 0)


def test_flat_f32_filter_gives_up_cleanly(amd):
    """rows sorted by similarity to the queries (the leading sample says nothing about the rest: candidate lists
    overflow), a non-finite row, non-finite queries, labels: the exact path answers, same results"""
    rng = np.random.default_rng(11)
    n, D, nq, k = 140_000, 128, 80, 10
    x = rng.normal(size=(n, D)).astype(np.float32)
    q = rng.normal(size=(nq, D)).astype(np.float32)
    order = np.argsort(-(x @ q[0]))[::-1]        # best matches of query 0 last
    xs = np.ascontiguousarray(x[order])
    labels = (np.arange(n, dtype=np.int64) * 3 + 5)
    cases = [("sorted", xs, q, None), ("labels", x, q, labels)]
    xn = x.copy(); xn[135_000, 7] = np.inf
    cases.append(("inf row", xn, q, None))
    qn = q.copy(); qn[3, 0] = np.nan; qn[4, 5] = np.inf
    cases.append(("nan query", x, qn, None))
    try:
        for name, xx, qq, lab in cases:
            out = {}
            for v in (2, 1):
                amd.set_tuning("flat_variant", v)
                ix = amd.FlatIndex(1, D); ix.add(xx, labels=lab)
                out[v] = ix.search(qq, k)
                if v == 2:
                    assert ix.last_search()[0] == (name == "labels"), name   # labels do not stop the filter, the others do
            assert np.array_equal(out[2][1], out[1][1]), name
            assert np.array_equal(bits(out[2][0]), bits(out[1][0])), name
    finally:
        amd.set_tuning("flat_variant", 0)


def test_flat_f32_filter_ties_arrive_out_of_order(amd):
    """hundreds of rows at exactly the same distance, reaching the final selection in no particular order: the k best are
    still the lowest row numbers (the selection must offer candidates that tie the current k-th key)"""
    rng = np.random.default_rng(3)
    n, D, nq, k = 131072, 48, 300, 100
    x = rng.normal(size=(n, D)).astype(np.float32)
    x = x[np.argsort(x[:, 0])]
    x[100_000:100_900] = x[70_000]                    # 900 duplicates outside the leading sample
    q = (x[rng.integers(0, n, nq)] + 0.1 * rng.normal(size=(nq, D))).astype(np.float32)
    q[:40] = x[70_000] + (0.01 * rng.normal(size=(40, D))).astype(np.float32)
    out = {}
    try:
        for v in (2, 1):
            amd.set_tuning("flat_variant", v)
            ix = amd.FlatIndex(1, D); ix.add(x)
            out[v] = ix.search(q, k)
    finally:
        amd.set_tuning("flat_variant", 0)
    assert np.array_equal(out[2][1], out[1][1]) and np.array_equal(bits(out[2][0]), bits(out[1][0]))


@pytest.mark.parametrize("D,nq", [(512, 500), (128, 700), (256, 130)])
def test_flat_u8_filter_wide_kernel_variant(amd, orc, D, nq):
    """flat_u8_gfilter 4 (one wave per SIMD, 96 / 128 queries per wave, ring of 4 tiles) against the default 8-wave filter kernel
    and the checker: query counts that leave the last query block partly empty, duplicates, odd and even tile counts per split"""
    rng = np.random.default_rng(D + nq)
    n, k = 262_144 + 32 * 7 + 5, 10
    x = rng.integers(0, 256, size=(n, D), dtype=np.uint8)
    x[200_000:200_050] = x[9]; x[n - 1] = x[9]
    q = x[rng.integers(0, n, nq)].copy()
    q[:, :3] ^= 1
    q[0] = x[9]
    out = {}
    try:
        amd.set_tuning("flat_variant", 2)
        for gf in (1, 4):
            amd.set_tuning("flat_u8_gfilter", gf)
            ix = amd.FlatIndex(L2U8, D); ix.add(x)
            out[gf] = ix.search(q, k)
            assert ix.last_search()[0]
            ix.close()
    finally:
        amd.set_tuning("flat_variant", 0); amd.set_tuning("flat_u8_gfilter", 1)
    assert np.array_equal(out[4][1], out[1][1]) and np.array_equal(out[4][0], out[1][0])
    od, odi, oi = orc.flat_search(L2U8, x, q[:12], k)
    assert np.array_equal(out[4][1][:12], oi) and np.array_equal(out[4][0][:12], odi)


@pytest.mark.parametrize("D,k,hi", [(512, 10, 256), (64, 64, 6), (96, 1, 256), (256, 33, 40)])
def test_flat_u8_filter_pipeline(amd, orc, D, k, hi):
    """uint8 L2 through sample + i8 matrix-core filter + sort (flat_variant 2) against the row-tile kernels (flat_variant 1) and
    the checker: few distinct byte values (masses of equal distances), duplicates of rows and of queries, appends"""
    rng = np.random.default_rng(D + k)
    n, nq = 270_000, 300
    x = rng.integers(0, hi, size=(n, D), dtype=np.uint8)
    x[200_000:200_400] = x[9]; x[100:140] = x[9]
    q = x[rng.integers(0, n, nq)].copy()
    flip = rng.integers(0, D, size=(nq, 3))
    for i in range(nq):
        q[i, flip[i]] ^= 1
    q[0] = x[9]
    out = {}
    try:
        for v in (2, 1):
            amd.set_tuning("flat_variant", v)
            ix = amd.FlatIndex(L2U8, D); ix.add(x[:150_000]); ix.add(x[150_000:])
            out[v] = ix.search(q, k)
            if v == 2:
                used, worst = ix.last_search()
                assert used, worst
    finally:
        amd.set_tuning("flat_variant", 0)
    assert np.array_equal(out[2][1], out[1][1]) and np.array_equal(out[2][0], out[1][0])
    od, odi, oi = orc.flat_search(L2U8, x, q[:16], k)
    assert np.array_equal(out[2][1][:16], oi) and np.array_equal(out[2][0][:16], odi)


@pytest.mark.parametrize("D,nq,k,hi", [(512, 1, 10, 256), (128, 3, 128, 4), (256, 4, 1, 256), (512, 2, 33, 256), (512, 8, 10, 256), (128, 16, 5, 256),
                                        (256, 11, 20, 4), (512, 7, 128, 256)])
def test_flat_u8_tiny_batch_stream(amd, orc, D, nq, k, hi):
    """1..16 uint8 queries through the streaming matrix-core kernel + minima-based selection (flat_variant 0) -- against the
    row-per-lane / row-tile kernels (flat_variant 1) and the checker; ragged row count (last tile partly empty), duplicate rows at both
    ends of the table (ties resolved by row), few distinct byte values (masses of equal distances), a far query (distances > 2^24, not exact in f32)"""
    rng = np.random.default_rng(D * 7 + k)
    n = 262_144 + 12_345
    x = rng.integers(0, hi, size=(n, D), dtype=np.uint8)
    x[n - 1] = x[3]; x[131_072] = x[3]
    q = x[rng.integers(0, n, nq)].copy()
    q[0] = x[3]
    if nq > 1:
        q[-1] = np.where(x[7] < 128, 255, 0)                    # far query: distances near 512 * 200^2 > 2^24
    out = {}
    try:
        for v in (0, 1):
            amd.set_tuning("flat_variant", v)
            ix = amd.FlatIndex(L2U8, D); ix.add(x[:100_000]); ix.add(x[100_000:])
            out[v] = ix.search(q, k)
            ix.close()
    finally:
        amd.set_tuning("flat_variant", 0)
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][0], out[1][0])
    od, odi, oi = orc.flat_search(L2U8, x, q, k)
    assert np.array_equal(out[0][1], oi) and np.array_equal(out[0][0], odi)
    assert out[0][1][0, :min(k, 3)].tolist() == [3, 131_072, n - 1][:min(k, 3)]


@pytest.mark.parametrize("n,D,nq,k", [(4096, 512, 1, 128), (4097, 128, 40, 128), (5000, 256, 300, 10), (40_000, 512, 130, 100), (200_000, 128, 64, 1)])
def test_flat_u8_stream_small_tables(amd, orc, n, D, nq, k):
    """Round 5: the uint8 stream takes tables from 4096 rows on (it used to start at 262 144; the row-tile kernels were 2-20x behind on
    everything smaller).  4096 rows is the structural bound -- k <= 128 waves with one 32-row tile each --: most waves hold no tile at
    all, a batch above 128 queries runs in passes.  Against the exact kernels (flat_variant 1) and the checker, duplicates included."""
    rng = np.random.default_rng(n + D + k)
    x = rng.integers(0, 256, size=(n, D), dtype=np.uint8)
    x[n - 1] = x[3]; x[n // 2] = x[3]
    q = x[rng.integers(0, n, nq)].copy()
    q[0] = x[3]
    out = {}
    try:
        for v in (0, 1):
            amd.set_tuning("flat_variant", v)
            ix = amd.FlatIndex(L2U8, D); ix.add(x[: n // 3]); ix.add(x[n // 3:])
            out[v] = ix.search(q, k)
            ix.close()
    finally:
        amd.set_tuning("flat_variant", 0)
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][0], out[1][0])
    od, odi, oi = orc.flat_search(L2U8, x, q, k)
    assert np.array_equal(out[0][1], oi) and np.array_equal(out[0][0], odi)


@pytest.mark.parametrize("metric,nq", [(IP, 16), (L2F, 17), (IP, 40), (L2F, 63)])
def test_flat_f32_filter_small_batches(amd, orc, metric, nq):
    """16..63 queries take the matrix-core filter by default (flat_variant 0): same answer as the exact kernels and the checker,
    distances bit for bit; duplicates of a row give (distance, row) ties"""
    rng = np.random.default_rng(nq)
    n, D, k = 300_000, 128, 10
    x = rng.standard_normal((n, D)).astype(np.float32)
    x[250_000:250_020] = x[11]
    q = (x[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, D))).astype(np.float32)
    q[0] = x[11]
    bits = lambda a: np.asarray(a, dtype=np.float32).view(np.uint32)
    ix = amd.FlatIndex(metric, D); ix.add(x)
    try:
        d0, i0 = ix.search(q, k)
        assert ix.last_search()[0]
        amd.set_tuning("flat_variant", 1)
        d1, i1 = ix.search(q, k)
        assert not ix.last_search()[0]
    finally:
        amd.set_tuning("flat_variant", 0)
    assert np.array_equal(i0, i1) and np.array_equal(bits(d0), bits(d1))
    od, _, oi = orc.flat_search(metric, x, q[:8], k, flavour=4 if metric == IP else 8)
    assert np.array_equal(i0[:8], oi) and np.array_equal(bits(d0[:8]), bits(od))
    ix.close()


@pytest.mark.parametrize("D,nq,k,hi", [(512, 9, 10, 256), (128, 17, 128, 4), (256, 33, 1, 256), (512, 64, 33, 256), (128, 100, 10, 256),
                                        (256, 128, 20, 6), (512, 40, 128, 256), (512, 300, 100, 256), (128, 129, 70, 4)])
def test_flat_u8_mid_batch_stream(amd, orc, D, nq, k, hi):
    """9..128 uint8 queries -- and larger batches the filter pipeline does not take (k > 64), in balanced passes of <= 128 --: matrix-core
    stream over the raw rows keeping tile / wave minima, then selection among the ~k tiles that qualify (flat_variant 0) -- against the row-tile kernels (flat_variant 1) and the checker; ragged row count (last tile partly
    empty), duplicate rows at both ends and in the middle (ties resolved by row across finish slices), few distinct byte values"""
    rng = np.random.default_rng(D * 11 + nq)
    n = 262_144 + 32 * 333 + 7
    x = rng.integers(0, hi, size=(n, D), dtype=np.uint8)
    x[n - 1] = x[3]; x[131_072] = x[3]; x[40_000:40_040] = x[3]
    q = x[rng.integers(0, n, nq)].copy()
    q[:, :2] ^= 1
    q[0] = x[3]
    q[-1] = np.where(x[7] < 128, 255, 0)                      # far query: distances > 2^24
    out = {}
    try:
        for v in (0, 1):
            amd.set_tuning("flat_variant", v)
            ix = amd.FlatIndex(L2U8, D); ix.add(x[:100_000]); ix.add(x[100_000:])
            out[v] = ix.search(q, k)
            ix.close()
    finally:
        amd.set_tuning("flat_variant", 0)
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][0], out[1][0])
    od, odi, oi = orc.flat_search(L2U8, x, q[:6], k)
    assert np.array_equal(out[0][1][:6], oi) and np.array_equal(out[0][0][:6], odi)
    assert out[0][1][0, 0] == 3


def _clustered(rng, n, D, metric):
    cen = rng.normal(size=(300, D)).astype(np.float32)
    x = (cen[rng.integers(0, 300, n)] + 0.5 * rng.normal(size=(n, D))).astype(np.float32)
    if metric == IP:
        x /= np.linalg.norm(x, axis=1, keepdims=True)
    else:
        x *= np.float32(11.0)
    return x


@pytest.mark.parametrize("metric,D,nq,k", [(IP, 128, 1, 100), (L2F, 128, 1, 100), (IP, 128, 7, 10), (L2F, 128, 33, 100), (IP, 128, 64, 100),
                                            (L2F, 128, 96, 128), (IP, 128, 97, 1), (L2F, 64, 128, 10), (IP, 32, 100, 50), (L2F, 96, 70, 20),
                                            (IP, 192, 40, 100), (L2F, 256, 33, 100), (IP, 256, 5, 3), (L2F, 128, 250, 100), (IP, 128, 384, 100),
                                            (L2F, 128, 500, 10), (IP, 256, 100, 10), (L2F, 32, 600, 128), (IP, 96, 129, 7)])
def test_flat_f32_stream(amd, orc, metric, D, nq, k):
    """fp32 search as one stream over the rows (flat_f32_stream 2: bf16 matrix-core scores, group best / second best, exact
    distances of the candidates) against the exact kernels (flat_variant 1) on the whole batch and against the checker on a few
    queries: ragged row count (last 64-row block partly empty), duplicates of rows inside one group and across groups, queries that
    are rows (zero distances, (distance, row) ties), appends between searches"""
    rng = np.random.default_rng(D * 13 + nq * 3 + k + metric)
    n = 70_000 + 37
    x = _clustered(rng, n, D, metric)
    x[60_000:60_150] = x[5]                 # 150 duplicates: more than k for small k, ties by row number
    x[32 * 1024 + 5] = x[5]                 # same wave, same lane, next tile of the group (the second best matters)
    x[69_999] = x[123]
    q = (x[rng.integers(0, n, nq)] + 0.05 * rng.normal(size=(nq, D))).astype(np.float32)
    q[0] = x[5]
    if nq > 3:
        q[3] = x[123]
    q = np.ascontiguousarray(q, np.float32)
    try:
        amd.set_tuning("flat_f32_stream", 2)
        ix = amd.FlatIndex(metric, D); ix.add(x[:50_001]); ix.add(x[50_001:])
        ds, is_ = ix.search(q, k)
        assert ix.last_search()[0] == 2
        ix.add(x[:777] * np.float32(0.5))   # append after a search
        ds2, is2 = ix.search(q, k)
        assert ix.last_search()[0] == 2
        amd.set_tuning("flat_variant", 1)
        de2, ie2 = ix.search(q, k)
        assert ix.last_search()[0] == 0
        ix.close()
        ix = amd.FlatIndex(metric, D); ix.add(x)
        de, ie = ix.search(q, k)
        ix.close()
    finally:
        amd.set_tuning("flat_variant", 0); amd.set_tuning("flat_f32_stream", 1)
    assert np.array_equal(is_, ie) and np.array_equal(bits(ds), bits(de))
    assert np.array_equal(is2, ie2) and np.array_equal(bits(ds2), bits(de2))
    m = min(nq, 6)
    od, _, oi = orc.flat_search(metric, x, q[:m], k, flavour=4 if metric == IP else 8)
    assert np.array_equal(is_[:m], oi) and np.array_equal(bits(ds[:m]), bits(od))


def test_flat_f32_stream_hands_hard_queries_to_the_exact_kernels(amd):
    """what the stream's bound does not cover is re-run by the exact kernels inside the same call, per query: non-finite queries,
    a query 2^70 times larger than the rows, masses of exact ties around the k-th place (lists run over); a non-finite ROW sends
    the whole index down the exact path.  Same results as flat_variant 1 everywhere, labels included"""
    rng = np.random.default_rng(5)
    n, D, nq, k = 66_000, 128, 40, 10
    x = rng.normal(size=(n, D)).astype(np.float32)
    x[2_000:7_000] = x[1]                                   # 5000 equal rows: every one ties at the k-th place of query 1
    q = rng.normal(size=(nq, D)).astype(np.float32)
    q[1] = x[1]
    q[3, 0] = np.nan; q[4, 5] = np.inf; q[6] *= np.float32(2.0 ** 70)
    labels = np.arange(n, dtype=np.int64) * 3 + 5
    xn = x.copy(); xn[65_000, 7] = np.inf
    try:
        for name, xx, lab in (("ties", x, None), ("labels", x, labels), ("inf row", xn, None)):
            out = {}
            for v in (0, 1):
                amd.set_tuning("flat_variant", v); amd.set_tuning("flat_f32_stream", 2 if v == 0 else 0)
                ix = amd.FlatIndex(L2F, D); ix.add(xx, labels=lab)
                out[v] = ix.search(q, k)
                if v == 0:
                    assert ix.last_search()[0] == (0 if name == "inf row" else 2), name
                ix.close()
            assert np.array_equal(out[0][1], out[1][1]), name
            assert np.array_equal(bits(out[0][0]), bits(out[1][0])), name
    finally:
        amd.set_tuning("flat_variant", 0); amd.set_tuning("flat_f32_stream", 1)


@pytest.mark.parametrize("share", [1, 2])
def test_flat_f32_stream_shared_ring_variants(amd, share):
    """both forms of the shared-ring kernel (four waves x 32 QB queries, eight waves x 32 queries) against the exact kernels,
    batch sizes around the pass boundaries, ragged last tile"""
    rng = np.random.default_rng(17 + share)
    n, D, k = 40_000 + 21, 128, 50
    x = _clustered(rng, n, D, L2F)
    x[30_000:30_060] = x[9]
    try:
        amd.set_tuning("flat_f32_share", share)
        ix = amd.FlatIndex(L2F, D); ix.add(x)
        for nq in (97, 256, 257, 385, 600):
            q = (x[rng.integers(0, n, nq)] + 0.05 * rng.normal(size=(nq, D))).astype(np.float32)
            q[0] = x[9]
            amd.set_tuning("flat_variant", 0)
            ds, is_ = ix.search(q, k)
            assert ix.last_search()[0] == 2
            amd.set_tuning("flat_variant", 1)
            de, ie = ix.search(q, k)
            assert np.array_equal(is_, ie) and np.array_equal(bits(ds), bits(de)), nq
        ix.close()
    finally:
        amd.set_tuning("flat_variant", 0); amd.set_tuning("flat_f32_share", 0)


@pytest.mark.parametrize("metric,D,k,prods", [(L2F, 128, 100, 2), (IP, 128, 100, 2), (L2F, 64, 10, 2), (IP, 64, 128, 2), (L2F, 128, 1, 3), (IP, 128, 37, 3),
                                               (L2F, 128, 100, 1), (IP, 64, 100, 1), (L2F, 64, 64, 3)])
def test_flat_f32_threshold_filter(amd, orc, metric, D, k, prods):
    """large fp32 batches as a threshold filter (round 6, flat_f32_tfilter.hip: sample maxima -> per-query threshold -> queries in
    LDS, the rows' bf16 operand copy in registers, records of 16 scores per hit -> per-query lists -> radix select, exact distances)
    with 1 / 2 / 3 bf16 products, against the exact kernels (flat_variant 1) on every query and the checker on a few: ragged row
    count, batch sizes around the query-block, workgroup-chunk and pass boundaries, duplicated rows (ties by row number, more of them
    than k), queries that are rows, rows appended between searches (the operand copy is rebuilt)"""
    rng = np.random.default_rng(D * 7 + k * 3 + metric + prods)
    n = 262_144 + 8_000 + 37
    x = _clustered(rng, n, D, metric)
    x[200_000:200_150] = x[5]
    x[77] = x[5]
    x[n - 1] = x[123]
    try:
        amd.set_tuning("flat_f32_tfilter", prods); amd.set_tuning("flat_f32_tfilter_min", 16)
        ix = amd.FlatIndex(metric, D); ix.add(x[:n - 5_000])
        out = {}
        for nq in (16, 129, 160, 513, 1030):
            q = (x[rng.integers(0, n - 5_000, nq)] + 0.05 * rng.normal(size=(nq, D))).astype(np.float32)
            q[0] = x[5]; q[3] = x[123]
            q = np.ascontiguousarray(q, np.float32)
            ds, is_ = ix.search(q, k)
            assert ix.last_search()[0] == 3, nq
            out[nq] = (q, ds, is_)
        ix.add(x[n - 5_000:])                   # append after searches
        q160 = out[160][0]
        ds2, is2 = ix.search(q160, k)
        assert ix.last_search()[0] == 3
        amd.set_tuning("flat_variant", 1)
        de2, ie2 = ix.search(q160, k)
        assert ix.last_search()[0] == 0
        ix.close()
        ix = amd.FlatIndex(metric, D); ix.add(x[:n - 5_000])
        for nq, (q, ds, is_) in out.items():
            de, ie = ix.search(q, k)
            assert np.array_equal(is_, ie) and np.array_equal(bits(ds), bits(de)), nq
        ix.close()
    finally:
        amd.set_tuning("flat_variant", 0); amd.set_tuning("flat_f32_tfilter", 4); amd.set_tuning("flat_f32_tfilter_min", 0)
    assert np.array_equal(is2, ie2) and np.array_equal(bits(ds2), bits(de2))
    od, _, oi = orc.flat_search(metric, x, q160[:4], k, flavour=4 if metric == IP else 8)
    assert np.array_equal(is2[:4], oi) and np.array_equal(bits(ds2[:4]), bits(od))


@pytest.mark.parametrize("metric,D,nq,k", [(L2F, 32, 300, 100), (IP, 96, 100, 10), (L2F, 160, 200, 100), (IP, 192, 700, 50), (L2F, 256, 129, 128),
                                            (IP, 384, 100, 100), (L2F, 512, 520, 20), (IP, 512, 33, 100), (L2F, 768, 150, 10), (IP, 1024, 100, 100),
                                            (L2F, 1024, 1100, 5), (IP, 100, 200, 100), (L2F, 100, 600, 10), (L2F, 20, 64, 100), (IP, 200, 128, 50),
                                            (L2F, 300, 1000, 100), (IP, 900, 40, 100)])
def test_flat_f32_threshold_filter_widths(amd, orc, metric, D, nq, k):
    """the threshold filter at every width it takes (the kernels' 32 ... 1024-d, 768 / 1024-d with one wave per SIMD; widths in between --
    100-d, 200-d, 300-d, 900-d, 20-d -- on the next kernel over zero-padded operands), products as the dispatch picks them, against the exact kernels on every query and the checker on three: ragged row count, duplicates, queries that are rows"""
    rng = np.random.default_rng(D + nq + k + metric)
    n = 262_144 + 1_000 + 13
    x = _clustered(rng, n, D, metric)
    x[100_000:100_140] = x[5]
    x[n - 1] = x[123]
    q = (x[rng.integers(0, n, nq)] + 0.05 * rng.normal(size=(nq, D))).astype(np.float32)
    q[0] = x[5]; q[3] = x[123]
    q = np.ascontiguousarray(q, np.float32)
    try:
        ix = amd.FlatIndex(metric, D); ix.add(x)
        ds, is_ = ix.search(q, k)
        assert ix.last_search()[0] == 3
        amd.set_tuning("flat_variant", 1)
        de, ie = ix.search(q, k)
        assert ix.last_search()[0] == 0
        ix.close()
    finally:
        amd.set_tuning("flat_variant", 0)
    assert np.array_equal(is_, ie) and np.array_equal(bits(ds), bits(de))
    od, _, oi = orc.flat_search(metric, x, q[:3], k, flavour=4 if metric == IP else 8)
    assert np.array_equal(is_[:3], oi) and np.array_equal(bits(ds[:3]), bits(od))


@pytest.mark.parametrize("metric,D,nq,k", [(L2F, 2048, 70, 10), (IP, 2048, 33, 100), (IP, 1536, 100, 100), (L2F, 1200, 64, 20)])
def test_flat_f32_threshold_filter_two_k_halves(amd, orc, metric, D, nq, k):
    """1536 / 2048-d rows (and widths padded up to them): a row tile's K steps go through a wave's registers in two halves, one query
    block per workgroup.  A table of 40 000 rows takes the pipeline through "flat_f32_tfilter_min_rows"; against the exact kernels on
    every query and the checker on three"""
    rng = np.random.default_rng(D + nq + k + metric)
    n = 40_000 + 13
    x = _clustered(rng, n, D, metric)
    x[10_000:10_140] = x[5]
    x[n - 1] = x[123]
    q = (x[rng.integers(0, n, nq)] + 0.05 * rng.normal(size=(nq, D))).astype(np.float32)
    q[0] = x[5]; q[3] = x[123]
    q = np.ascontiguousarray(q, np.float32)
    try:
        amd.set_tuning("flat_f32_tfilter_min_rows", 32768)
        ix = amd.FlatIndex(metric, D); ix.add(x)
        ds, is_ = ix.search(q, k)
        assert ix.last_search()[0] == 3
        amd.set_tuning("flat_variant", 1)
        de, ie = ix.search(q, k)
        assert ix.last_search()[0] == 0
        ix.close()
    finally:
        amd.set_tuning("flat_variant", 0); amd.set_tuning("flat_f32_tfilter_min_rows", 262144)
    assert np.array_equal(is_, ie) and np.array_equal(bits(ds), bits(de))
    od, _, oi = orc.flat_search(metric, x, q[:3], k, flavour=4 if metric == IP else 8)
    assert np.array_equal(is_[:3], oi) and np.array_equal(bits(ds[:3]), bits(od))


@pytest.mark.parametrize("metric,D,k", [(L2F, 128, 100), (IP, 128, 10), (L2F, 64, 128), (IP, 256, 100), (L2F, 32, 1), (IP, 192, 50)])
def test_flat_f32_stream_over_operand_copy(amd, orc, metric, D, k):
    """small batches (1 ... 64 queries) on a table that keeps the threshold filter's bf16 operand copy stream that copy's first terms
    ("flat_f32_packed" 1, round 6: half the bytes, one product, margins from the query's own rounding residues) instead of the fp32 rows:
    same lists and bits as the fp32 stream and the exact kernels; one / two / three query blocks per wave, ragged last tile, duplicates,
    queries that are rows, a non-finite query and a huge one (handed to the exact kernels), rows appended between searches"""
    rng = np.random.default_rng(D + k + metric)
    n = 262_144 + 3_000 + 21
    x = _clustered(rng, n, D, metric)
    x[150_000:150_140] = x[5]
    x[n - 1] = x[123]
    try:
        ix = amd.FlatIndex(metric, D); ix.add(x[:n - 2_000])
        for nq in (1, 5, 33, 64):
            q = (x[rng.integers(0, n - 2_000, nq)] + 0.05 * rng.normal(size=(nq, D))).astype(np.float32)
            q[0] = x[5]
            if nq > 4:
                q[2, 0] = np.nan; q[4] *= np.float32(2.0 ** 70)
            q = np.ascontiguousarray(q, np.float32)
            amd.set_tuning("flat_variant", 1); de, ie = ix.search(q, k); amd.set_tuning("flat_variant", 0)
            for pk in (1, 0):
                amd.set_tuning("flat_f32_packed", pk)
                ds, is_ = ix.search(q, k)
                assert ix.last_search()[0] == 2, (nq, pk)
                assert np.array_equal(is_, ie) and np.array_equal(bits(ds), bits(de)), (nq, pk)
        amd.set_tuning("flat_f32_packed", 1)
        ix.add(x[n - 2_000:])
        q = np.ascontiguousarray(x[[5, 123, n - 1, 77]] + np.float32(0.01))
        ds, is_ = ix.search(q, k)
        od, _, oi = orc.flat_search(metric, x, q, k, flavour=4 if metric == IP else 8)
        assert np.array_equal(is_, oi) and np.array_equal(bits(ds), bits(od))
        ix.close()
    finally:
        amd.set_tuning("flat_variant", 0); amd.set_tuning("flat_f32_packed", 1)


@pytest.mark.parametrize("metric,D,nq,k", [(L2F, 128, 1, 129), (IP, 128, 70, 300), (L2F, 64, 300, 1000), (IP, 128, 20, 2048), (L2F, 256, 33, 500),
                                            (IP, 512, 10, 200)])
def test_flat_f32_threshold_filter_big_k(amd, orc, metric, D, nq, k):
    """k = 129 ... 2048 (round 6: 4096 sample maxima, candidate lists of 32 768, ft_finish_big_kernel: keys in LDS, radix select, bitonic
    sort of the exact distances) against the exact kernels ("flat_f32_tfilter_bigk" 0) on every query and the checker on two: duplicates
    (more of them than k for the smaller k), queries that are rows, ragged row count"""
    rng = np.random.default_rng(D + nq + k + metric)
    n = 262_144 + 4_000 + 7
    x = _clustered(rng, n, D, metric)
    x[100_000:100_400] = x[5]
    x[n - 1] = x[123]
    q = (x[rng.integers(0, n, nq)] + 0.05 * rng.normal(size=(nq, D))).astype(np.float32)
    q[0] = x[5]
    q = np.ascontiguousarray(q, np.float32)
    try:
        ix = amd.FlatIndex(metric, D); ix.add(x)
        ds, is_ = ix.search(q, k)
        assert ix.last_search()[0] == 3
        amd.set_tuning("flat_f32_tfilter_bigk", 0)
        de, ie = ix.search(q, k)
        assert ix.last_search()[0] == 0
        ix.close()
    finally:
        amd.set_tuning("flat_f32_tfilter_bigk", 1)
    assert np.array_equal(is_, ie) and np.array_equal(bits(ds), bits(de))
    od, _, oi = orc.flat_search(metric, x, q[:2], k, flavour=4 if metric == IP else 8)
    assert np.array_equal(is_[:2], oi) and np.array_equal(bits(ds[:2]), bits(od))


def test_flat_f32_threshold_filter_second_attempt(amd):
    """candidate lists that run over: with "flat_f32_tfilter_retry" 1 such a query takes a second filter pass under the threshold its
    stored candidates give ("flat_f32_dbg" 32 loosens the sample's thresholds so that lists do run over and second attempts succeed),
    without it the exact kernels answer; masses of ties make no progress and go to the exact kernels either way.  Same lists"""
    rng = np.random.default_rng(11)
    n, D, nq, k = 280_000, 128, 200, 20
    x = _clustered(rng, n, D, L2F)
    x[5_000:11_000] = x[1]
    q = (x[rng.integers(0, n, nq)] + 0.05 * rng.normal(size=(nq, D))).astype(np.float32)
    q[1] = x[1]
    try:
        ix = amd.FlatIndex(L2F, D); ix.add(x)
        amd.set_tuning("flat_variant", 1)
        de, ie = ix.search(q, k)
        amd.set_tuning("flat_variant", 0)
        for retry, dbg in ((0, 0), (1, 0), (1, 32), (0, 32)):
            amd.set_tuning("flat_f32_tfilter_retry", retry); amd.set_tuning("flat_f32_dbg", dbg)
            ds, is_ = ix.search(q, k)
            assert ix.last_search()[0] == 3
            assert np.array_equal(is_, ie) and np.array_equal(bits(ds), bits(de)), (retry, dbg)
        ix.close()
    finally:
        amd.set_tuning("flat_variant", 0); amd.set_tuning("flat_f32_tfilter_retry", 0); amd.set_tuning("flat_f32_dbg", 0)


def test_flat_f32_threshold_filter_hands_hard_queries_to_the_exact_kernels(amd):
    """what the filter's bound does not cover is re-run by the exact kernels inside the same call, per query: non-finite queries, a
    query 2^70 times larger than the rows, masses of exact ties around the k-th place (more rows at the threshold than a list
    holds), a zero query; a non-finite ROW sends the whole index to the other paths; a small batch forced through the pipeline
    ("flat_f32_tfilter_min").  Same results as flat_variant 1 everywhere, labels included"""
    rng = np.random.default_rng(6)
    n, D, nq, k = 270_000, 128, 150, 10
    x = rng.normal(size=(n, D)).astype(np.float32)
    x[2_000:8_000] = x[1]                                   # 6000 equal rows: every one ties at the k-th place of query 1
    q = rng.normal(size=(nq, D)).astype(np.float32)
    q[1] = x[1]
    q[3, 0] = np.nan; q[4, 5] = np.inf; q[6] *= np.float32(2.0 ** 70); q[7] = 0
    labels = np.arange(n, dtype=np.int64) * 3 + 5
    xn = x.copy(); xn[265_000, 7] = np.inf
    try:
        for name, xx, lab, qq in (("ties", x, None, q), ("labels", x, labels, q), ("inf row", xn, None, q), ("five queries", x, None, q[:5])):
            out = {}
            for v in (0, 1):
                amd.set_tuning("flat_variant", v); amd.set_tuning("flat_f32_tfilter_min", 1 if name == "five queries" else 0)
                ix = amd.FlatIndex(L2F, D); ix.add(xx, labels=lab)
                out[v] = ix.search(qq, k)
                if v == 0:
                    assert ix.last_search()[0] == (3 if name != "inf row" else 0), name
                ix.close()
            assert np.array_equal(out[0][1], out[1][1]), name
            assert np.array_equal(bits(out[0][0]), bits(out[1][0])), name
    finally:
        amd.set_tuning("flat_variant", 0); amd.set_tuning("flat_f32_tfilter_min", 0)


@pytest.mark.parametrize("d,n", [(512, 20_000), (256, 17_001), (128, 40_000), (100, 16_385), (516, 16_400), (4, 70_000)])
def test_sq8_decode_through_table(amd, orc, d, n):
    """>= 16384 rows decode through the per-column tables in LDS (sq8_decode_lut_kernel): every byte value of every column,
    slabs that end inside a row, hard vdiff values (zero, all-ones significand, tiny, huge) -- bit for bit with the checker's
    double arithmetic; and the no-write-back form of the normalising encode gives the same codes and leaves x alone"""
    rng = np.random.default_rng(d + n)
    vmin = (rng.normal(size=d) * np.exp2(rng.integers(-20, 20, size=d))).astype(np.float32)
    vdiff = np.abs(rng.normal(size=d) * np.exp2(rng.integers(-20, 20, size=d))).astype(np.float32)
    vdiff[0] = 0.0
    vdiff[1 % d] = np.frombuffer(np.uint32(0x3d7fffff).tobytes(), np.float32)[0]
    vdiff[2 % d] = np.float32(1e-41)
    vdiff[3 % d] = np.float32(1e30)
    codes = rng.integers(0, 256, size=(n, d), dtype=np.uint8)
    codes[:256] = np.arange(256, dtype=np.uint8)[:, None]          # every byte value in every column
    dec = amd.sq8_decode(vmin, vdiff, codes)
    assert np.array_equal(bits(dec), bits(orc.sq8_decode(vmin, vdiff, codes)))
    # the faiss-path arithmetic (Int8Decode(uint8_t*), int8_quan.cc:96-104): fp32 codec, through the same table kernel and, on a few
    # rows, the element-wise kernels; it is a different function of the same bytes
    decf = amd.sq8_decode_faiss(vmin, vdiff, codes)
    wantf = orc.sq8_decode_faiss(vmin, vdiff, codes)
    assert np.array_equal(bits(decf), bits(wantf))
    assert np.array_equal(bits(amd.sq8_decode_faiss(vmin, vdiff, codes[:300])), bits(wantf[:300]))
    assert not np.array_equal(bits(decf), bits(dec))
    if d in (512, 256, 100):
        x = np.abs(rng.normal(size=(5000, d))).astype(np.float32)
        tv, td = amd.sq8_train(x, l2norm=True)
        x1, x2 = x.copy(), x.copy()
        c1 = amd.sq8_encode(tv, td, x1, l2norm=True)
        c2 = amd.sq8_encode(tv, td, x2, l2norm=2)
        assert np.array_equal(c1, c2) and np.array_equal(bits(x2), bits(x)) and not np.array_equal(bits(x1), bits(x))


@pytest.mark.parametrize("D,nq,k,hi", [(512, 300, 129, 256), (512, 7, 2048, 256), (256, 260, 500, 256), (128, 40, 1000, 256), (128, 300, 200, 6),
                                       (512, 1000, 10, 256), (384, 129, 64, 256), (192, 257, 100, 256), (96, 300, 1, 256), (64, 513, 128, 256), (128, 130, 33, 6),
                                       (96, 5, 10, 256), (64, 2, 100, 256), (384, 33, 128, 256), (128, 1100, 10, 256), (256, 2100, 600, 256),
                                       (32, 300, 10, 256), (160, 7, 100, 256), (224, 130, 129, 256), (288, 64, 10, 256), (320, 1000, 64, 256), (352, 2, 1, 256),
                                       (416, 200, 500, 256), (448, 31, 10, 256), (480, 257, 100, 256)])
def test_flat_u8_threshold_filter(amd, orc, D, nq, k, hi):
    """uint8 L2 batches and every batch with k = 129 .. 2048 (round 6, flat_u8_tfilter.hip: exact integer scores on the i8 matrix cores
    over the operand copy, 4096 sample maxima -> the threshold itself, candidate lists, radix select + sort) against the round-5 paths
    ("flat_u8_tfilter" 0: stream passes / sample + filter pipeline / exact kernels) on every query and the checker on two; every width
    with a kernel, query counts around the 256-query passes and their 32-query blocks; duplicates (more of them than k for the smaller k), queries that are rows, a ragged row
    count, appends; few distinct byte values (hi = 6: masses of equal distances -- the call may fall back as a whole, same lists)"""
    rng = np.random.default_rng(D + nq + k + hi)
    n = 262_144 + 4_000 + 7
    x = rng.integers(0, hi, size=(n, D), dtype=np.uint8)
    centres = rng.integers(0, hi, size=(64, D), dtype=np.uint8)
    near = rng.integers(0, n, 40_000)
    x[near] = centres[rng.integers(0, 64, near.size)]
    x[near, rng.integers(0, D, near.size)] ^= 1            # clusters: thousands of rows a step or two from each centre
    x[100_000:100_400] = x[5]
    x[n - 1] = x[123]
    q = x[rng.integers(0, n, nq)].copy()
    q[:, :3] ^= 1
    q[0] = x[5]
    q[1] = centres[3]
    try:
        ix = amd.FlatIndex(L2U8, D); ix.add(x[:100_000]); ix.add(x[100_000:n - 3_001])
        ix.search(q, k)                      # (the operand copy exists from here on: the rows added next are packed behind it, from a partly filled tile on)
        ix.add(x[n - 3_001:])
        ds, is_ = ix.search(q, k)
        how = ix.last_search()[0]
        assert how == 4 or hi < 256
        amd.set_tuning("flat_u8_tfilter", 0)
        de, ie = ix.search(q, k)
        assert ix.last_search()[0] != 4
        ix.close()
    finally:
        amd.set_tuning("flat_u8_tfilter", 1)
    assert np.array_equal(is_, ie) and np.array_equal(ds, de)
    _, odi, oi = orc.flat_search(L2U8, x, q[:2], k)
    assert np.array_equal(is_[:2], oi) and np.array_equal(ds[:2], odi)


def test_flat_u8_threshold_filter_hands_hard_queries_to_the_other_kernels(amd, orc):
    """queries the uint8 threshold filter cannot answer are re-run inside the call, by the row-per-lane kernels under the flags as a predicate (nothing
    waits for the device): 6000 equal rows tie at the k-th place of the
    query that equals them (the finish keeps 4096), and 40 000 copies of another row overflow that query's candidate list; the rest of the
    batch stays with the filter.  Same lists as "flat_u8_tfilter" 0, labels included"""
    rng = np.random.default_rng(12)
    n, D, nq, k = 300_000, 128, 200, 10
    x = rng.integers(0, 256, size=(n, D), dtype=np.uint8)
    x[2_000:8_000] = x[1]
    x[50_000:90_000] = x[3]
    q = x[rng.integers(100_000, n, nq)].copy()
    q[:, :3] ^= 1
    q[1] = x[1]; q[7] = x[3]; q[7, 0] ^= 1
    labels = rng.permutation(n).astype(np.int64)
    try:
        ix = amd.FlatIndex(L2U8, D); ix.add(x, labels)
        ds, is_ = ix.search(q, k)
        how, _ = ix.last_search()
        assert how == 4
        amd.set_tuning("flat_u8_tfilter", 0)
        de, ie = ix.search(q, k)
        assert ix.last_search()[0] != 4
        ix.close()
    finally:
        amd.set_tuning("flat_u8_tfilter", 1)
    assert np.array_equal(is_, ie) and np.array_equal(ds, de)
    _, odi, oi = orc.flat_search(L2U8, x, q[[1, 7, 20]], k)
    assert np.array_equal(is_[[1, 7, 20]], labels[oi]) and np.array_equal(ds[[1, 7, 20]], odi)


@pytest.mark.parametrize("D,n,nq,k", [(512, 65_536 + 5, 300, 129), (128, 70_000, 50, 1000), (256, 200_003, 1000, 2048), (512, 131_072, 3, 2048),
                                      (512, 65_536 + 5, 300, 10), (64, 70_000, 8, 100), (128, 100_003, 1000, 128)])
def test_flat_u8_threshold_filter_small_tables(amd, orc, D, n, nq, k):
    """tables from 65 536 rows (the sample fills two slots per wave that gets a tile group; every group is in it): k > 128 at every batch, k <= 128
    from 129 queries, widths without a streaming kernel from two -- against the round-5 kernels on every query and the checker on two"""
    rng = np.random.default_rng(D + n + k)
    x = rng.integers(0, 256, size=(n, D), dtype=np.uint8)
    x[1_000:1_300] = x[5]
    q = x[rng.integers(0, n, nq)].copy()
    q[:, :3] ^= 1
    q[0] = x[5]
    try:
        ix = amd.FlatIndex(L2U8, D); ix.add(x)
        ds, is_ = ix.search(q, k)
        assert ix.last_search()[0] == 4
        amd.set_tuning("flat_u8_tfilter", 0)
        de, ie = ix.search(q, k)
        assert ix.last_search()[0] != 4
        ix.close()
    finally:
        amd.set_tuning("flat_u8_tfilter", 1)
    assert np.array_equal(is_, ie) and np.array_equal(ds, de)
    _, odi, oi = orc.flat_search(L2U8, x, q[:2], k)
    assert np.array_equal(is_[:2], oi) and np.array_equal(ds[:2], odi)


@pytest.mark.parametrize("metric,D,n,nq,k", [(IP, 128, 65_536, 200, 129), (L2F, 96, 70_001, 50, 1000), (L2F, 512, 100_000, 300, 2048), (IP, 1024, 80_000, 40, 200)])
def test_flat_f32_threshold_filter_big_k_small_tables(amd, orc, metric, D, n, nq, k):
    """k > 128 on tables from 65 536 rows and 48 k (below the pipeline's 262 144): against the exact kernels on every query, the checker on two"""
    rng = np.random.default_rng(D + n + k)
    x = _clustered(rng, n, D, metric)
    x[10_000:10_300] = x[5]
    q = (x[rng.integers(0, n, nq)] + 0.05 * rng.normal(size=(nq, D))).astype(np.float32)
    q[0] = x[5]
    q = np.ascontiguousarray(q, np.float32)
    try:
        ix = amd.FlatIndex(metric, D); ix.add(x)
        ds, is_ = ix.search(q, k)
        assert ix.last_search()[0] == 3
        amd.set_tuning("flat_f32_tfilter_bigk", 0)
        de, ie = ix.search(q, k)
        assert ix.last_search()[0] == 0
        ix.close()
    finally:
        amd.set_tuning("flat_f32_tfilter_bigk", 1)
    assert np.array_equal(is_, ie) and np.array_equal(bits(ds), bits(de))
    od, _, oi = orc.flat_search(metric, x, q[:2], k, flavour=4 if metric == IP else 8)
    assert np.array_equal(is_[:2], oi) and np.array_equal(bits(ds[:2]), bits(od))


@pytest.mark.parametrize("metric,D,n,nq,k", [(IP, 512, 65_536, 200, 10), (L2F, 100, 70_001, 16, 100), (L2F, 1024, 100_000, 130, 100), (IP, 1024, 80_000, 20, 32),
                                             (L2F, 2048, 66_000, 64, 5)])
def test_flat_f32_threshold_filter_small_tables(amd, orc, metric, D, n, nq, k):
    """k <= 128 at the widths without a stream kernel on tables from 65 536 rows (below the pipeline's 262 144): against the exact kernels on
    every query, the checker on two"""
    rng = np.random.default_rng(D + n + k)
    x = _clustered(rng, n, D, metric)
    x[10_000:10_300] = x[5]
    q = (x[rng.integers(0, n, nq)] + 0.05 * rng.normal(size=(nq, D))).astype(np.float32)
    q[0] = x[5]
    q = np.ascontiguousarray(q, np.float32)
    try:
        ix = amd.FlatIndex(metric, D); ix.add(x)
        ds, is_ = ix.search(q, k)
        assert ix.last_search()[0] == 3
        amd.set_tuning("flat_variant", 1)
        de, ie = ix.search(q, k)
        assert ix.last_search()[0] == 0
        ix.close()
    finally:
        amd.set_tuning("flat_variant", 0)
    assert np.array_equal(is_, ie) and np.array_equal(bits(ds), bits(de))
    od, _, oi = orc.flat_search(metric, x, q[:2], k, flavour=4 if metric == IP else (8 if D % 16 == 0 else 4))
    assert np.array_equal(is_[:2], oi) and np.array_equal(bits(ds[:2]), bits(od))


@pytest.mark.parametrize("metric,D", [(IP, 128), (L2F, 100), (L2F, 1024)])
def test_flat_f32_rows_copy(amd, metric, D):
    """the threshold filter's exact finish out of the row-major copy of the rows ("flat_f32_rows_copy", round 6: the blocked layout gathers 16 of every
    128 bytes it fetches) -- built at the first such search, extended behind itself after appends that end inside a 64-row block; same lists and
    bits as without the copy and as the exact kernels"""
    rng = np.random.default_rng(D)
    n, nq, k = 262_144 + 4_000 + 7, 150, 20
    x = _clustered(rng, n, D, metric)
    x[100_000:100_200] = x[5]
    q = (x[rng.integers(0, n, nq)] + 0.05 * rng.normal(size=(nq, D))).astype(np.float32)
    q[0] = x[5]
    q = np.ascontiguousarray(q, np.float32)
    out = {}
    try:
        for copy in (4, 0):
            amd.set_tuning("flat_f32_rows_copy", copy)
            ix = amd.FlatIndex(metric, D); ix.add(x[:262_144 + 33])
            ix.search(q, k)                                   # the copy covers the first rows from here on
            ix.add(x[262_144 + 33:n - 1_001]); ix.search(q[:70], k)
            ix.add(x[n - 1_001:])
            out[copy] = ix.search(q, k)
            assert ix.last_search()[0] == 3
            if copy == 0:
                amd.set_tuning("flat_variant", 1)
                out["exact"] = ix.search(q, k)
                amd.set_tuning("flat_variant", 0)
            ix.close()
    finally:
        amd.set_tuning("flat_f32_rows_copy", 4); amd.set_tuning("flat_variant", 0)
    for key in (0, "exact"):
        assert np.array_equal(out[4][1], out[key][1]) and np.array_equal(bits(out[4][0]), bits(out[key][0])), key


def test_flat_threshold_filters_random_shapes(amd):
    """tools/flat_fuzz_ab.py: random rows / width / batch / k / duplicates / appends around the threshold filters' dispatch bounds, uint8 and fp32 -- the
    default routes against the round-5 kernels, lists and distance bits (1020 cases of it ran clean at the end of round 6)"""
    import importlib.util, pathlib
    spec = importlib.util.spec_from_file_location("flat_fuzz_ab", pathlib.Path(__file__).resolve().parent.parent / "tools" / "flat_fuzz_ab.py")
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    bad, paths = mod.run(40, 11, verbose=False)
    assert bad == 0 and paths.get(3, 0) > 5 and paths.get(4, 0) > 5, (bad, paths)


@pytest.mark.parametrize("metric,D", [(L2U8, 128), (IP, 128), (L2F, 512)])
def test_flat_threshold_filters_concurrent_searches_on_one_handle(amd, metric, D):
    """eight host threads search ONE handle at once through the threshold filters (host-pointer entry: every call leases its own scratch set and
    stream; the operand copies are built once under the exclusive lock), batches of different sizes and k, while a ninth appends rows in between
    two rounds: every answer equals the serial one on the same rows"""
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.default_rng(D + metric)
    n = 270_000
    if metric == L2U8:
        x = rng.integers(0, 256, size=(n, D), dtype=np.uint8)
        qs = [x[rng.integers(0, n, nq)].copy() for nq in (130, 257, 300, 1, 513, 140, 200, 1000)]
        for q in qs: q[:, :3] ^= 1
    else:
        x = _clustered(rng, n, D, metric)
        qs = [np.ascontiguousarray(x[rng.integers(0, n, nq)] + 0.05 * rng.normal(size=(nq, D)), np.float32) for nq in (130, 257, 300, 100, 513, 140, 200, 1000)]
    ks = (10, 100, 129, 300, 1, 64, 128, 20)
    ix = amd.FlatIndex(metric, D); ix.add(x[:265_000])
    same = lambda a, b: np.array_equal(a[1], b[1]) and np.array_equal(bits(a[0]) if metric != L2U8 else a[0], bits(b[0]) if metric != L2U8 else b[0])
    for rnd in range(2):
        with ThreadPoolExecutor(max_workers=8) as ex:
            got = list(ex.map(lambda j: ix.search(qs[j], ks[j]), range(8)))
        want = [ix.search(qs[j], ks[j]) for j in range(8)]
        assert all(same(g, w) for g, w in zip(got, want)), rnd
        if rnd == 0: ix.add(x[265_000:])
    ix.close()


def test_flat_u8_threshold_filter_record_regions_run_over(amd):
    """CVTMI_UT_DBG 4 shrinks the waves' record regions to two records: every pass raises its flag, its finish flags every query, and the row-per-lane
    kernels answer under the predicate -- same lists as without the hook and as "flat_u8_tfilter" 0 (two passes: 1100 queries)"""
    import os
    rng = np.random.default_rng(4)
    n, D, nq, k = 270_000, 128, 1100, 10
    x = rng.integers(0, 256, size=(n, D), dtype=np.uint8)
    q = x[rng.integers(0, n, nq)].copy(); q[:, :3] ^= 1
    ix = amd.FlatIndex(L2U8, D); ix.add(x)
    want = ix.search(q, k)
    assert ix.last_search()[0] == 4
    try:
        os.environ["CVTMI_UT_DBG"] = "4"
        got = ix.search(q, k)
        assert ix.last_search()[0] == 4
    finally:
        del os.environ["CVTMI_UT_DBG"]
    again = ix.search(q, k)
    ix.close()
    for g in (got, again):
        assert np.array_equal(g[1], want[1]) and np.array_equal(g[0], want[0])
