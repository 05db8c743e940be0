"""CPU: the oracle (oracle/cvt_oracle.c) against the golden vectors produced by the reference itself
(tests/golden/make_golden.py), and -- when oracle/_ref is present -- against the reference live."""
import numpy as np
import pytest

from conftest import OPQ_CASES, SQ8_NORM_GROUPS, bits


@pytest.mark.parametrize("case", OPQ_CASES)
def test_opq_pipeline_matches_reference(orc, golden, case):
    g = golden.opq[case]
    # rotation = IVFOPQ::reorder
    assert np.array_equal(orc.reorder(g["perm"], g["queries"]), g["q_rot"])
    assert np.array_equal(orc.reorder(g["perm"], g["db"]), g["db_rot"])
    # encode = IVFOPQ::Add; the reference stores entries per list in insertion order
    lists, codes = orc.pq_encode(g["db_rot"], g["coarse"], g["books"])
    order = np.argsort(lists, kind="stable")
    assert np.array_equal(codes[order], g["codes"])
    assert np.array_equal(golden.video_of_row(case)[order], g["video_id"])
    off = np.zeros(g["coarse"].shape[0] + 1, dtype=np.int64)
    np.cumsum(np.bincount(lists, minlength=g["coarse"].shape[0]), out=off[1:])
    assert np.array_equal(off, g["list_off"])
    # query = IVFOPQ::QueryThrehold
    ms = orc.query_video(g["q_rot"], g["coarse"], g["books"], int(g["nk"]), g["list_off"], g["codes"], g["video_id"],
                         len(g["video_rows"]))
    assert np.array_equal(bits(ms), bits(g["match_score"]))
    total, rd, ri = orc.video_rank(g["match_score"], len(g["rank_d"]))
    assert np.array_equal(bits(total), bits(g["total"]))
    assert np.array_equal(bits(rd), bits(g["rank_d"])) and np.array_equal(ri, g["rank_i"])


def test_exhaustive_topk_form_equals_query_video(orc, golden):
    """One vector per video + no clamp: Query()'s match scores ARE the per-vector ADC distances, so the
    north-star form (LUT + scan + k smallest) must give the same numbers as the reference run."""
    g = golden.opq["opq_vec_m16"]
    d, i = orc.adc_search(g["q_rot"], g["books"], g["codes"], 100, centroid=g["coarse"][0])
    ms = g["match_score"]
    for f in range(ms.shape[0]):
        order = np.lexsort((np.arange(ms.shape[1]), ms[f]))[:100]  # (score, id) ascending
        assert np.array_equal(i[f], order)
        assert np.array_equal(bits(d[f]), bits(ms[f][order]))
    # duplicates of db row 7 tie on distance and must come back in id order
    assert list(i[0][:3]) == [7, 100, 200]


def test_lut_pinned_by_single_subquantiser(orc, golden):
    """M = 1: every reference score is 0.0f + LUT[0][code] -> pins the LUT arithmetic bit for bit."""
    g = golden.opq["opq_m1"]
    for f in range(g["q_rot"].shape[0]):
        lut = orc.lut(g["q_rot"][f], g["coarse"][0], g["books"])
        assert np.array_equal(bits(lut[0][g["codes"][:, 0]]), bits(g["match_score"][f]))


def test_known_answer_real_features(golden):
    """opq/data query 6231519245_6 is row 7 of db video 6231519245 -> that video ranks first."""
    for case in ("opq_real_q1", "opq_real_q9"):
        assert golden.opq[case]["rank_i"][0] == 0
    q = golden.opq["opq_real_q1"]["queries"]
    db = golden.opq["opq_real_q1"]["db"]
    assert np.array_equal(q[0], db[7])


def test_flat_matches_reference(orc, golden):
    from oracle import binding as ob
    f = golden.flat
    d, _, i = orc.flat_search(ob.IP, f["ip_db"], f["ip_q"], 100)
    assert np.array_equal(bits(d), bits(f["ip_d"])) and np.array_equal(i, f["ip_i"])
    d, _, i = orc.flat_search(ob.IP, f["ip_db"], f["ip_q"], 10, labels=f["ipl_labels"])
    assert np.array_equal(bits(d), bits(f["ipl_d"])) and np.array_equal(i, f["ipl_i"])
    d, _, i = orc.flat_search(ob.L2F, f["l2_db"], f["l2_q"], 10)
    assert np.array_equal(bits(d), bits(f["l2_d"])) and np.array_equal(i, f["l2_i"])
    for tag in ("u8a", "u8b", "u8c"):
        _, di, i = orc.flat_search(ob.L2U8, f[tag + "_db"], f[tag + "_q"], 10)
        assert np.array_equal(di, f[tag + "_d"]) and np.array_equal(i, f[tag + "_i"])
    # self queries come back first with distance 0 (uint8) / ties resolved by label
    assert f["u8a_d"][0][0] == 0 and f["u8c_i"][0][0] == 5


def test_rotate_fma_equals_permutation(orc):
    rng = np.random.default_rng(3)
    D = 64
    perm = rng.permutation(D).astype(np.int32)
    R = np.zeros((D, D), dtype=np.float32)
    R[np.arange(D), perm] = 1.0
    x = rng.normal(size=(37, D)).astype(np.float32)
    assert np.array_equal(orc.rotate_fma(R, x), orc.reorder(perm, x))


@pytest.mark.parametrize("group", SQ8_NORM_GROUPS)
def test_sq8_normalisation_pinned_by_reference(orc, golden, group):
    """a-Q / a-T: the L2 normalisation in front of Int8Encode and SQ training (int8_quan.cc:46-56 == utils/math_util.h:29-39)
    IS pinned: the golden rows are outputs of the reference's own MathUtil::L2NormArray compiled in place
    (oracle/_ref/libref_math.so), incl. zero rows, norms under the 1e-12 clamp, denormals, squares that overflow fp32, NaN / inf."""
    x, want = golden.sq8_norm[group + "_x"], golden.sq8_norm[group + "_array"]
    with np.errstate(all="ignore"):
        got = np.stack([orc.sq8_l2norm(r) for r in x])
    assert np.array_equal(bits(got), bits(want))
    # the encode entry normalises through the same function, row by row, and writes the rows back
    d = x.shape[1]
    _, xn = orc.sq8_encode(np.zeros(d, np.float32), np.ones(d, np.float32), x, l2norm=True)
    assert np.array_equal(bits(xn), bits(want))
    # train = per-dimension min / max - min of exactly these normalised rows (faiss RS_minmax; order-free, so nothing to round)
    finite = np.all(np.isfinite(want), axis=1)
    if finite.any():
        vmin, vdiff = orc.sq8_train(x[finite], l2norm=True)
        w = want[finite]
        assert np.array_equal(bits(vmin), bits(w.min(axis=0))) and np.array_equal(bits(vdiff), bits(w.max(axis=0) - w.min(axis=0)))


def test_sq8_normalisation_against_live_reference(orc, golden):
    """the same, re-run against oracle/_ref/libref_math.so when it is present (build container), on fresh rows; and the
    std::vector twin L2NormVec (:18-27: norm rounded to float before the clamp) agrees wherever the golden says it does"""
    from oracle import binding as ob
    if not ob.RefMath.available():
        pytest.skip("oracle/_ref/libref_math.so not built (no /root/reference here)")
    rm = ob.RefMath()
    rng = np.random.default_rng(77)
    for d in (3, 64, 129, 512):
        x = (rng.normal(size=(200, d)) * np.exp2(rng.integers(-40, 40, size=(200, 1)))).astype(np.float32)
        x[rng.random(size=x.shape) < 0.4] = 0
        want = rm.l2norm_array(x)
        got = np.stack([orc.sq8_l2norm(r) for r in x])
        assert np.array_equal(bits(got), bits(want)), d
    for g in SQ8_NORM_GROUPS:
        assert np.array_equal(bits(rm.l2norm_array(golden.sq8_norm[g + "_x"])), bits(golden.sq8_norm[g + "_array"]))
        assert np.array_equal(bits(rm.l2norm_vec(golden.sq8_norm[g + "_x"])), bits(golden.sq8_norm[g + "_vec"]))


def test_sq8_restatement_properties(orc, golden):
    """The quantise / decode formulas of scalar quantisation stay restated-only (int8_quan.cc:72-94, :117-132; faiss absent, no
    expected outputs in the reference) -- the normalisation in front of them is pinned above; here: self-consistency."""
    rng = np.random.default_rng(11)
    x = np.abs(rng.normal(size=(200, 64))).astype(np.float32)
    x[0] = golden.sq8["int8_quan_test_x"]  # the reference demo's input vector
    vmin, vdiff = orc.sq8_train(x)
    codes, xn = orc.sq8_encode(vmin, vdiff, x)
    assert np.allclose(np.linalg.norm(xn, axis=1), 1.0, atol=1e-6)
    assert codes.min() >= 0 and codes.max() == 255  # the per-dimension maximum maps to 255
    dec = orc.sq8_decode(vmin, vdiff, codes)
    assert np.max(np.abs(dec - xn) / np.maximum(vdiff, 1e-12)) <= 1.0 / 255 + 1e-6  # within one bucket
    # the faiss-path decode (Int8Decode(uint8_t*) -> sq.decode, fp32 codec): the same values to within an ulp, not the same bits
    decf = orc.sq8_decode_faiss(vmin, vdiff, codes)
    want = (vmin[None, :] + ((codes.astype(np.float32) + np.float32(0.5)) / np.float32(255.0)) * vdiff[None, :]).astype(np.float32)
    assert np.array_equal(decf.view(np.uint32), want.view(np.uint32))     # numpy float32 arithmetic = separate roundings
    ulp = np.abs(decf.view(np.int32).astype(np.int64) - dec.view(np.int32).astype(np.int64))
    assert ulp.max() <= 4 and 0.05 < (ulp != 0).mean() < 0.7   # (three fp32 roundings against one: a few ulps where vmin cancels)
    # zero-range dimension encodes to 0 (int8_quan.cc:81)
    v2 = vdiff.copy(); v2[3] = 0
    c2, _ = orc.sq8_encode(vmin, v2, x, l2norm=True)
    assert np.all(c2[:, 3] == 0)
    # zero vector: norm clamps at 1e-12 and stays zero
    z = np.zeros((1, 64), dtype=np.float32)
    _, zn = orc.sq8_encode(vmin, vdiff, z)
    assert np.all(zn == 0)


def test_merge_topk(orc):
    rng = np.random.default_rng(5)
    nq, L, k = 7, 4, 10
    d = np.sort(rng.integers(0, 6, size=(nq, L, k)).astype(np.float32), axis=2)
    ids = np.empty((nq, L, k), dtype=np.int64)
    for l in range(L):
        ids[:, l, :] = l * 1000 + np.arange(k)
    ids[:, 2, 7:] = -1
    od, oi = orc.merge_topk(d, ids, k)
    for q in range(nq):
        flat = [(d[q, l, j], ids[q, l, j]) for l in range(L) for j in range(k) if ids[q, l, j] >= 0]
        flat.sort()
        assert [x[1] for x in flat[:k]] == list(oi[q])


@pytest.mark.skipif(not __import__("oracle.binding", fromlist=["x"]).ref_available(), reason="oracle/_ref not built")
def test_oracle_against_live_reference(orc):
    """Fresh random case against the reference compiled in place (container only)."""
    from oracle import binding as ob
    rng = np.random.default_rng(99)
    D, M, K, coarseK = 64, 8, 256, 8
    vids = [rng.normal(size=(n, D)).astype(np.float32) * 0.1 for n in (30, 41, 27)]
    perm = rng.permutation(D).astype(np.int32)
    allv = np.concatenate(vids)
    coarse = allv[:, perm][rng.choice(allv.shape[0], coarseK, replace=False)].copy()
    books = rng.normal(size=(M, K, D // M)).astype(np.float32) * 0.05
    ref = ob.RefOPQ(coarse, books, perm)
    ref.index(vids)
    off, vid, codes = ref.dump()
    q = rng.normal(size=(5, D)).astype(np.float32) * 0.1
    ms = ref.query(q, 3, 3)
    ref.close()
    rot = orc.reorder(perm, allv)
    lists, oc = orc.pq_encode(rot, coarse, books)
    order = np.argsort(lists, kind="stable")
    assert np.array_equal(oc[order], codes)
    oms = orc.query_video(orc.reorder(perm, q), coarse, books, 3, off, codes, vid, 3)
    assert np.array_equal(bits(oms), bits(ms))


def test_oracle_kmeans_is_a_lloyd_fixed_point(orc):
    """orc_kmeans (the specification of cvtmi_kmeans; yael itself is not vendored): deterministic in the seed,
    converges to a fixed point, never increases the distortion, leaves NaN rows unassigned."""
    rng = np.random.default_rng(3)
    cen = rng.normal(size=(10, 6)).astype(np.float32) * 4
    x = (cen[rng.integers(0, 10, 3000)] + 0.4 * rng.normal(size=(3000, 6))).astype(np.float32)
    x[11] = np.nan
    c, a, it = orc.kmeans(x, 10, 0, 1)
    c2, a2, it2 = orc.kmeans(x, 10, 0, 1)
    assert it == it2 and np.array_equal(a, a2) and np.array_equal(c.view(np.uint32), c2.view(np.uint32))
    assert a[11] == -1 and it >= 1
    ok = a >= 0
    d = ((x[ok, None, :] - c[None]) ** 2).sum(-1)
    assert np.array_equal(d.argmin(1), a[ok])                       # assignments are nearest-centroid
    for j in range(10):
        if (a == j).any():
            assert np.allclose(c[j], x[a == j].astype(np.float64).mean(0), atol=1e-5)   # centroids are member means
    dist = [((x[ok] - orc.kmeans(x, 10, n, 1)[0][orc.kmeans(x, 10, n, 1)[1][ok]]) ** 2).sum() for n in (1, 2, 4, 8)]
    assert all(b <= a_ * (1 + 1e-6) for a_, b in zip(dist, dist[1:]))
    c3, _, _ = orc.kmeans(x, 10, 0, 2)
    assert not np.array_equal(c3, c)                                 # the seed matters


HNSW_CASES = ["ip32", "l2f16", "ip20", "l2f7", "ip128"]


@pytest.mark.parametrize("case", HNSW_CASES)
def test_oracle_hnsw_search_matches_reference(orc, golden, case):
    """orc_hnsw_search on graph files written by the reference's saveIndex == the reference's own searchKnn
    answers (labels and distance bits), duplicates / custom labels / ef < k included."""
    g = golden.hnsw
    metric, D, n, M, efc, k, ef = (int(v) for v in g[case + "_meta"])
    d, lab = orc.hnsw_search(g[case + "_index"].tobytes(), metric, D, g[case + "_q"], k, ef)
    assert np.array_equal(lab, g[case + "_l"])
    assert np.array_equal(d.view(np.uint32), g[case + "_d"].view(np.uint32))
    # a different ef / k still runs and returns sorted (dist, label) lists
    d2, lab2 = orc.hnsw_search(g[case + "_index"].tobytes(), metric, D, g[case + "_q"], 3, 5)
    assert np.all(d2[:, 1:] >= d2[:, :-1])


def test_oracle_opq_rotation_learning(orc):
    """orc_opq_learn_rotation (optional; not in the reference: self-specified): the Procrustes step recovers a known
    rotation and reaches the optimum tr(R C) = sum of singular values also for a rank-deficient C (completed columns); the learned R is
    orthonormal and quantises mixed, unbalanced rows better than the identity; outer = 0 is plain PQ training."""
    rng = np.random.default_rng(21)
    D, M, K, n = 32, 4, 16, 3000
    A = rng.normal(size=(D, D))
    x = ((rng.normal(size=(n, D)) * np.exp(-np.arange(D) / 6.0)) @ A.T).astype(np.float32)
    Q, _ = np.linalg.qr(rng.normal(size=(D, D)))
    rc, R = orc.procrustes(orc.xty(x, (x @ Q.T).astype(np.float32)))
    assert rc == 0 and np.abs(R - Q).max() < 1e-5
    C = orc.xty(x, np.where(np.arange(D)[None, :] < 5, x, 0).astype(np.float32))          # rank 5
    rc, R = orc.procrustes(C)
    Rd = R.astype(np.float64)
    assert rc == 0 and np.abs(Rd @ Rd.T - np.eye(D)).max() < 1e-6
    assert abs(np.trace(Rd @ C) - np.linalg.svd(C, compute_uv=False).sum()) < 1e-6 * np.abs(C).sum()
    assert orc.procrustes(np.zeros((D, D)))[0] == 1 and orc.procrustes(np.full((D, D), np.nan))[0] == 1
    want = x.astype(np.float64).T @ x.astype(np.float64)
    assert np.abs(orc.xty(x, x) - want).max() <= 1e-12 * np.abs(want).max()

    def distortion(R, books):
        xr = orc.rotate_fma(R, x)
        _, codes = orc.pq_encode(xr, np.zeros((1, D), np.float32), books)
        y = np.concatenate([books[m][codes[:, m]] for m in range(M)], axis=1)
        return float(((xr.astype(np.float64) - y) ** 2).sum() / n)
    R0, b0 = orc.opq_learn_rotation(x, M, K, 0, 4, 1)
    R6, b6 = orc.opq_learn_rotation(x, M, K, 6, 4, 1)
    assert np.array_equal(R0, np.eye(D, dtype=np.float32))
    assert np.abs(R6.astype(np.float64) @ R6.astype(np.float64).T - np.eye(D)).max() < 1e-6
    assert distortion(R6, b6) < 0.6 * distortion(R0, b0)
