#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference mounted and oracle/_ref built by
`make -C oracle`).  Every expected output stored here was produced by the reference's own sources
(opq/src/IVFOPQ.cpp, brute_force_search/src/*.hpp, hnsw_sifts_retrieval/hnswlib/space_l2.h) compiled
in place; the script also asserts that the C restatement (oracle/cvt_oracle.c) reproduces each of
them bit for bit before anything is written.  Fixtures are data only: inputs + expected outputs.

    python tests/golden/make_golden.py
"""
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
REF_DATA = "/root/reference/opq/data"


def unit(x):
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def make_books(rng, train, M, K):
    """Sub-codebooks sampled from (rotated) training rows plus jitter: [M][K][step]."""
    n, D = train.shape
    step = D // M
    books = np.empty((M, K, step), dtype=np.float32)
    for m in range(M):
        idx = rng.integers(0, n, size=K)
        books[m] = train[idx, m * step:(m + 1) * step] + rng.normal(0, 0.02, size=(K, step)).astype(np.float32)
    return books


def run_opq_case(name, rng, D, coarseK, M, K, videos, queries, nk, orc, scale=1.0, topk=5):
    """videos: list of raw arrays; queries: raw array.  Returns dict of arrays (inputs + ref outputs)."""
    perm = rng.permutation(D).astype(np.int32)
    allv = np.concatenate(videos, axis=0)
    rot = allv[:, perm]
    if coarseK == 1:
        coarse = np.zeros((1, D), dtype=np.float32)
    else:
        coarse = rot[rng.choice(rot.shape[0], size=coarseK, replace=coarseK > rot.shape[0])].copy()
        coarse += rng.normal(0, 0.01, size=coarse.shape).astype(np.float32)
    # residual training rows for the sub-codebooks
    assign = orc.coarse_assign(rot, coarse)
    resid = rot - coarse[assign]
    books = make_books(rng, resid, M, K)

    ref = ob.RefOPQ(coarse, books, perm)
    img_num = ref.index(videos)
    assert img_num == len(videos)
    list_off, video_id, codes = ref.dump()
    q_rot = ref.load_feat(queries)
    db_rot = ref.load_feat(allv)
    ms = ref.query(queries, nk, img_num)
    total = ms.sum(axis=0, dtype=np.float32)  # placeholder, replaced below by frame-ordered sum
    total = np.zeros(img_num, dtype=np.float32)
    for f in range(ms.shape[0]):
        total += ms[f]
    kk = min(topk, img_num)
    rank_d, rank_i = ref.sort_results(total, kk)
    # the reference's SaveIndex file (IVFOPQ.cpp:516-583) minus the trailing imgNum x char[260] path block
    import glob as _glob
    ref.save_index(ref.tmp.name)
    idx_files = _glob.glob(os.path.join(ref.tmp.name, "OPQ_Index_db_*.fvecs"))
    assert len(idx_files) == 1
    raw = np.fromfile(idx_files[0], dtype=np.uint8)
    index_head = raw[:raw.size - img_num * 260].copy()
    index_name = os.path.basename(idx_files[0])
    if os.path.exists("log.txt"):
        os.remove("log.txt")  # SaveIndex drops a log.txt in the cwd
    ref.close()

    # ---- the restatement must reproduce the reference bit for bit ----
    assert np.array_equal(orc.reorder(perm, queries), q_rot), name
    assert np.array_equal(orc.reorder(perm, allv), db_rot), name
    o_lists, o_codes = orc.pq_encode(db_rot, coarse, books)
    # reference stores entries grouped by list, in insertion order within a list
    order = np.argsort(o_lists, kind="stable")
    assert np.array_equal(o_codes[order], codes), name
    vid_of_row = np.concatenate([np.full(v.shape[0], i, dtype=np.int32) for i, v in enumerate(videos)])
    assert np.array_equal(vid_of_row[order], video_id), name
    o_off = np.zeros(coarseK + 1, dtype=np.int64)
    np.cumsum(np.bincount(o_lists, minlength=coarseK), out=o_off[1:])
    assert np.array_equal(o_off, list_off), name
    o_ms = orc.query_video(q_rot, coarse, books, nk, list_off, codes, video_id, img_num)
    assert np.array_equal(o_ms.view(np.uint32), ms.view(np.uint32)), name
    o_total, o_rd, o_ri = orc.video_rank(ms, kk)
    assert np.array_equal(o_total.view(np.uint32), total.view(np.uint32)), name
    assert np.array_equal(o_rd.view(np.uint32), rank_d.view(np.uint32)) and np.array_equal(o_ri, rank_i), name
    print("  %-14s D=%d coarseK=%d M=%d K=%d rows=%d frames=%d nk=%d  clamp-hit=%.0f%%  OK" % (
        name, D, coarseK, M, K, allv.shape[0], queries.shape[0], nk, 100.0 * np.mean(ms == 1.0)))
    return dict(perm=perm, coarse=coarse, books=books, db=allv, video_rows=np.array([v.shape[0] for v in videos]),
                queries=queries, nk=np.int32(nk), db_rot=db_rot, q_rot=q_rot, list_off=list_off,
                video_id=video_id, codes=codes, match_score=ms, total=total, rank_d=rank_d, rank_i=rank_i,
                index_head=index_head, index_name=np.frombuffer(index_name.encode(), dtype=np.uint8))


def main():
    assert ob.ref_available(), "oracle/_ref missing: run `make -C oracle` with /root/reference mounted"
    orc = ob.Oracle()
    rng = np.random.default_rng(20260927)
    fx = {}

    print("OPQ (reference: opq/src/IVFOPQ.cpp)")
    # A. C1-shaped plumbing case: exhaustive (coarseK=1, zero centroid), M=8, multi-row videos
    vids = [unit(rng.normal(size=(n, 128))) for n in (40, 64, 33, 70, 51, 42)]
    qs = np.concatenate([vids[2][5:9] + rng.normal(0, 0.05, size=(4, 128)).astype(np.float32),
                         unit(rng.normal(size=(8, 128)))]).astype(np.float32)
    fx["opq_exh_m8"] = run_opq_case("opq_exh_m8", rng, 128, 1, 8, 256, vids, qs, 1, orc)

    # B. one vector per video, scaled so every ADC score < 1.0 (no clamp): per-vector top-k form, M=16
    base = (0.4 * unit(rng.normal(size=(384, 128)))).astype(np.float32)
    base[100] = base[7]          # exact duplicates -> (dist,id) ties inside the top-k
    base[200] = base[7]
    vids = [base[i:i + 1] for i in range(base.shape[0])]
    qs = np.concatenate([base[7:8], base[300:303] + rng.normal(0, 0.01, size=(3, 128)).astype(np.float32),
                         (0.4 * unit(rng.normal(size=(12, 128)))).astype(np.float32)]).astype(np.float32)
    fx["opq_vec_m16"] = run_opq_case("opq_vec_m16", rng, 128, 1, 16, 256, vids, qs, 1, orc, topk=100)
    assert np.all(fx["opq_vec_m16"]["match_score"] < 1.0)

    # C. IVF path: coarseK=16, nprobe=3, residual LUTs, small D
    vids = [unit(rng.normal(size=(n, 32))) for n in (50, 61, 47, 55, 38)]
    qs = np.concatenate([vids[1][:6], unit(rng.normal(size=(6, 32)))]).astype(np.float32)
    fx["opq_ivf"] = run_opq_case("opq_ivf", rng, 32, 16, 4, 256, vids, qs, 3, orc)

    # D. LUT pin: M=1 -> every score is a single LUT entry (0.0f + LUT[0][code])
    vids = [(0.3 * unit(rng.normal(size=(1, 8)))).astype(np.float32) for _ in range(300)]
    qs = (0.3 * unit(rng.normal(size=(5, 8)))).astype(np.float32)
    fx["opq_m1"] = run_opq_case("opq_m1", rng, 8, 1, 1, 256, vids, qs, 1, orc, topk=10)

    # E. the reference's real feature files (opq/data): db = 5 videos, queries = its two query files
    data_dir = os.path.join(OUT, "opq_data")
    os.makedirs(os.path.join(data_dir, "db"), exist_ok=True)
    os.makedirs(os.path.join(data_dir, "query"), exist_ok=True)
    # order of opq/data/5_feats_list.txt
    names = ["6231519245", "6231075428", "6230951284", "6230880830", "6231307582"]
    vids = []
    for nme in names:
        src = os.path.join(REF_DATA, "db", nme + "_feat.bin")
        shutil.copyfile(src, os.path.join(data_dir, "db", nme + "_feat.bin"))
        os.chmod(os.path.join(data_dir, "db", nme + "_feat.bin"), 0o644)
        vids.append(np.fromfile(src, dtype=np.float32).reshape(-1, 128))
    for nme in ("6231519245_6_feat.bin", "6231519245_feat.bin"):
        shutil.copyfile(os.path.join(REF_DATA, "query", nme), os.path.join(data_dir, "query", nme))
        os.chmod(os.path.join(data_dir, "query", nme), 0o644)
    q1 = np.fromfile(os.path.join(REF_DATA, "query", "6231519245_6_feat.bin"), dtype=np.float32).reshape(-1, 128)
    q2 = np.fromfile(os.path.join(REF_DATA, "query", "6231519245_feat.bin"), dtype=np.float32).reshape(-1, 128)
    for tag, qq in (("real_q1", q1), ("real_q9", q2)):
        r2 = np.random.default_rng(77)  # same synthetic model for both query files
        fx["opq_" + tag] = run_opq_case("opq_" + tag, r2, 128, 4, 16, 256, vids, qq, 3, orc)
        for key in ("db", "queries"):  # raw inputs live in opq_data/*.bin, do not duplicate them
            del fx["opq_" + tag][key]
    # known answer: query row == row 7 of video 0 -> video 0 must rank first
    assert fx["opq_real_q1"]["rank_i"][0] == 0 and fx["opq_real_q9"]["rank_i"][0] == 0

    np.savez_compressed(os.path.join(OUT, "opq_golden.npz"),
                        **{"%s/%s" % (c, k): v for c, d in fx.items() for k, v in d.items()})

    print("Brute force (reference: brute_force_search/src, hnsw_sifts_retrieval/hnswlib/space_l2.h)")
    rf = ob.RefFlat()
    fl = {}
    # IP, D=128 (SIMD16 path), k=100 as in brute_force.cpp:14-15; duplicates + self-queries
    db = unit(rng.normal(size=(700, 128)))
    db[11] = db[3]; db[650] = db[3]; db[651] = db[400]
    qs = np.concatenate([db[3:4], db[400:401], unit(rng.normal(size=(10, 128)))]).astype(np.float32)
    d, i = rf.search(ob.IP, db, qs, 100)
    od, _, oi = orc.flat_search(ob.IP, db, qs, 100, flavour=4)
    assert np.array_equal(od.view(np.uint32), d.view(np.uint32)) and np.array_equal(oi, i), "IP restatement"
    fl.update(ip_db=db, ip_q=qs, ip_d=d, ip_i=i)
    # IP with non-sequential labels (label != row)
    labels = rng.permutation(5000)[:700].astype(np.int64)
    d, i = rf.search(ob.IP, db, qs, 10, labels=labels)
    od, _, oi = orc.flat_search(ob.IP, db, qs, 10, labels=labels, flavour=4)
    assert np.array_equal(od.view(np.uint32), d.view(np.uint32)) and np.array_equal(oi, i)
    fl.update(ipl_labels=labels, ipl_d=d, ipl_i=i)
    # L2 float, D=64
    db = rng.normal(size=(500, 64)).astype(np.float32)
    db[20] = db[19]
    qs = np.concatenate([db[19:20], rng.normal(size=(7, 64)).astype(np.float32)])
    d, i = rf.search(ob.L2F, db, qs, 10)
    od, _, oi = orc.flat_search(ob.L2F, db, qs, 10, flavour=8)
    assert np.array_equal(od.view(np.uint32), d.view(np.uint32)) and np.array_equal(oi, i), "L2F restatement"
    fl.update(l2_db=db, l2_q=qs, l2_d=d, l2_i=i)
    # L2 uint8: D=64 with few distinct values -> many exact ties; D=30 -> dim%4 tail dropped
    for tag, D, hi in (("u8a", 64, 4), ("u8b", 30, 256), ("u8c", 512, 256)):
        db = rng.integers(0, hi, size=(400, D), dtype=np.uint8)
        qs = np.concatenate([db[5:7], rng.integers(0, hi, size=(6, D), dtype=np.uint8)])
        d, i = rf.search(ob.L2U8, db, qs, 10)
        od, odi, oi = orc.flat_search(ob.L2U8, db, qs, 10)
        assert np.array_equal(odi, d) and np.array_equal(oi, i), "L2U8 restatement " + tag
        fl.update({tag + "_db": db, tag + "_q": qs, tag + "_d": d, tag + "_i": i})
    np.savez_compressed(os.path.join(OUT, "flat_golden.npz"), **fl)
    print("  IP / IP+labels / L2F / L2U8 x3  OK")

    # SQ8: int8_quan.cc / sq_train.cpp cannot be built here (faiss 1.5.3 absent) -> their quantise / decode formulas stay
    # restated-only.  What CAN run is the normalisation in front of them: utils/math_util.h:29-39 (MathUtil::L2NormArray,
    # header-only) is the same arithmetic as Int8Quan::L2NormalizeVector (int8_quan.cc:46-56) -> oracle/_ref/libref_math.so.
    np.savez_compressed(os.path.join(OUT, "sq8_inputs.npz"), int8_quan_test_x=sq_in)
    make_sq8_norm(orc, sq_in)
    # HNSW: graphs BUILT AND SAVED by the reference's HierarchicalNSW (oracle/_ref/libref_hnsw.so =
    # hnswalg.h compiled in place), queries answered by its own searchKnn.  The saved index files are data the
    # reference wrote, committed as fixtures; the C restatement must reproduce every answer bit for bit.
    import tempfile
    rh = ob.RefHnsw()
    hn = {}
    cases = (("ip32", ob.IP, 32, 2500, 8, 40, (10, 50), True), ("l2f16", ob.L2F, 16, 2000, 6, 30, (5, 10), False),
             ("ip20", ob.IP, 20, 1500, 8, 40, (7, 200), True), ("l2f7", ob.L2F, 7, 800, 4, 20, (3, 30), False),
             ("ip128", ob.IP, 128, 1200, 16, 80, (5, 1000), True))
    for name, metric, D, n, M, efc, (k, ef), norm in cases:
        x = rng.normal(size=(n, D)).astype(np.float32)
        if norm:
            x = unit(x)
        x[100] = x[7]; x[101] = x[7]; x[500] = x[7]      # exact duplicates: equal distances, heap order decides
        labels = (np.arange(n, dtype=np.int64) * 3 + 11) if name == "ip20" else None
        q = rng.normal(size=(48, D)).astype(np.float32)
        if norm:
            q = unit(q)
        q[0] = x[7]
        path = os.path.join(tempfile.gettempdir(), "cvt_golden_%s.hnsw" % name)
        rh.build(metric, x, path, M, efc, labels=labels)
        rd, rl = rh.search(metric, D, path, q, k, ef)
        blob = np.fromfile(path, dtype=np.uint8)
        od, ol = orc.hnsw_search(blob.tobytes(), metric, D, q, k, ef)
        assert np.array_equal(rl, ol) and np.array_equal(rd.view(np.uint32), od.view(np.uint32)), name
        hn[name + "_index"] = blob; hn[name + "_q"] = q; hn[name + "_d"] = rd; hn[name + "_l"] = rl
        hn[name + "_meta"] = np.array([metric, D, n, M, efc, k, ef], dtype=np.int64)
        os.remove(path)
    np.savez_compressed(os.path.join(OUT, "hnsw_golden.npz"), **hn)
    print("  HNSW x%d (reference-built graphs, reference answers)  OK" % len(cases))
    make_pca(orc)
    print("wrote", OUT)


def make_sq8_norm(orc, sq_in):
    """rows normalised by the reference's own MathUtil::L2NormArray / L2NormVec -> tests/golden/sq8_norm_golden.npz.  Row groups
    (one width each): the reference's demo vector (int8_quan_test.cpp:26), CNN-like rows at d = 64 / 512 / 2048 / 37, and the
    corners of the formula: all-zero rows (0 / 1e-12 -> 0), norms below the 1e-12 clamp, denormal entries, entries whose
    squares overflow fp32 (the product is a FLOAT product: inf -> 0 or NaN rows), a norm that overflows only as a sum of
    finite squares (never in double), -0.0, one huge entry among small ones, NaN / inf entries."""
    rm = ob.RefMath()
    rng = np.random.default_rng(0x5108)   # its own stream: the other sections' draws do not move
    out = {}

    def put(tag, rows):
        rows = np.atleast_2d(np.asarray(rows, dtype=np.float32))
        ref = rm.l2norm_array(rows)
        mine = np.stack([orc.sq8_l2norm(r) for r in rows])
        assert np.array_equal(ref.view(np.uint32), mine.view(np.uint32)), "sq8_l2norm restatement: " + tag
        out[tag + "_x"] = rows; out[tag + "_array"] = ref; out[tag + "_vec"] = rm.l2norm_vec(rows)

    put("demo64", sq_in)
    for d in (64, 512, 2048, 37):
        put("cnn%d" % d, np.maximum(rng.normal(size=(48, d)), 0).astype(np.float32) * rng.uniform(0.01, 30, size=(48, 1)).astype(np.float32))
    d = 64
    rows = np.zeros((12, d), np.float32)
    rows[1, 3] = 5e-13                       # norm below the clamp: divided by float(1e-12)
    rows[2, :] = 1e-14                       # norm 8e-14
    rows[3, :4] = (1e-40, 3e-41, 1.4e-45, 7e-42)   # denormal entries: their float squares are 0
    rows[4, :] = 1e-22; rows[4, 0] = 1e-20   # squares are denormal floats
    rows[5, :3] = (3e19, 1e19, -2e19)        # squares near FLT_MAX / overflow -> inf float product -> norm inf -> 0
    rows[6, :] = 1.5e19                      # every square finite (2.25e38), the sum exceeds FLT_MAX but not double
    rows[7, 0] = 3.0e38; rows[7, 1] = 1.0    # one square overflows
    rows[8, :] = -0.0; rows[8, 5] = -2.0
    rows[9, :] = 1e-3; rows[9, 17] = 1e30
    rows[10, 2] = np.nan; rows[10, 3] = 1.0
    rows[11, 2] = np.inf; rows[11, 3] = 1.0; rows[11, 4] = -np.inf
    with np.errstate(all="ignore"):
        put("corners64", rows)
        # norms straddling the clamp: float(1e-12) = 9.99999996e-13 sits BELOW 1e-12, so L2NormArray (clamp in double, then
        # round) and L2NormVec (round, then clamp in double) can part ways here -- both are recorded
        put("clamp16", np.array([[v] + [0.0] * 15 for v in np.float32(1e-12) * (1 + np.arange(-6, 7) * np.float32(2.0 ** -23))]))
    np.savez_compressed(os.path.join(OUT, "sq8_norm_golden.npz"), **out)
    print("  SQ8 normalisation (MathUtil::L2NormArray / L2NormVec, %d row groups)  OK" % (len(out) // 3))


def read_opencv_matrix(text, name):
    """one `name: !!opencv-matrix` node of an OpenCV FileStorage YAML -> fp32 array"""
    import re
    m = re.search(r"^%s: !!opencv-matrix\s+rows: (\d+)\s+cols: (\d+)\s+dt: (\w)\s+data: \[(.*?)\]" % name, text, re.S | re.M)
    vals = np.array([float(v) for v in m.group(4).replace("\n", " ").split(",")], dtype=np.float64)
    return vals.astype(np.float32).reshape(int(m.group(1)), int(m.group(2)))


def make_pca(orc):
    """PCA.  The reference's projection is an OpenCV call and OpenCV is not installed: no expected
    outputs can be produced here (PARITY UNPINNED).  The fixture holds DATA only: the reference's own model
    (pca_train_project/model/pca_1024_128_300w_googlenet.yml: `vectors` 128 x 1024, `mean`, `values`) as fp32 arrays,
    inputs are seeded in the tests and the checker's outputs are recomputed there, not stored."""
    print("PCA (model data of pca_train_project/model; OpenCV absent -> unpinned)")
    text = open("/root/reference/pca_train_project/model/pca_1024_128_300w_googlenet.yml").read()
    vectors, mean, values = (read_opencv_matrix(text, k) for k in ("vectors", "mean", "values"))
    assert vectors.shape == (128, 1024) and mean.shape == (1, 1024) and values.shape == (128, 1)
    rng = np.random.default_rng(0x9CA)
    x = np.maximum(rng.normal(size=(300, 1024)), 0).astype(np.float32) * rng.gamma(2.0, 1.0, size=(1, 1024)).astype(np.float32)
    x = unit(x)                                   # pooled CNN features: non-negative, sparse-ish, unit norm
    x[7] = mean[0]                                # projects to (almost) zero: the 1e-12 clamp region
    y = orc.pca_project(mean, vectors, x, True, flavour=0)
    assert np.all(np.abs(np.linalg.norm(y[:7].astype(np.float64), axis=1) - 1) < 1e-6)
    np.savez_compressed(os.path.join(OUT, "pca_model.npz"), vectors=vectors, mean=mean, values=values)
    print("  model 1024 -> 128  OK")


if __name__ == "__main__":
    if sys.argv[1:] == ["pca"]:
        make_pca(ob.Oracle())
    elif sys.argv[1:] == ["sq8_norm"]:
        make_sq8_norm(ob.Oracle(), np.load(os.path.join(OUT, "sq8_inputs.npz"))["int8_quan_test_x"])
    else:
        main()
