"""GPU: the C++ mirror of the reference classes (cvt_amd/host: IVFOPQ, hnswlib::BruteforceSearch,
cvtk::quant::Int8Quan) driven through its CLIs, the way the reference's own mains drive the originals.
Expected outputs are the golden vectors produced by the reference (index file bytes, ranks, top-k)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "cvt_amd", "bin")
NAMES = ["6231519245", "6231075428", "6230951284", "6230880830", "6231307582"]  # opq/data/5_feats_list.txt order


def run(args, cwd):
    r = subprocess.run(args, cwd=cwd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (args, r.stdout[-2000:], r.stderr[-2000:])
    return r.stdout


def write_model(path, g):
    from oracle import binding as ob
    ob.write_opq_model(path, g["coarse"], g["books"], g["perm"])


def test_opq_index_and_query_cli(tmp_path, golden):
    for exe in ("opq_index", "opq_query"):
        assert os.path.exists(os.path.join(BIN, exe)), "host CLIs not built: __graft_entry__.build()"
    g = golden.opq["opq_real_q9"]
    model = str(tmp_path / "model.bin")
    write_model(model, g)
    lst = tmp_path / "feats.txt"
    lst.write_text("\n".join(os.path.join(golden.dir, "opq_data", "db", n + "_feat.bin") for n in NAMES) + "\n")
    out_dir = tmp_path / "out"; out_dir.mkdir()
    run([os.path.join(BIN, "opq_index"), model, str(lst), str(out_dir)], cwd=str(tmp_path))
    name = bytes(g["index_name"]).decode()
    idx_file = out_dir / name
    assert idx_file.exists(), os.listdir(out_dir)                       # same file name as the reference builds
    raw = np.fromfile(idx_file, dtype=np.uint8)
    head = raw[:raw.size - 5 * 260]
    assert np.array_equal(head, g["index_head"]), "index file differs from the reference's SaveIndex bytes"
    paths = raw[raw.size - 5 * 260:].reshape(5, 260)
    assert bytes(paths[0]).split(b"\0")[0].decode().endswith(NAMES[0] + "_feat.bin")
    # query: both reference query files, nprobe 3, top 5
    for case, qf in (("opq_real_q1", "6231519245_6_feat.bin"), ("opq_real_q9", "6231519245_feat.bin")):
        gg = golden.opq[case]
        res = tmp_path / ("res_%s.txt" % case)
        run([os.path.join(BIN, "opq_query"), model, str(idx_file), str(res),
             os.path.join(golden.dir, "opq_data", "query", qf), "--nearest", "3", "--show", "5"], cwd=str(tmp_path))
        lines = res.read_text().splitlines()
        names = lines[1].split()
        scores = [float(x) for x in lines[2].split()]
        assert names == [NAMES[i] + "_feat" for i in gg["rank_i"]]
        assert np.allclose(scores, gg["rank_d"], rtol=2e-5)            # text output: 6 significant digits
    assert names[0] == "6231519245_feat"                                # the known answer of SURVEY.md 4


def write_records(path, ids, x):
    with open(path, "wb") as f:
        f.write(struct.pack("i", len(ids)))
        for s, row in zip(ids, x):
            b = s.encode()
            f.write(struct.pack("i", len(b))); f.write(b)
            f.write(struct.pack("i", row.shape[0])); f.write(np.ascontiguousarray(row, dtype=np.float32).tobytes())


def test_brute_force_cli(tmp_path, golden):
    f = golden.flat
    db, q = f["ip_db"], f["ip_q"]
    write_records(tmp_path / "db.bin", ["id%05d" % i for i in range(db.shape[0])], db)
    write_records(tmp_path / "querys.bin", ["q%03d" % i for i in range(q.shape[0])], q)
    run([os.path.join(BIN, "brute_force"), "db.bin", "querys.bin", "index.bin", "gt.txt", "128", "100"], cwd=str(tmp_path))
    lines = (tmp_path / "gt.txt").read_text().splitlines()
    assert len(lines) == q.shape[0]
    for qi, line in enumerate(lines):
        head, rest = line.split(" topK: ")
        ids_s, d_s = rest.split("dists: ")
        ids = [int(t[2:]) for t in ids_s.split()]
        ds = np.array([float(t) for t in d_s.split()])
        assert head == "q%03d" % qi
        assert ids == list(f["ip_i"][qi])                                # same neighbours, same order as the reference
        assert np.allclose(ds, 1.0 - f["ip_d"][qi].astype(np.float64), rtol=1e-5, atol=1e-6)
    # index.bin layout: size_t max, size_t per_elem, size_t count, rows of [vector][size_t label]
    raw = (tmp_path / "index.bin").read_bytes()
    mx, per, cnt = struct.unpack("QQQ", raw[:24])
    assert (mx, per, cnt) == (db.shape[0], 128 * 4 + 8, db.shape[0])
    row3 = np.frombuffer(raw[24 + 3 * per:24 + 3 * per + 512], dtype=np.float32)
    assert np.array_equal(row3, db[3]) and struct.unpack("Q", raw[24 + 3 * per + 512:24 + 4 * per])[0] == 3


def test_sq8_cli(tmp_path, orc, golden):
    rng = np.random.default_rng(6)
    d, n = 64, 500
    x = np.abs(rng.normal(size=(n, d))).astype(np.float32)
    x[0] = golden.sq8["int8_quan_test_x"]
    write_records(tmp_path / "feats.bin", ["v%d" % i for i in range(n)], x)
    run([os.path.join(BIN, "sq_train"), "feats.bin", "model.bin", str(d)], cwd=str(tmp_path))
    raw = (tmp_path / "model.bin").read_bytes()
    assert struct.unpack("i", raw[:4])[0] == d
    vmin = np.frombuffer(raw[4:4 + 4 * d], dtype=np.float32); vdiff = np.frombuffer(raw[4 + 4 * d:], dtype=np.float32)
    ovmin, ovdiff = orc.sq8_train(x)
    assert np.array_equal(bits(vmin), bits(ovmin)) and np.array_equal(bits(vdiff), bits(ovdiff))
    out = run([os.path.join(BIN, "int8_quan_demo"), "model.bin"], cwd=str(tmp_path))
    codes = [int(t) for t in [l for l in out.splitlines() if l.startswith("int8: ")][0].split()[1:]]
    ocodes, _ = orc.sq8_encode(ovmin, ovdiff, x[:1])
    assert codes == list(ocodes[0])
    # Int8Decode(uint8_t*): the faiss-path arithmetic (fp32), not the double formula of Int8Decode(std::string&)
    fb = [int(t, 16) for t in [l for l in out.splitlines() if l.startswith("decoded_faiss_bits:")][0].split()[1:]]
    want = orc.sq8_decode_faiss(ovmin, ovdiff, ocodes[:1])
    assert fb == [int(v) for v in bits(want)[0]]
    # the same through the faiss "IxSQ" container (what sq_train.cpp:103 writes and Int8Quan(model_path) loads, int8_quan.cc:14)
    run([os.path.join(BIN, "sq_train"), "feats.bin", "model_faiss.bin", str(d), "--faiss"], cwd=str(tmp_path))
    raw = (tmp_path / "model_faiss.bin").read_bytes()
    assert raw[:4] == b"IxSQ" and struct.unpack("<i", raw[4:8])[0] == d and len(raw) == 4 + 33 + 36 + 8 * d + 8
    assert np.array_equal(np.frombuffer(raw[73:73 + 8 * d], dtype=np.float32), np.concatenate([vmin, vdiff]))
    assert run([os.path.join(BIN, "int8_quan_demo"), "model_faiss.bin"], cwd=str(tmp_path)) == out


def test_opq_train_cli(tmp_path, orc):
    """train_PQ's main with its 8 arguments: reorder file of `long int`, raw fp32 features in, model file out.
    The model equals the oracle's training of the permuted rows bit for bit (yael is not vendored: the
    reference's own centroids are unpinned), carries the reference's file layout incl. its truncated reorder
    block, and is loadable by the IVFOPQ mirror (opq_index)."""
    assert os.path.exists(os.path.join(BIN, "opq_train")), "host CLIs not built: __graft_entry__.build()"
    rng = np.random.default_rng(5)
    D, M, K, coarseK, n = 32, 4, 32, 3, 2500
    cen = rng.normal(size=(12, D)).astype(np.float32) * 2
    x = (cen[rng.integers(0, 12, n + 100)] + 0.3 * rng.normal(size=(n + 100, D))).astype(np.float32)
    perm = rng.permutation(D).astype(np.int64)
    (tmp_path / "reorder.bin").write_bytes(perm.tobytes())
    (tmp_path / "feats.bin").write_bytes(x.tobytes())
    out = run([os.path.join(BIN, "opq_train"), str(tmp_path / "reorder.bin"), str(tmp_path / "feats.bin"), str(tmp_path),
               str(n), str(coarseK), str(D), str(M), str(K)], cwd=str(tmp_path))
    path = tmp_path / ("OPQ_db_%d_dim_%d_k_%d_PQ_m%d_k%d.fvecs" % (n, D, coarseK, M, K))
    assert path.exists(), out
    raw = path.read_bytes()
    assert struct.unpack("4i", raw[:16]) == (D, coarseK, M, K)
    o = 16
    coarse = np.frombuffer(raw, np.float32, coarseK * D, o).reshape(coarseK, D); o += coarse.nbytes
    books = np.frombuffer(raw, np.float32, K * D, o).reshape(M, K, D // M); o += books.nbytes
    assert raw[o:] == perm.tobytes()[:4 * D] and len(raw) == o + 4 * D     # train_PQ_codebook.cpp:287
    oc, ob = orc.opq_train(x[:n][:, perm], coarseK, M, K, 0, 1)
    assert np.array_equal(bits(coarse), bits(oc)) and np.array_equal(bits(books), bits(ob))
    # the file is a valid LoadModel input: index two small "videos" with it
    proper = tmp_path / "model_int32.bin"
    proper.write_bytes(raw[:o] + perm.astype(np.int32).tobytes())
    v0 = tmp_path / "v0_feat.bin"; v0.write_bytes(x[:40].tobytes())
    v1 = tmp_path / "v1_feat.bin"; v1.write_bytes(x[40:100].tobytes())
    (tmp_path / "list.txt").write_text("%s\n%s\n" % (v0, v1))
    (tmp_path / "idx").mkdir()
    run([os.path.join(BIN, "opq_index"), str(proper), str(tmp_path / "list.txt"), str(tmp_path / "idx")], cwd=str(tmp_path))
    assert any(f.startswith("OPQ_Index_db_2_dim_%d_k_%d_PQ_m%d_k%d" % (D, coarseK, M, K)) for f in os.listdir(tmp_path / "idx"))


def test_opq_learned_rotation_cli(tmp_path, orc):
    """opq_train --learn-rotation (TrainPQ::LearnRotation, not in the reference) -> model + <model>.R.f32; opq_search --rotation
    (IVFOPQ::LoadRotation) indexes and searches under it.  R and the codebooks equal the oracle's orc_opq_learn_rotation bit for bit,
    the search lists equal the oracle's ADC search over the oracle's codes of the rotated rows."""
    rng = np.random.default_rng(15)
    D, M, K, n, nq, k = 32, 4, 32, 3000, 40, 10
    A = rng.normal(size=(D, D))
    x = ((rng.normal(size=(n + nq, D)) * np.exp(-np.arange(D) / 6.0)) @ A.T).astype(np.float32)
    db, q = x[:n], x[n:]
    (tmp_path / "reorder.bin").write_bytes(np.arange(D, dtype=np.int64).tobytes())     # identity: the rotation does the work
    (tmp_path / "feats.bin").write_bytes(db.tobytes())
    run([os.path.join(BIN, "opq_train"), str(tmp_path / "reorder.bin"), str(tmp_path / "feats.bin"), str(tmp_path), str(n), "1", str(D),
         str(M), str(K), "--learn-rotation=3"], cwd=str(tmp_path))
    model = tmp_path / ("OPQ_db_%d_dim_%d_k_1_PQ_m%d_k%d.fvecs" % (n, D, M, K))
    raw = model.read_bytes()
    books = np.frombuffer(raw, np.float32, K * D, 16 + 4 * D).reshape(M, K, D // M)
    R = np.frombuffer((tmp_path / (model.name + ".R.f32")).read_bytes(), np.float32).reshape(D, D)
    oR, ob = orc.opq_learn_rotation(db, M, K, 3, 0, 1)
    assert np.array_equal(bits(R), bits(oR)) and np.array_equal(bits(books), bits(ob))
    assert np.all(np.frombuffer(raw, np.float32, D, 16) == 0)                          # the zero coarse centroid
    proper = tmp_path / "model_int32.bin"                                              # LoadModel reads int32 reorder[D]
    proper.write_bytes(raw[:16 + 4 * D + 4 * K * D] + np.arange(D, dtype=np.int32).tobytes())
    (tmp_path / "q.bin").write_bytes(q.tobytes())
    run([os.path.join(BIN, "opq_search"), str(proper), str(tmp_path / "feats.bin"), str(tmp_path / "q.bin"), str(tmp_path / "res.txt"),
         "--k", str(k), "--rotation", str(tmp_path / (model.name + ".R.f32"))], cwd=str(tmp_path))
    _, codes = orc.pq_encode(orc.rotate_fma(oR, db), np.zeros((1, D), np.float32), ob)
    od, oi = orc.adc_search(orc.rotate_fma(oR, q), ob, codes, k)
    lines = (tmp_path / "res.txt").read_text().strip().splitlines()
    assert len(lines) == nq
    for qi, line in enumerate(lines):
        ids = [int(t) for t in line.split("topK:")[1].split("dists:")[0].split()]
        assert ids == list(oi[qi]), qi


def test_opq_learned_rotation_with_a_reorder_file(tmp_path, orc):
    """opq_train permutes the sample by the reorder file BEFORE it learns R, so R and the codebooks belong to
    y = R (P x); opq_search --rotation applies the saved matrix to the raw vector and drops the permutation.  The saved matrix must
    therefore be the product R P -- with a non-identity reorder file: the saved matrix is the oracle's R (learned on the permuted
    sample) with its columns moved, bit for bit, and the search lists equal the oracle's under that matrix."""
    rng = np.random.default_rng(16)
    D, M, K, n, nq, k = 32, 4, 32, 3000, 40, 10
    A = rng.normal(size=(D, D))
    x = ((rng.normal(size=(n + nq, D)) * np.exp(-np.arange(D) / 6.0)) @ A.T).astype(np.float32)
    db, q = x[:n], x[n:]
    perm = rng.permutation(D)
    assert not np.array_equal(perm, np.arange(D))
    (tmp_path / "reorder.bin").write_bytes(perm.astype(np.int64).tobytes())
    (tmp_path / "feats.bin").write_bytes(db.tobytes())
    run([os.path.join(BIN, "opq_train"), str(tmp_path / "reorder.bin"), str(tmp_path / "feats.bin"), str(tmp_path), str(n), "1", str(D),
         str(M), str(K), "--learn-rotation=3"], cwd=str(tmp_path))
    model = tmp_path / ("OPQ_db_%d_dim_%d_k_1_PQ_m%d_k%d.fvecs" % (n, D, M, K))
    raw = model.read_bytes()
    books = np.frombuffer(raw, np.float32, K * D, 16 + 4 * D).reshape(M, K, D // M)
    Rf = np.frombuffer((tmp_path / (model.name + ".R.f32")).read_bytes(), np.float32).reshape(D, D)
    oR, ob = orc.opq_learn_rotation(np.ascontiguousarray(db[:, perm]), M, K, 3, 0, 1)     # what the trainer saw: (P x)[n] = x[perm[n]]
    folded = np.zeros_like(oR)
    folded[:, perm] = oR                                                                  # R'[i][perm[n]] = R[i][n]
    assert np.array_equal(bits(Rf), bits(folded)) and np.array_equal(bits(books), bits(ob))
    proper = tmp_path / "model_int32.bin"
    proper.write_bytes(raw[:16 + 4 * D + 4 * K * D] + perm.astype(np.int32).tobytes())    # a reorder the rotation must override
    (tmp_path / "q.bin").write_bytes(q.tobytes())
    run([os.path.join(BIN, "opq_search"), str(proper), str(tmp_path / "feats.bin"), str(tmp_path / "q.bin"), str(tmp_path / "res.txt"),
         "--k", str(k), "--rotation", str(tmp_path / (model.name + ".R.f32"))], cwd=str(tmp_path))
    _, codes = orc.pq_encode(orc.rotate_fma(folded, db), np.zeros((1, D), np.float32), ob)
    od, oi = orc.adc_search(orc.rotate_fma(folded, q), ob, codes, k)
    lines = (tmp_path / "res.txt").read_text().strip().splitlines()
    assert len(lines) == nq
    for qi, line in enumerate(lines):
        ids = [int(t) for t in line.split("topK:")[1].split("dists:")[0].split()]
        assert ids == list(oi[qi]), qi
    # and the model is a good one: the learned rotation beats the plain permutation on quantisation error of the database rows
    y = orc.rotate_fma(folded, db)
    rec = np.concatenate([ob[m][codes[:, m]] for m in range(M)], axis=1)
    err_rot = float(((y - rec) ** 2).sum(1).mean())
    assert err_rot < float((db ** 2).sum(1).mean())


def test_hnsw_search_cli(tmp_path, golden):
    """hnswlib::HierarchicalNSW mirror (loadIndex + setEf + searchKnnBatch) through its CLI, on a graph file the
    reference wrote: labels and distances of every query equal the reference's own answers."""
    assert os.path.exists(os.path.join(BIN, "hnsw_search")), "host CLIs not built: __graft_entry__.build()"
    g = golden.hnsw
    for case, space in (("ip32", "ip"), ("l2f16", "l2")):
        metric, D, n, M, efc, k, ef = (int(v) for v in g[case + "_meta"])
        idx = tmp_path / (case + ".hnsw"); idx.write_bytes(g[case + "_index"].tobytes())
        qf = tmp_path / (case + "_q.bin"); qf.write_bytes(np.ascontiguousarray(g[case + "_q"], np.float32).tobytes())
        out = tmp_path / (case + "_out.txt")
        run([os.path.join(BIN, "hnsw_search"), str(idx), str(qf), str(D), str(k), str(ef), str(out), space], cwd=str(tmp_path))
        lines = out.read_text().splitlines()
        assert len(lines) == g[case + "_q"].shape[0]
        for qi, line in enumerate(lines):
            pairs = [p.split(":") for p in line.split()]
            assert [int(p[0]) for p in pairs] == list(g[case + "_l"][qi])
            assert np.array_equal(np.array([float(p[1]) for p in pairs], np.float32).view(np.uint32), g[case + "_d"][qi].view(np.uint32))


def test_hnsw_build_then_search_cli(tmp_path, golden):
    """hnsw_build (host addPoint + saveIndex of the mirror) -> hnsw_search (GPU) on the file it wrote: the
    reference's answers for the graph the reference built from the same rows."""
    import struct
    g = golden.hnsw
    case = "ip20"
    metric, D, n, M, efc, k, ef = (int(v) for v in g[case + "_meta"])
    blob = g[case + "_index"]
    off0, cap, cnt, per, offl, offd = struct.unpack("<6Q", blob[:48].tobytes())
    body = blob[96:96 + cap * per].reshape(cap, per)
    (tmp_path / "rows.bin").write_bytes(body[:, offd:offd + 4 * D].tobytes())
    (tmp_path / "labels.bin").write_bytes(body[:, offl:offl + 8].tobytes())
    run([os.path.join(BIN, "hnsw_build"), "rows.bin", str(D), str(M), str(efc), "built.hnsw", "ip", "labels.bin"], cwd=str(tmp_path))
    assert (tmp_path / "built.hnsw").read_bytes() == blob.tobytes()
    (tmp_path / "q.bin").write_bytes(np.ascontiguousarray(g[case + "_q"], np.float32).tobytes())
    run([os.path.join(BIN, "hnsw_search"), "built.hnsw", "q.bin", str(D), str(k), str(ef), "out.txt", "ip"], cwd=str(tmp_path))
    for qi, line in enumerate((tmp_path / "out.txt").read_text().splitlines()):
        assert [int(p.split(":")[0]) for p in line.split()] == list(g[case + "_l"][qi])


@pytest.mark.parametrize("gpus,transport", [(1, "rccl"), (2, "shm"), (3, "shm"), (1, "single")])
def test_opq_search_cli_row_sharded(tmp_path, orc, gpus, transport):
    """opq_search --gpus N --fork: one process per rank (forked by the CLI), each indexes its row block of the feature file, one
    all-gather inside libcvtmi, rank 0 writes the global top-k.  N = 1 goes through real RCCL; N > 1 on this one-GPU box
    through the shm transport (RCCL refuses two ranks per device).  "single": the default form, ONE process driving the
    devices through IVFOPQ::SetDevices (this box: one).  Expected = the oracle over the whole file."""
    from oracle import binding as ob
    exe = os.path.join(BIN, "opq_search")
    assert os.path.exists(exe), "host CLIs not built: __graft_entry__.build()"
    rng = np.random.default_rng(5 + gpus)
    D, M, K, n, nq, k = 128, 16, 256, 30_001, 21, 20
    perm = rng.permutation(D).astype(np.int32)
    books = (rng.normal(size=(M, K, D // M)) * 0.05).astype(np.float32)
    coarse = np.zeros((1, D), np.float32)
    x = (rng.normal(size=(n, D)) * 0.1).astype(np.float32)
    x[n // 2 - 3:n // 2 + 3] = x[n // 2]  # duplicates across the shard boundary
    x[n // 3 - 2:n // 3 + 2] = x[n // 3]
    q = (rng.normal(size=(nq, D)) * 0.1).astype(np.float32)
    q[0] = x[n // 2]
    model = str(tmp_path / "model.bin")
    ob.write_opq_model(model, coarse, books, perm)
    x.tofile(tmp_path / "db.bin"); q.tofile(tmp_path / "q.bin")
    if transport == "single":
        out = run([exe, model, "db.bin", "q.bin", "res.txt", "--k", str(k), "--gpus", str(gpus)], cwd=str(tmp_path))
        assert "device(s) of one process" in out, out
    else:
        out = run([exe, model, "db.bin", "q.bin", "res.txt", "--k", str(k), "--gpus", str(gpus), "--fork", "--transport", transport], cwd=str(tmp_path))
        assert ("transport %s" % transport) in out and ("all-gathers 1 x" in out), out
    xr, qr = orc.reorder(perm, x), orc.reorder(perm, q)
    _, codes = orc.pq_encode(xr, coarse, books)
    od, oi = orc.adc_search(qr, books, codes, k)
    lines = (tmp_path / "res.txt").read_text().splitlines()
    assert len(lines) == nq
    for qi, line in enumerate(lines):
        head, rest = line.split(" topK: ")
        ids_s, d_s = rest.split(" dists: ")
        assert int(head) == qi
        assert [int(v) for v in ids_s.split()] == oi[qi].tolist()
        assert np.array_equal(bits(np.array([float(v) for v in d_s.split()], dtype=np.float32)), bits(od[qi]))


def test_bruteforce_mirror_incremental_sync(tmp_path):
    """hnswlib::BruteforceSearch (host mirror): searches interleaved with addPoint / removePoint answer like an index built from
    scratch -- ascending labels are appended to the device copy, anything else re-sorts it (cvt_amd/host/cli/bf_sync_check.cpp)."""
    out = run([os.path.join(BIN, "bf_sync_check")], cwd=str(tmp_path))
    assert out.strip().endswith("OK"), out


def test_opq_search_cli_failing_rank_does_not_hang(tmp_path):
    """a rank that fails (here: every rank, the model file does not exist -- rank 0 before it could publish the RCCL id) must not
    leave its peers or the parent waiting: the CLI comes back with a non-zero status well inside the timeout"""
    import subprocess, time
    exe = os.path.join(BIN, "opq_search")
    np.zeros((10, 128), np.float32).tofile(tmp_path / "db.bin")
    t0 = time.time()
    for extra in (["--fork", "--transport", "rccl"], ["--fork", "--transport", "shm"]):
        p = subprocess.run([exe, "missing_model.bin", "db.bin", "db.bin", "res.txt", "--gpus", "3"] + extra, cwd=str(tmp_path),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
        assert p.returncode != 0
    assert time.time() - t0 < 100


def test_bruteforce_mirror_concurrent_searchknn(tmp_path):
    """four host threads inside BruteforceSearch::searchKnn on ONE index at once: same answers as one thread (the library leases
    a scratch set and a stream per call; only index mutation is exclusive)"""
    out = run([os.path.join(BIN, "bf_concurrent"), "200000", "128", "4", "32", "50"], cwd=str(tmp_path))
    assert out.strip().endswith("OK"), out


def test_ivfopq_mirror_concurrent_searchtopk(tmp_path):
    """four host threads inside IVFOPQ::SearchTopK (8 queries per call, the reference's call pattern: 1-9 query frames per Query,
    opq/src/multi_frame_index_test.cpp:45-54) on ONE index at once: the answers of one thread, bit for bit -- every search leases its
    own scratch set and stream from the OPQ handle, only add / reset and the lazily built row copy are exclusive"""
    out = run([os.path.join(BIN, "opq_concurrent"), "300000", "4", "24", "8", "100"], cwd=str(tmp_path))
    assert out.strip().endswith("OK"), out
    # larger batches through the pipelined host-pointer entry (chunks alternating between two scratch sets / streams)
    out = run([os.path.join(BIN, "opq_concurrent"), "100000", "3", "4", "5000", "10"], cwd=str(tmp_path))
    assert out.strip().endswith("OK"), out
