"""CPU: graph construction of the hnswlib::HierarchicalNSW mirror (cvt_amd/host/hnswlib/hnswalg.h: addPoint +
saveIndex, a host algorithm in the reference too -- hnsw_sifts_retrieval/makeIdx.cpp:325-396 inserts one row at
a time).  The files it writes are compared byte for byte with files the reference wrote: the committed golden
graphs (tests/golden/hnsw_golden.npz), and, where oracle/_ref is built, fresh larger graphs."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "cvt_amd", "bin")
CASES = ("ip32", "l2f16", "ip20", "l2f7", "ip128")


def rows_of(blob, D):
    """rows and labels, in insertion order, out of a saveIndex file (hnswalg.h:491-519)."""
    off0, cap, cnt, per, offl, offd = struct.unpack("<6Q", blob[:48].tobytes())
    body = blob[96:96 + cap * per].reshape(cap, per)[:cnt]
    return body[:, offd:offd + 4 * D].copy().view(np.float32), body[:, offl:offl + 8].copy().view(np.uint64).ravel()


def build(tmp_path, rows, labels, D, M, efc, space, name="o", threads=None):
    assert os.path.exists(os.path.join(BIN, "hnsw_build")), "host CLIs not built: __graft_entry__.build()"
    rf = tmp_path / (name + "_rows.bin"); rf.write_bytes(np.ascontiguousarray(rows, np.float32).tobytes())
    cmd = [os.path.join(BIN, "hnsw_build"), str(rf), str(D), str(M), str(efc), str(tmp_path / (name + ".hnsw")), space]
    if labels is not None:
        lf = tmp_path / (name + "_labels.bin"); lf.write_bytes(np.ascontiguousarray(labels, np.uint64).tobytes())
        cmd.append(str(lf))
    elif threads is not None:
        cmd.append("-")
    if threads is not None:
        cmd.append(str(threads))
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return np.fromfile(tmp_path / (name + ".hnsw"), dtype=np.uint8)


@pytest.mark.parametrize("case", CASES)
def test_rebuilds_golden_graph_byte_for_byte(tmp_path, golden, case):
    g = golden.hnsw
    blob = g[case + "_index"]
    metric, D, n, M, efc, k, ef = (int(v) for v in g[case + "_meta"])
    rows, labels = rows_of(blob, D)
    assert rows.shape == (n, D)
    out = build(tmp_path, rows, labels, D, M, efc, "l2" if metric == 1 else "ip", case)
    assert out.size == blob.size and np.array_equal(out, blob)


def test_default_labels_are_row_numbers(tmp_path, golden):
    g = golden.hnsw
    blob = g["l2f7_index"]
    metric, D, n, M, efc, k, ef = (int(v) for v in g["l2f7_meta"])
    rows, labels = rows_of(blob, D)
    assert np.array_equal(labels, np.arange(n, dtype=np.uint64))
    assert np.array_equal(build(tmp_path, rows, None, D, M, efc, "l2"), blob)


def graph_of(blob, D):
    """(levels, level-0 neighbour lists) out of a saveIndex file"""
    off0, cap, cnt, per, offl, offd = struct.unpack("<6Q", blob[:48].tobytes())
    maxM, maxM0, M = struct.unpack("<3Q", blob[56:80].tobytes())
    body = blob[96:96 + cap * per].reshape(cap, per)[:cnt]
    l0 = body[:, off0:off0 + 4 + 4 * maxM0].copy().view(np.uint32)
    return int(cnt), int(maxM0), l0


@pytest.mark.parametrize("case", ("ip32", "l2f16"))
def test_add_points_one_thread_is_the_sequential_build(tmp_path, golden, case):
    """addPoints (ids and levels handed out first, then insertion under the reference's locks) with ONE worker writes the golden file"""
    g = golden.hnsw
    blob = g[case + "_index"]
    metric, D, n, M, efc, k, ef = (int(v) for v in g[case + "_meta"])
    rows, labels = rows_of(blob, D)
    out = build(tmp_path, rows, labels, D, M, efc, "l2" if metric == 1 else "ip", case, threads=1)
    assert out.size == blob.size and np.array_equal(out, blob)


def test_parallel_build_is_a_sound_graph(tmp_path, orc):
    """8 workers (hnswalg.h:594-608 lock discipline): same rows, labels and LEVELS as the sequential file (levels are drawn in row order
    before the workers start), well-formed lists (bounded, in range, no self links, no duplicates), and searchKnn over it finds the
    exact neighbour as often as over the sequential graph (within 2 %)."""
    rng = np.random.default_rng(9)
    n, D, M, efc = 20000, 32, 12, 80
    x = rng.normal(size=(n, D)).astype(np.float32)
    seq = build(tmp_path, x, None, D, M, efc, "l2", "seq")
    par = build(tmp_path, x, None, D, M, efc, "l2", "par", threads=8)
    assert seq.size == par.size
    rs, ls = rows_of(seq, D); rp, lp = rows_of(par, D)
    assert np.array_equal(rs, rp) and np.array_equal(ls, lp)
    cnt, maxM0, l0 = graph_of(par, D)
    assert cnt == n
    # level sizes section: identical (same draws)
    off0, cap, _, per, _, _ = struct.unpack("<6Q", par[:48].tobytes())
    tail_s, tail_p = seq[96 + cap * per:], par[96 + cap * per:]
    def level_sizes(t):
        out, p = [], 0
        for _ in range(cap):
            sz = int(t[p:p + 4].view(np.uint32)[0]); out.append(sz); p += 4 + sz
        return out
    assert level_sizes(tail_s) == level_sizes(tail_p)
    deg = l0[:, 0]
    assert deg.max() <= maxM0 and deg.min() >= 1
    for i in range(0, n, 97):
        nb = l0[i, 1:1 + deg[i]]
        assert nb.max() < n and i not in nb and len(set(nb.tolist())) == len(nb)
    q = rng.normal(size=(200, D)).astype(np.float32)
    exact = ((q[:, None, :] - x[None, :, :]) ** 2).sum(-1).argmin(1)
    hit = {}
    for name, blob in (("seq", seq), ("par", par)):
        _, lab = orc.hnsw_search(blob.tobytes(), 1, D, q, 1, 64)
        hit[name] = float((lab[:, 0] == exact).mean())
    assert hit["par"] >= hit["seq"] - 0.02 and hit["par"] > 0.9, hit


HAVE_REF = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref_hnsw.so"))


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("metric,D,n,M,efc", [(0, 64, 20000, 16, 100), (1, 32, 12000, 5, 40), (1, 10, 6000, 24, 30), (0, 128, 4000, 48, 200)])
def test_matches_reference_build(tmp_path, metric, D, n, M, efc):
    """fresh graphs, with exact duplicates and quantised coordinates (ties everywhere), against the reference
    compiled in place"""
    from oracle import binding as ob
    rng = np.random.default_rng(1000 + D)
    x = np.round(rng.normal(size=(n, D)) * 4).astype(np.float32) / 4
    x[n // 2:n // 2 + 200] = x[:200]
    ref_path = str(tmp_path / "ref.hnsw")
    ob.RefHnsw().build(metric, x, ref_path, M, efc)
    ref = np.fromfile(ref_path, dtype=np.uint8)
    out = build(tmp_path, x, None, D, M, efc, "l2" if metric == 1 else "ip")
    assert out.size == ref.size and np.array_equal(out, ref)
