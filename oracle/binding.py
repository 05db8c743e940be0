"""ctypes binding of the CPU oracle (oracle/cvt_oracle.c) and, when present, of the reference's own
sources compiled in place (oracle/_ref/*.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by bench.py's
``cpu_baseline`` leg.  Nothing under cvt_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
IP, L2F, L2U8 = 0, 1, 2


def build(o3=False):
    """Compile the C restatement (and oracle/_ref when /root/reference is mounted)."""
    targets = ["libcvt_oracle.so", "ref"] + (["libcvt_oracle_o3.so"] if o3 else [])
    subprocess.run(["make", "-s", "-C", _HERE] + targets, check=True, capture_output=True)


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Oracle:
    def __init__(self, o3=False):
        name = "libcvt_oracle_o3.so" if o3 else "libcvt_oracle.so"
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build(o3=o3)
        self.lib = C.CDLL(path)
        self.lib.orc_dist.restype = C.c_float
        self.lib.orc_topk_pairs.restype = C.c_int64

    # ---- OPQ ----
    def reorder(self, perm, x):
        x = _f32(x); n, D = x.shape
        perm = np.ascontiguousarray(perm, dtype=np.int32)
        y = np.empty_like(x)
        self.lib.orc_reorder(_p(perm, C.c_int32), C.c_int(D), _p(x, C.c_float), C.c_int64(n), _p(y, C.c_float))
        return y

    def rotate_fma(self, R, x):
        x = _f32(x); n, D = x.shape
        R = _f32(R)
        y = np.empty_like(x)
        self.lib.orc_rotate_fma(_p(R, C.c_float), C.c_int(D), _p(x, C.c_float), C.c_int64(n), _p(y, C.c_float))
        return y

    def pca_project(self, mean, vectors, x, l2norm=True, flavour=0):
        """flavour 0: OpenCV's built-in gemm (double accumulators); 1: the k-ordered fmaf chain (kernel spec)"""
        x = _f32(x); vectors = _f32(vectors); mean = _f32(mean).ravel()
        n, din = x.shape; dout = vectors.shape[0]
        assert vectors.shape[1] == din and mean.size == din
        y = np.empty((n, dout), np.float32)
        self.lib.orc_pca_project(_p(mean, C.c_float), _p(vectors, C.c_float), C.c_int(din), C.c_int(dout), _p(x, C.c_float),
                                 C.c_int64(n), C.c_int(1 if l2norm else 0), C.c_int(flavour), _p(y, C.c_float))
        return y

    def coarse_assign(self, x, coarse):
        x = _f32(x); coarse = _f32(coarse)
        n, D = x.shape
        out = np.empty(n, dtype=np.int32)
        self.lib.orc_coarse_assign(_p(x, C.c_float), C.c_int64(n), C.c_int(D), _p(coarse, C.c_float),
                                   C.c_int(coarse.shape[0]), _p(out, C.c_int32))
        return out

    def pq_encode(self, x, coarse, books):
        """x [n][D] (rotated), coarse [coarseK][D], books [M][K][step] -> (list_id, codes [n][M])"""
        x = _f32(x); coarse = _f32(coarse); books = _f32(books)
        n, D = x.shape
        M, K, _ = books.shape
        lists = np.empty(n, dtype=np.int32)
        codes = np.empty((n, M), dtype=np.uint8)
        self.lib.orc_pq_encode(_p(x, C.c_float), C.c_int64(n), C.c_int(D), _p(coarse, C.c_float),
                               C.c_int(coarse.shape[0]), _p(books, C.c_float), C.c_int(M), C.c_int(K),
                               _p(lists, C.c_int32), _p(codes, C.c_uint8))
        return lists, codes

    def lut(self, q, centroid, books):
        q = _f32(q); books = _f32(books)
        M, K, _ = books.shape
        cen = None if centroid is None else _f32(centroid)
        out = np.empty((M, K), dtype=np.float32)
        self.lib.orc_lut(_p(q, C.c_float), C.c_int(q.shape[0]), _p(cen, C.c_float), _p(books, C.c_float),
                         C.c_int(M), C.c_int(K), _p(out, C.c_float))
        return out

    def adc_scan(self, lut, codes):
        lut = _f32(lut); codes = np.ascontiguousarray(codes, dtype=np.uint8)
        M, K = lut.shape
        n = codes.shape[0]
        out = np.empty(n, dtype=np.float32)
        self.lib.orc_adc_scan(_p(lut, C.c_float), C.c_int(M), C.c_int(K), _p(codes, C.c_uint8), C.c_int64(n),
                              _p(out, C.c_float))
        return out

    def topk_pairs(self, scores, k, ids=None):
        scores = _f32(scores)
        n = scores.shape[0]
        ids_a = None if ids is None else np.ascontiguousarray(ids, dtype=np.int64)
        d = np.empty(k, dtype=np.float32); i = np.empty(k, dtype=np.int64)
        m = self.lib.orc_topk_pairs(_p(scores, C.c_float), _p(ids_a, C.c_int64), C.c_int64(n), C.c_int64(k),
                                    _p(d, C.c_float), _p(i, C.c_int64))
        return d[:m], i[:m]

    def adc_search(self, q, books, codes, k, centroid=None, id_base=0):
        q = _f32(q); books = _f32(books); codes = np.ascontiguousarray(codes, dtype=np.uint8)
        nq, D = q.shape
        M, K, _ = books.shape
        cen = None if centroid is None else _f32(centroid)
        d = np.empty((nq, k), dtype=np.float32); i = np.empty((nq, k), dtype=np.int64)
        self.lib.orc_adc_search(_p(q, C.c_float), C.c_int64(nq), C.c_int(D), _p(cen, C.c_float),
                                _p(books, C.c_float), C.c_int(M), C.c_int(K), _p(codes, C.c_uint8),
                                C.c_int64(codes.shape[0]), C.c_int64(id_base), C.c_int64(k),
                                _p(d, C.c_float), _p(i, C.c_int64))
        return d, i

    def query_video(self, q, coarse, books, nk, list_off, codes, video_id, img_num):
        q = _f32(q); coarse = _f32(coarse); books = _f32(books)
        nq, D = q.shape
        M, K, _ = books.shape
        list_off = np.ascontiguousarray(list_off, dtype=np.int64)
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        video_id = np.ascontiguousarray(video_id, dtype=np.int32)
        ms = np.empty((nq, img_num), dtype=np.float32)
        self.lib.orc_query_video(_p(q, C.c_float), C.c_int64(nq), C.c_int(D), _p(coarse, C.c_float),
                                 C.c_int(coarse.shape[0]), _p(books, C.c_float), C.c_int(M), C.c_int(K),
                                 C.c_int(nk), _p(list_off, C.c_int64), _p(codes, C.c_uint8),
                                 _p(video_id, C.c_int32), C.c_int(img_num), _p(ms, C.c_float))
        return ms

    def video_rank(self, match_score, k):
        ms = _f32(match_score)
        nq, img = ms.shape
        total = np.empty(img, dtype=np.float32)
        kk = min(k, img)
        d = np.empty(kk, dtype=np.float32); i = np.empty(kk, dtype=np.int64)
        self.lib.orc_video_rank(_p(ms, C.c_float), C.c_int64(nq), C.c_int(img), C.c_int64(kk),
                                _p(total, C.c_float), _p(d, C.c_float), _p(i, C.c_int64))
        return total, d, i

    # ---- flat ----
    def dist(self, metric, flavour, a, b):
        if metric == L2U8:
            a = np.ascontiguousarray(a, dtype=np.uint8); b = np.ascontiguousarray(b, dtype=np.uint8)
        else:
            a = _f32(a); b = _f32(b)
        return float(self.lib.orc_dist(C.c_int(metric), C.c_int(flavour), a.ctypes.data_as(C.c_void_p),
                                       b.ctypes.data_as(C.c_void_p), C.c_int(a.shape[0])))

    def flat_search(self, metric, data, queries, k, labels=None, flavour=4):
        dt = np.uint8 if metric == L2U8 else np.float32
        data = np.ascontiguousarray(data, dtype=dt); queries = np.ascontiguousarray(queries, dtype=dt)
        n, D = data.shape
        nq = queries.shape[0]
        lab = None if labels is None else np.ascontiguousarray(labels, dtype=np.int64)
        d = np.empty((nq, k), dtype=np.float32); di = np.empty((nq, k), dtype=np.int32)
        i = np.empty((nq, k), dtype=np.int64)
        self.lib.orc_flat_search(C.c_int(metric), C.c_int(flavour), C.c_int(D), data.ctypes.data_as(C.c_void_p),
                                 _p(lab, C.c_int64), C.c_int64(n), queries.ctypes.data_as(C.c_void_p),
                                 C.c_int64(nq), C.c_int64(k), _p(d, C.c_float), _p(di, C.c_int32),
                                 _p(i, C.c_int64))
        return d, di, i

    # ---- SQ8 ----
    def sq8_l2norm(self, v):
        v = _f32(v).copy()
        self.lib.orc_sq8_l2norm(_p(v, C.c_float), C.c_int(v.shape[0]))
        return v

    def sq8_encode(self, vmin, vdiff, x, l2norm=True):
        """returns (codes, x_after) -- x_after is the in-place-normalised input"""
        x = _f32(x).copy(); vmin = _f32(vmin); vdiff = _f32(vdiff)
        n, d = x.shape
        out = np.empty((n, d), dtype=np.uint8)
        self.lib.orc_sq8_encode(_p(vmin, C.c_float), _p(vdiff, C.c_float), C.c_int(d), _p(x, C.c_float),
                                C.c_int64(n), C.c_int(1 if l2norm else 0), _p(out, C.c_uint8))
        return out, x

    def sq8_decode(self, vmin, vdiff, codes):
        codes = np.ascontiguousarray(codes, dtype=np.uint8); vmin = _f32(vmin); vdiff = _f32(vdiff)
        n, d = codes.shape
        out = np.empty((n, d), dtype=np.float32)
        self.lib.orc_sq8_decode(_p(vmin, C.c_float), _p(vdiff, C.c_float), C.c_int(d), _p(codes, C.c_uint8),
                                C.c_int64(n), _p(out, C.c_float))
        return out

    def sq8_decode_faiss(self, vmin, vdiff, codes):
        codes = np.ascontiguousarray(codes, dtype=np.uint8); vmin = _f32(vmin); vdiff = _f32(vdiff)
        n, d = codes.shape
        out = np.empty((n, d), dtype=np.float32)
        self.lib.orc_sq8_decode_faiss(_p(vmin, C.c_float), _p(vdiff, C.c_float), C.c_int(d), _p(codes, C.c_uint8),
                                C.c_int64(n), _p(out, C.c_float))
        return out

    def sq8_train(self, x, l2norm=True):
        x = _f32(x).copy()
        n, d = x.shape
        vmin = np.empty(d, dtype=np.float32); vdiff = np.empty(d, dtype=np.float32)
        self.lib.orc_sq8_train(_p(x, C.c_float), C.c_int64(n), C.c_int(d), C.c_int(1 if l2norm else 0),
                               _p(vmin, C.c_float), _p(vdiff, C.c_float))
        return vmin, vdiff

    def kmeans(self, x, k, niter=0, seed=1):
        x = _f32(x); n, d = x.shape
        cent = np.empty((k, d), np.float32); assign = np.empty(n, np.int32); it = C.c_int(0)
        rc = self.lib.orc_kmeans(_p(x, C.c_float), C.c_int64(d), C.c_int64(n), C.c_int(d), C.c_int(k), C.c_int(niter),
                                 C.c_uint64(seed), _p(cent, C.c_float), _p(assign, C.c_int32), C.byref(it))
        assert rc == 0
        return cent, assign, it.value

    def opq_train(self, x, coarseK, M, K, niter=0, seed=1):
        x = _f32(x); n, D = x.shape
        coarse = np.empty((coarseK, D), np.float32); books = np.empty((M, K, D // M), np.float32)
        rc = self.lib.orc_opq_train(_p(x, C.c_float), C.c_int64(n), C.c_int(D), C.c_int(coarseK), C.c_int(M), C.c_int(K),
                                    C.c_int(niter), C.c_uint64(seed), _p(coarse, C.c_float), _p(books, C.c_float))
        assert rc == 0
        return coarse, books

    def xty(self, x, y):
        x = _f32(x); y = _f32(y); n, D = x.shape
        C_ = np.empty((D, D), np.float64)
        self.lib.orc_xty(_p(x, C.c_float), _p(y, C.c_float), C.c_int64(n), C.c_int(D), _p(C_, C.c_double))
        return C_

    def procrustes(self, Cm):
        Cm = np.ascontiguousarray(Cm, dtype=np.float64); D = Cm.shape[0]
        R = np.zeros((D, D), np.float32)
        rc = self.lib.orc_procrustes(_p(Cm, C.c_double), C.c_int(D), _p(R, C.c_float))
        return rc, R

    def opq_learn_rotation(self, x, M, K, outer, niter=0, seed=1):
        """(R [D][D], books [M][K][D/M]) -- the specification of cvtmi_opq_learn_rotation (not in the reference)"""
        x = _f32(x); n, D = x.shape
        R = np.empty((D, D), np.float32); books = np.empty((M, K, D // M), np.float32)
        rc = self.lib.orc_opq_learn_rotation(_p(x, C.c_float), C.c_int64(n), C.c_int(D), C.c_int(M), C.c_int(K), C.c_int(outer),
                                             C.c_int(niter), C.c_uint64(seed), _p(R, C.c_float), _p(books, C.c_float))
        assert rc == 0
        return R, books

    def hnsw_search(self, index_bytes, metric, D, queries, k, ef):
        """searchKnn over a graph file written by the reference's saveIndex; ascending (dist, label)."""
        buf = np.frombuffer(index_bytes, dtype=np.uint8)
        q = _f32(queries); nq = q.shape[0]
        d = np.empty((nq, k), np.float32); lab = np.empty((nq, k), np.int64)
        rc = self.lib.orc_hnsw_search(_p(buf, C.c_uint8), C.c_int64(buf.size), C.c_int(metric), C.c_int(D), _p(q, C.c_float),
                                      C.c_int64(nq), C.c_int64(k), C.c_int64(ef), _p(d, C.c_float), _p(lab, C.c_int64))
        assert rc == 0
        return d, lab

    def hnsw_search_adc(self, index_bytes, books, codes, q_rot, k, ef):
        buf = np.frombuffer(index_bytes, dtype=np.uint8)
        q = _f32(q_rot); nq, D = q.shape
        books = _f32(books); M, K, _ = books.shape
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        d = np.empty((nq, k), np.float32); lab = np.empty((nq, k), np.int64)
        rc = self.lib.orc_hnsw_search_adc(_p(buf, C.c_uint8), C.c_int64(buf.size), C.c_int(D), _p(books, C.c_float), C.c_int(M),
                                          C.c_int(K), _p(codes, C.c_uint8), _p(q, C.c_float), C.c_int64(nq), C.c_int64(k),
                                          C.c_int64(ef), _p(d, C.c_float), _p(lab, C.c_int64))
        assert rc == 0
        return d, lab

    def merge_topk(self, in_d, in_id, k):
        in_d = _f32(in_d); in_id = np.ascontiguousarray(in_id, dtype=np.int64)
        nq, L, kk = in_d.shape
        assert kk == k
        d = np.empty((nq, k), dtype=np.float32); i = np.empty((nq, k), dtype=np.int64)
        self.lib.orc_merge_topk(_p(in_d, C.c_float), _p(in_id, C.c_int64), C.c_int64(nq), C.c_int64(L),
                                C.c_int64(k), _p(d, C.c_float), _p(i, C.c_int64))
        return d, i


# ------------------------------------------------------------------------------------------------
# The reference itself (oracle/_ref), file-driven like its own mains.
# ------------------------------------------------------------------------------------------------
def ref_available():
    return all(os.path.exists(os.path.join(_HERE, "_ref", n))
               for n in ("libref_opq.so", "libref_bf_ip.so", "libref_bf_l2.so"))


def write_opq_model(path, coarse, books, perm):
    """Model file layout read by IVFOPQ::LoadModel (opq/src/IVFOPQ.cpp:75-95)."""
    coarse = _f32(coarse); books = _f32(books)
    coarseK, D = coarse.shape
    M, K, step = books.shape
    assert M * step == D
    with open(path, "wb") as f:
        np.array([D, coarseK, M, K], dtype=np.int32).tofile(f)
        coarse.tofile(f); books.tofile(f)
        np.ascontiguousarray(perm, dtype=np.int32).tofile(f)


class RefOPQ:
    """Drives the reference's IVFOPQ through oracle/_ref/libref_opq.so."""

    def __init__(self, coarse, books, perm, max_index_num=1 << 20):
        self.lib = C.CDLL(os.path.join(_HERE, "_ref", "libref_opq.so"))
        self.lib.ref_opq_new.restype = C.c_void_p
        self.lib.ref_opq_dump.restype = C.c_int64
        self.tmp = tempfile.TemporaryDirectory(prefix="cvt_ref_")
        self.D = coarse.shape[1]; self.coarseK = coarse.shape[0]; self.M = books.shape[0]
        mp = os.path.join(self.tmp.name, "model.bin")
        write_opq_model(mp, coarse, books, perm)
        self.h = C.c_void_p(self.lib.ref_opq_new(mp.encode(), C.c_int(max_index_num)))
        assert self.h.value, "reference LoadModel failed"
        self._nfile = 0

    def _feat_file(self, x):
        p = os.path.join(self.tmp.name, "feat_%d.bin" % self._nfile)
        self._nfile += 1
        _f32(x).tofile(p)
        return p

    def load_feat(self, x):
        x = _f32(x)
        out = np.empty_like(x)
        n = self.lib.ref_opq_load_feat(self.h, self._feat_file(x).encode(), _p(out, C.c_float), C.c_int(x.shape[0]))
        assert n == x.shape[0]
        return out

    def index(self, videos):
        """videos: list of [n_i][D] raw (un-rotated) arrays, one per 'video' file."""
        files = [self._feat_file(v).encode() for v in videos]
        arr = (C.c_char_p * len(files))(*files)
        return self.lib.ref_opq_index(self.h, arr, C.c_int(len(files)))

    def dump(self):
        tot = self.lib.ref_opq_dump(self.h, None, None, None, C.c_int64(0))
        off = np.empty(self.coarseK + 1, dtype=np.int64)
        vid = np.empty(tot, dtype=np.int32)
        codes = np.empty((tot, self.M), dtype=np.uint8)
        self.lib.ref_opq_dump(self.h, _p(off, C.c_int64), _p(vid, C.c_int32), _p(codes, C.c_uint8), C.c_int64(tot))
        return off, vid, codes

    def query(self, q, nk, img_num):
        q = _f32(q)
        ms = np.empty((q.shape[0], img_num), dtype=np.float32)
        fr = C.c_int(0); im = C.c_int(0)
        self.lib.ref_opq_query(self.h, self._feat_file(q).encode(), C.c_int(nk), _p(ms, C.c_float),
                               C.c_int64(ms.size), C.byref(fr), C.byref(im))
        assert fr.value == q.shape[0] and im.value == img_num, (fr.value, im.value)
        return ms

    def save_index(self, directory):
        self.lib.ref_opq_save_index(self.h, directory.encode())

    def sort_results(self, score, k):
        score = _f32(score)
        d = np.empty(k, dtype=np.float32); i = np.empty(k, dtype=np.uint32)
        self.lib.ref_sort_results(_p(score, C.c_float), C.c_int(score.shape[0]), C.c_int(k), _p(d, C.c_float),
                                  _p(i, C.c_uint32))
        return d, i.astype(np.int64)

    def close(self):
        if self.h.value:
            self.lib.ref_opq_delete(self.h); self.h = C.c_void_p(0)
        self.tmp.cleanup()


class RefFlat:
    def __init__(self):
        self.ip = C.CDLL(os.path.join(_HERE, "_ref", "libref_bf_ip.so"))
        self.l2 = C.CDLL(os.path.join(_HERE, "_ref", "libref_bf_l2.so"))
        self.ip.ref_ip_dist.restype = C.c_float

    def search(self, metric, data, queries, k, labels=None):
        lab = None if labels is None else np.ascontiguousarray(labels, dtype=np.int64)
        if metric == L2U8:
            data = np.ascontiguousarray(data, dtype=np.uint8); queries = np.ascontiguousarray(queries, dtype=np.uint8)
            d = np.empty((queries.shape[0], k), dtype=np.int32)
            i = np.empty((queries.shape[0], k), dtype=np.int64)
            self.l2.ref_bf_l2u8_search(C.c_int(data.shape[1]), _p(data, C.c_uint8), _p(lab, C.c_int64),
                                       C.c_int64(data.shape[0]), _p(queries, C.c_uint8),
                                       C.c_int64(queries.shape[0]), C.c_int64(k), _p(d, C.c_int32), _p(i, C.c_int64))
            return d, i
        data = _f32(data); queries = _f32(queries)
        d = np.empty((queries.shape[0], k), dtype=np.float32)
        i = np.empty((queries.shape[0], k), dtype=np.int64)
        fn = self.ip.ref_bf_ip_search if metric == IP else self.l2.ref_bf_l2f_search
        fn(C.c_int(data.shape[1]), _p(data, C.c_float), _p(lab, C.c_int64), C.c_int64(data.shape[0]),
           _p(queries, C.c_float), C.c_int64(queries.shape[0]), C.c_int64(k), _p(d, C.c_float), _p(i, C.c_int64))
        return d, i


    def ip_search_timed(self, data, queries, k):
        """the reference's BruteforceSearch<float> + InnerProductSpace: (dist, labels, seconds of addPoint, seconds of searchKnn)"""
        data = _f32(data); queries = _f32(queries)
        d = np.empty((queries.shape[0], k), dtype=np.float32)
        i = np.empty((queries.shape[0], k), dtype=np.int64)
        ta, ts = C.c_double(0), C.c_double(0)
        self.ip.ref_bf_ip_search_timed(C.c_int(data.shape[1]), _p(data, C.c_float), C.c_int64(data.shape[0]), _p(queries, C.c_float),
                                       C.c_int64(queries.shape[0]), C.c_int64(k), _p(d, C.c_float), _p(i, C.c_int64),
                                       C.byref(ta), C.byref(ts))
        return d, i, ta.value, ts.value


class RefMath:
    """cvtk::MathUtil of the reference's utils/math_util.h compiled in place (oracle/_ref/libref_math.so): the L2
    normalisation in front of Int8Encode / SQ training, the one part of that path that runs here without faiss."""

    def __init__(self):
        self.lib = C.CDLL(os.path.join(_HERE, "_ref", "libref_math.so"))

    @staticmethod
    def available():
        return os.path.exists(os.path.join(_HERE, "_ref", "libref_math.so"))

    def l2norm_array(self, x):
        x = np.atleast_2d(_f32(x)).copy()
        assert self.lib.ref_l2norm_array(_p(x, C.c_float), C.c_int64(x.shape[0]), C.c_int(x.shape[1])) == 0
        return x

    def l2norm_vec(self, x):
        x = np.atleast_2d(_f32(x))
        out = np.empty_like(x)
        assert self.lib.ref_l2norm_vec(_p(x, C.c_float), C.c_int64(x.shape[0]), C.c_int(x.shape[1]), _p(out, C.c_float)) == 0
        return out


class RefHnsw:
    """The reference's HierarchicalNSW compiled in place (oracle/_ref/libref_hnsw.so)."""
    def __init__(self):
        self.lib = C.CDLL(os.path.join(_HERE, "_ref", "libref_hnsw.so"))

    def build(self, metric, data, path, M=16, ef_construction=200, labels=None, max_elements=None, threads=1):
        data = _f32(data); n, D = data.shape
        lab = None if labels is None else np.ascontiguousarray(labels, dtype=np.int64)
        if threads > 1:  # the reference's addPoint from several host threads (its own per-node locks): order-dependent graph
            rc = self.lib.ref_hnsw_build_mt(C.c_int(metric), C.c_int(D), _p(data, C.c_float), _p(lab, C.c_int64), C.c_int64(n),
                                            C.c_int64(max_elements or n), C.c_int(M), C.c_int(ef_construction), path.encode(),
                                            C.c_int(threads))
            assert rc == 0
            return
        rc = self.lib.ref_hnsw_build(C.c_int(metric), C.c_int(D), _p(data, C.c_float), _p(lab, C.c_int64), C.c_int64(n),
                                     C.c_int64(max_elements or n), C.c_int(M), C.c_int(ef_construction), path.encode())
        assert rc == 0

    def search(self, metric, D, path, queries, k, ef):
        q = _f32(queries); nq = q.shape[0]
        d = np.empty((nq, k), np.float32); lab = np.empty((nq, k), np.int64)
        rc = self.lib.ref_hnsw_search(C.c_int(metric), C.c_int(D), path.encode(), _p(q, C.c_float), C.c_int64(nq), C.c_int64(k),
                                      C.c_int64(ef), _p(d, C.c_float), _p(lab, C.c_int64))
        assert rc == 0
        return d, lab
