"""ctypes binding of include/cvtmi.h.

There is NO fallback: if libcvtmi.so is missing or a call fails, an exception is raised.  Arrays
may be numpy (host-pointer entry points) or torch CUDA tensors (``*_dev`` entry points on the
tensor's device, launched on torch's current stream so torch events/streams order them).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CVTMI_LIB") or os.path.join(_HERE, "lib", "libcvtmi.so")   # CVTMI_LIB: an instrumented / experimental build
_lib = None


class CvtmiError(RuntimeError):
    pass


def _torch_first():
    """One HIP runtime per process.  libcvtmi.so asks the loader for libamdhip64 / libhsa-runtime64 by SONAME; a PyTorch-ROCm wheel carries
    its own copies under torch/lib.  Whoever loads first decides which copy the SONAME resolves to, and a process that ends up with the
    system libamdhip64 over the wheel's HSA runtime (library first, torch second) sees "no ROCm-capable device".  So where torch is
    installed it is imported before the library; without torch (a C / ctypes host) the system ROCm is the only runtime and nothing is done."""
    import sys
    if "torch" in sys.modules:
        return
    import importlib.util
    if importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CvtmiError("libcvtmi.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "or `make -C cvt_amd/csrc`" % LIB_PATH)
        _torch_first()
        _lib = C.CDLL(LIB_PATH)
        _lib.cvtmi_last_error.restype = C.c_char_p
    return _lib


def lib():
    return load_library()


def _check(rc):
    if rc != 0:
        raise CvtmiError("cvtmi error %d: %s" % (rc, lib().cvtmi_last_error().decode(errors="replace")))


def _is_torch(a):
    return type(a).__module__.startswith("torch")


def _ptr(a):
    if a is None:
        return C.c_void_p(0)
    if _is_torch(a):
        assert a.is_contiguous()
        return C.c_void_p(a.data_ptr())
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data)


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _np(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def device_count():
    n = C.c_int(0)
    _check(lib().cvtmi_device_count(C.byref(n)))
    return n.value


class OpqIndex:
    """cvtmi_opq_t: OPQ model (coarse, sub-codebooks, rotation) + HBM-resident code index."""

    def __init__(self, coarse, books, perm=None, R=None):
        coarse = _np(coarse, np.float32); books = _np(books, np.float32)
        self.coarseK, self.D = coarse.shape
        self.M, self.K, step = books.shape
        assert self.M * step == self.D
        perm_a = None if perm is None else _np(perm, np.int32)
        R_a = None if R is None else _np(R, np.float32)
        self.h = C.c_void_p(0)
        _check(lib().cvtmi_opq_create(C.c_int(self.D), C.c_int(self.coarseK), C.c_int(self.M), C.c_int(self.K),
                                      _ptr(coarse), _ptr(books), _ptr(R_a), _ptr(perm_a), C.byref(self.h)))

    def close(self):
        if self.h and self.h.value:
            lib().cvtmi_opq_destroy(self.h)
            self.h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- rotate ----
    def rotate(self, x):
        if _is_torch(x):
            import torch
            y = torch.empty_like(x)
            _check(lib().cvtmi_opq_rotate_dev(self.h, _ptr(x), C.c_int64(x.shape[0]), _ptr(y), _stream()))
            return y
        x = _np(x, np.float32)
        y = np.empty_like(x)
        _check(lib().cvtmi_opq_rotate(self.h, _ptr(x), C.c_int64(x.shape[0]), _ptr(y)))
        return y

    # ---- encode ----
    def encode(self, x_rot):
        n = x_rot.shape[0]
        if _is_torch(x_rot):
            import torch
            lists = torch.empty(n, dtype=torch.int32, device=x_rot.device)
            codes = torch.empty((n, self.M), dtype=torch.uint8, device=x_rot.device)
            _check(lib().cvtmi_opq_encode_dev(self.h, _ptr(x_rot), C.c_int64(n), _ptr(lists), _ptr(codes), _stream()))
            return lists, codes
        x_rot = _np(x_rot, np.float32)
        lists = np.empty(n, dtype=np.int32)
        codes = np.empty((n, self.M), dtype=np.uint8)
        _check(lib().cvtmi_opq_encode(self.h, _ptr(x_rot), C.c_int64(n), _ptr(lists), _ptr(codes)))
        return lists, codes

    def rotate_encode(self, x):
        """rotate + encode of RAW rows (cvtmi_opq_rotate_encode): (list ids, codes)"""
        n = x.shape[0]
        if _is_torch(x):
            import torch
            lists = torch.empty(n, dtype=torch.int32, device=x.device)
            codes = torch.empty((n, self.M), dtype=torch.uint8, device=x.device)
            _check(lib().cvtmi_opq_rotate_encode_dev(self.h, _ptr(x), C.c_int64(n), _ptr(lists), _ptr(codes), _stream()))
            return lists, codes
        x = _np(x, np.float32)
        lists = np.empty(n, dtype=np.int32)
        codes = np.empty((n, self.M), dtype=np.uint8)
        _check(lib().cvtmi_opq_rotate_encode(self.h, _ptr(x), C.c_int64(n), _ptr(lists), _ptr(codes)))
        return lists, codes

    # ---- index ----
    def add_codes(self, codes, list_id=None, video_id=None):
        n = codes.shape[0]
        if _is_torch(codes):
            _check(lib().cvtmi_opq_add_codes_dev(self.h, _ptr(codes), _ptr(list_id), _ptr(video_id), C.c_int64(n),
                                                 _stream()))
            return
        codes = _np(codes, np.uint8)
        l = None if list_id is None else _np(list_id, np.int32)
        v = None if video_id is None else _np(video_id, np.int32)
        _check(lib().cvtmi_opq_add_codes(self.h, _ptr(codes), _ptr(l), _ptr(v), C.c_int64(n)))

    def reserve(self, n):
        _check(lib().cvtmi_opq_reserve(self.h, C.c_int64(n)))

    def reset(self):
        _check(lib().cvtmi_opq_reset(self.h))

    @property
    def ntotal(self):
        n = C.c_int64(0)
        _check(lib().cvtmi_opq_ntotal(self.h, C.byref(n)))
        return n.value

    def set_id_base(self, base):
        _check(lib().cvtmi_opq_set_id_base(self.h, C.c_int64(base)))

    def get_entries(self):
        n = self.ntotal
        off = np.empty(self.coarseK + 1, dtype=np.int64)
        vid = np.empty(n, dtype=np.int32)
        codes = np.empty((n, self.M), dtype=np.uint8)
        _check(lib().cvtmi_opq_get_entries(self.h, _ptr(off), _ptr(vid), _ptr(codes)))
        tot = int(off[-1])
        return off, vid[:tot], codes[:tot]

    # ---- query ----
    def lut(self, q_rot, list_id=None):
        nq = q_rot.shape[0]
        if _is_torch(q_rot):
            import torch
            out = torch.empty((nq, self.M, self.K), dtype=torch.float32, device=q_rot.device)
            _check(lib().cvtmi_opq_lut_dev(self.h, _ptr(q_rot), C.c_int64(nq), _ptr(list_id), _ptr(out), _stream()))
            return out
        q_rot = _np(q_rot, np.float32)
        l = None if list_id is None else _np(list_id, np.int32)
        out = np.empty((nq, self.M, self.K), dtype=np.float32)
        _check(lib().cvtmi_opq_lut(self.h, _ptr(q_rot), C.c_int64(nq), _ptr(l), _ptr(out)))
        return out

    def search(self, q, k, rotate=True, out=None):
        nq = q.shape[0]
        if _is_torch(q):
            import torch
            if out is None:
                d = torch.empty((nq, k), dtype=torch.float32, device=q.device)
                i = torch.empty((nq, k), dtype=torch.int64, device=q.device)
            else:
                d, i = out
            _check(lib().cvtmi_opq_search_dev(self.h, _ptr(q), C.c_int64(nq), C.c_int(1 if rotate else 0), C.c_int(k),
                                              _ptr(d), _ptr(i), _stream()))
            return d, i
        q = _np(q, np.float32)
        if out is None:
            d = np.empty((nq, k), dtype=np.float32); i = np.empty((nq, k), dtype=np.int64)
        else:   # the caller's own (already touched) arrays: freshly allocated ones cost a page fault per 4 KB written
            d, i = out
            assert d.dtype == np.float32 and i.dtype == np.int64 and d.shape == (nq, k) and i.shape == (nq, k)
            assert d.flags.c_contiguous and i.flags.c_contiguous
        _check(lib().cvtmi_opq_search(self.h, _ptr(q), C.c_int64(nq), C.c_int(1 if rotate else 0), C.c_int(k),
                                      _ptr(d), _ptr(i)))
        return d, i

    def search_sharded(self, comm, q, k, rotate=True):
        """Row-sharded search: this handle holds the rank's row block; one all-gather + merge inside the library.
        Device tensors: nothing synchronises, so another rank's failure cannot be known when this returns -- its results are
        voided on the device (+inf / -1) and the error is raised by the next call on `comm`, by comm.status() after a
        synchronise, or at the latest by comm.close()."""
        nq = q.shape[0]
        if _is_torch(q):
            import torch
            d = torch.empty((nq, k), dtype=torch.float32, device=q.device)
            i = torch.empty((nq, k), dtype=torch.int64, device=q.device)
            comm.check(lib().cvtmi_opq_search_sharded_dev(self.h, comm.h, _ptr(q), C.c_int64(nq), C.c_int(1 if rotate else 0),
                                                      C.c_int(k), _ptr(d), _ptr(i), _stream()))
            return d, i
        q = _np(q, np.float32)
        d = np.empty((nq, k), dtype=np.float32); i = np.empty((nq, k), dtype=np.int64)
        comm.check(lib().cvtmi_opq_search_sharded(self.h, comm.h, _ptr(q), C.c_int64(nq), C.c_int(1 if rotate else 0), C.c_int(k),
                                              _ptr(d), _ptr(i)))
        return d, i

    def query_video(self, q, nprobe, img_num, rotate=True):
        if _is_torch(q):
            import torch
            ms = torch.empty((q.shape[0], img_num), dtype=torch.float32, device=q.device)
            _check(lib().cvtmi_opq_query_video_dev(self.h, _ptr(q), C.c_int64(q.shape[0]), C.c_int(1 if rotate else 0),
                                                   C.c_int(nprobe), C.c_int(img_num), _ptr(ms), _stream()))
            return ms
        q = _np(q, np.float32)
        ms = np.empty((q.shape[0], img_num), dtype=np.float32)
        _check(lib().cvtmi_opq_query_video(self.h, _ptr(q), C.c_int64(q.shape[0]), C.c_int(1 if rotate else 0),
                                           C.c_int(nprobe), C.c_int(img_num), _ptr(ms)))
        return ms

    def set_param(self, name, value):
        _check(lib().cvtmi_opq_set_param(self.h, name.encode(), C.c_int64(value)))

    def last_scan(self):
        ms = C.c_float(0); b = C.c_int64(0); qt = C.c_int(0); sp = C.c_int(0)
        _check(lib().cvtmi_opq_last_scan(self.h, C.byref(ms), C.byref(b), C.byref(qt), C.byref(sp)))
        return dict(ms=ms.value, code_bytes=b.value, qtile=qt.value, splits=sp.value)


COMM_ID_BYTES = 128
_ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class Comm:
    """cvtmi_comm_t: the communicator of the row-sharded search (one process per GPU).

    Comm.unique_id() on rank 0 -> bytes handed to the other ranks by any means -> Comm(id, rank, world) on every
    rank at the same time: RCCL (ncclCommInitRank inside libcvtmi).  Comm.over_torch_group(...) runs the same library
    path over a caller-supplied all-gather (torch.distributed, staged through the host): how several ranks share
    ONE GPU in tests, or what a gloo / MPI job would plug in."""

    def __init__(self, uid, rank, world):
        self.h = C.c_void_p(0)
        self._cb = None
        buf = None
        if uid is not None:
            assert len(uid) == COMM_ID_BYTES
            buf = (C.c_char * COMM_ID_BYTES).from_buffer_copy(bytes(uid))
        _check(lib().cvtmi_comm_create(buf, C.c_int(rank), C.c_int(world), C.byref(self.h)))
        self.rank, self.world = rank, world

    @staticmethod
    def unique_id():
        buf = (C.c_char * COMM_ID_BYTES)()
        _check(lib().cvtmi_comm_unique_id(buf))
        return bytes(buf.raw)

    @classmethod
    def create_all(cls, ndev, devices=None):
        """ONE process driving `ndev` GPUs: ncclCommInitAll inside the library; returns the list of per-device communicators."""
        hs = (C.c_void_p * ndev)()
        dv = None if devices is None else (C.c_int * ndev)(*devices)
        _check(lib().cvtmi_comm_create_all(C.c_int(ndev), dv, hs))
        out = []
        for d in range(ndev):
            self = cls.__new__(cls)
            self.h = C.c_void_p(hs[d]); self._cb = None; self.rank, self.world = d, ndev
            out.append(self)
        return out

    @classmethod
    def custom(cls, fn, rank, world):
        """fn(send_ptr, recv_ptr, nbytes, stream_ptr) -> 0: gather nbytes from every rank into recv (device pointers)."""
        self = cls.__new__(cls)
        self.h = C.c_void_p(0)
        self._cb_error = None

        def guarded(ctx, s, r, nb, st):
            # an exception must not escape a ctypes callback: ctypes would print it and return 0, and the library would merge
            # whatever the gather buffer held (stale lists included) as if the all-gather had happened
            try:
                return int(fn(s, r, nb, st) or 0)
            except BaseException as e:  # noqa: B902 -- KeyboardInterrupt included: re-raised by the caller below
                self._cb_error = e
                return -1

        self._cb = _ALLGATHER_FN(guarded)
        _check(lib().cvtmi_comm_create_custom(self._cb, None, C.c_int(rank), C.c_int(world), C.byref(self.h)))
        self.rank, self.world = rank, world
        return self

    def check(self, rc):
        """_check for calls that may run a caller-supplied all-gather: when that callback raised, the call failed with
        CVTMI_ECOMM -- the callback's own exception is what the caller gets (chained to the library's error)"""
        try:
            _check(rc)
        except CvtmiError as err:
            e, self._cb_error = getattr(self, "_cb_error", None), None
            if e is not None:
                raise e from err
            raise

    def status(self):
        """cvtmi_comm_status: raises CvtmiError (CVTMI_ECOMM) once if a search since the last report failed on some rank under the deferred
        status check (its results were voided: +inf / -1).  Does not synchronise -- synchronise the searches' stream first."""
        _check(lib().cvtmi_comm_status(self.h))

    @classmethod
    def over_torch_group(cls, rank, world, group=None):
        """All-gather through torch.distributed on host buffers (any backend): D2H of this rank's slot, all_gather,
        H2D of everybody's.  The library side (slot layout, merge kernel) is the one the RCCL transport uses."""
        import torch
        import torch.distributed as dist

        def fn(send, recv, nbytes, stream):
            st = torch.cuda.current_stream()
            assert (stream or 0) == st.cuda_stream, "library work was enqueued on another stream than torch's current one"
            st.synchronize()
            hip = C.CDLL("libamdhip64.so")
            mine = torch.empty(nbytes, dtype=torch.uint8)
            if hip.hipMemcpy(C.c_void_p(mine.data_ptr()), C.c_void_p(send), C.c_size_t(nbytes), C.c_int(2)) != 0:
                return 1
            allb = torch.empty(nbytes * world, dtype=torch.uint8)
            dist.all_gather_into_tensor(allb, mine, group=group)
            if hip.hipMemcpy(C.c_void_p(recv), C.c_void_p(allb.data_ptr()), C.c_size_t(nbytes * world), C.c_int(1)) != 0:
                return 1
            return 0

        return cls.custom(fn, rank, world)

    @classmethod
    def over_rendezvous(cls, rv):
        """All-gather staged through the host over a cvt_amd.rendezvous.Rendezvous (plain TCP, no torch.distributed): the
        debug transport that lets several ranks share ONE GPU (bench.py --backend host)."""
        import torch

        def fn(send, recv, nbytes, stream):
            st = torch.cuda.current_stream()
            assert (stream or 0) == st.cuda_stream, "library work was enqueued on another stream than torch's current one"
            st.synchronize()
            hip = C.CDLL("libamdhip64.so")
            mine = (C.c_char * nbytes)()
            if hip.hipMemcpy(mine, C.c_void_p(send), C.c_size_t(nbytes), C.c_int(2)) != 0:
                return 1
            allb = rv.allgather_bytes(mine.raw)
            if len(allb) != nbytes * rv.world:
                return 2
            if hip.hipMemcpy(C.c_void_p(recv), allb, C.c_size_t(len(allb)), C.c_int(1)) != 0:
                return 1
            return 0

        return cls.custom(fn, rv.rank, rv.world)

    def info(self):
        r, w, t = C.c_int(0), C.c_int(0), C.c_int(0)
        n, b = C.c_int64(0), C.c_int64(0)
        _check(lib().cvtmi_comm_info(self.h, C.byref(r), C.byref(w), C.byref(t), C.byref(n), C.byref(b)))
        return {"rank": r.value, "world": w.value, "transport": ("none", "rccl", "custom")[t.value],
                "collectives": n.value, "bytes_per_rank": b.value}

    def merge_topk(self, d, i, k):
        """The exchange step alone for per-shard lists (torch device tensors [nq][k], global ids)."""
        import torch
        nq = d.shape[0]
        od = torch.empty((nq, k), dtype=torch.float32, device=d.device)
        oi = torch.empty((nq, k), dtype=torch.int64, device=d.device)
        self.check(lib().cvtmi_shard_merge_topk_dev(self.h, _ptr(d.contiguous()), _ptr(i.contiguous()), C.c_int64(nq), C.c_int(k),
                                                _ptr(od), _ptr(oi), _stream()))
        return od, oi

    def close(self):
        """Destroys the communicator.  Under the deferred status check (the library default) a rank's failed search voids the
        results (+inf / -1) and leaves the error ON the communicator until the next call or status() collects it: an error nobody
        collected must not vanish with the handle, so close() raises it (after the handle is gone)."""
        if self.h and self.h.value:
            rc = lib().cvtmi_comm_status(self.h)
            msg = lib().cvtmi_last_error().decode(errors="replace") if rc != 0 else ""
            lib().cvtmi_comm_destroy(self.h)
            self.h = C.c_void_p(0)
            if rc != 0:
                raise CvtmiError("cvtmi error %d (uncollected when the communicator was closed): %s" % (rc, msg))

    def __del__(self):
        try:
            self.close()
        except CvtmiError as e:   # nowhere to raise from a finaliser: say it
            import sys
            print("cvt_amd.Comm: %s" % e, file=sys.stderr)
        except Exception:
            pass


def search_sharded_all(indexes, comms, q, k, rotate=True):
    """Single-process multi-GPU search: indexes[d] (OpqIndex or FlatIndex with its id base set) holds the row block of device d,
    comms = Comm.create_all(len(indexes)); q, results: host arrays."""
    nd = len(indexes)
    hs = (C.c_void_p * nd)(*[ix.h.value for ix in indexes])
    cs = (C.c_void_p * nd)(*[c.h.value for c in comms])
    nq = q.shape[0]
    i = np.empty((nq, k), dtype=np.int64)
    if isinstance(indexes[0], FlatIndex):
        q = _np(q, indexes[0]._dt())
        d = np.empty((nq, k), dtype=np.int32 if indexes[0].metric == 2 else np.float32)
        _check(lib().cvtmi_flat_search_sharded_all(hs, cs, C.c_int(nd), _ptr(q), C.c_int64(nq), C.c_int(k), _ptr(d), _ptr(i)))
    else:
        q = _np(q, np.float32)
        d = np.empty((nq, k), dtype=np.float32)
        _check(lib().cvtmi_opq_search_sharded_all(hs, cs, C.c_int(nd), _ptr(q), C.c_int64(nq), C.c_int(1 if rotate else 0), C.c_int(k),
                                                  _ptr(d), _ptr(i)))
    return d, i


def shard_range(n_total, rank, world):
    b, e = C.c_int64(0), C.c_int64(0)
    _check(lib().cvtmi_shard_range(C.c_int64(n_total), C.c_int(rank), C.c_int(world), C.byref(b), C.byref(e)))
    return b.value, e.value


def topk_merge(in_d, in_i, k):
    """[nq][L][k] lists -> [nq][k]"""
    nq, L, kk = in_d.shape
    assert kk == k
    if _is_torch(in_d):
        import torch
        d = torch.empty((nq, k), dtype=torch.float32, device=in_d.device)
        i = torch.empty((nq, k), dtype=torch.int64, device=in_d.device)
        _check(lib().cvtmi_topk_merge_dev(_ptr(in_d), _ptr(in_i), C.c_int64(nq), C.c_int(L), C.c_int(k), _ptr(d),
                                          _ptr(i), _stream()))
        return d, i
    in_d = _np(in_d, np.float32); in_i = _np(in_i, np.int64)
    d = np.empty((nq, k), dtype=np.float32); i = np.empty((nq, k), dtype=np.int64)
    _check(lib().cvtmi_topk_merge(_ptr(in_d), _ptr(in_i), C.c_int64(nq), C.c_int(L), C.c_int(k), _ptr(d), _ptr(i)))
    return d, i


def topk_select(scores, k):
    """scores [nq][n] (numpy, host) -> k smallest (score, index) per row"""
    scores = _np(scores, np.float32)
    nq, n = scores.shape
    d = np.empty((nq, k), dtype=np.float32); i = np.empty((nq, k), dtype=np.int64)
    _check(lib().cvtmi_topk_select(_ptr(scores), C.c_int64(nq), C.c_int64(n), C.c_int(k), _ptr(d), _ptr(i)))
    return d, i


class FlatIndex:
    """cvtmi_flat_t: exhaustive search over fp32 (IP, L2) or uint8 (L2) rows."""

    def __init__(self, metric, D):
        self.metric, self.D = metric, D
        self.h = C.c_void_p(0)
        _check(lib().cvtmi_flat_create(C.c_int(metric), C.c_int(D), C.byref(self.h)))

    def close(self):
        if self.h and self.h.value:
            lib().cvtmi_flat_destroy(self.h)
            self.h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _dt(self):
        return np.uint8 if self.metric == 2 else np.float32

    def add(self, x, labels=None):
        if _is_torch(x):
            _check(lib().cvtmi_flat_add_dev(self.h, _ptr(x), _ptr(labels), C.c_int64(x.shape[0]), _stream()))
            return
        x = _np(x, self._dt())
        lab = None if labels is None else _np(labels, np.int64)
        _check(lib().cvtmi_flat_add(self.h, _ptr(x), _ptr(lab), C.c_int64(x.shape[0])))

    @property
    def ntotal(self):
        n = C.c_int64(0)
        _check(lib().cvtmi_flat_ntotal(self.h, C.byref(n)))
        return n.value

    def search(self, q, k):
        nq = q.shape[0]
        if _is_torch(q):
            import torch
            d = torch.empty((nq, k), dtype=torch.int32 if self.metric == 2 else torch.float32, device=q.device)
            i = torch.empty((nq, k), dtype=torch.int64, device=q.device)
            _check(lib().cvtmi_flat_search_dev(self.h, _ptr(q), C.c_int64(nq), C.c_int(k), _ptr(d), _ptr(i), _stream()))
            return d, i
        q = _np(q, self._dt())
        d = np.empty((nq, k), dtype=np.int32 if self.metric == 2 else np.float32)
        i = np.empty((nq, k), dtype=np.int64)
        _check(lib().cvtmi_flat_search(self.h, _ptr(q), C.c_int64(nq), C.c_int(k), _ptr(d), _ptr(i)))
        return d, i

    def set_id_base(self, base):
        _check(lib().cvtmi_flat_set_id_base(self.h, C.c_int64(base)))

    def search_sharded(self, comm, q, k):
        """Row-sharded exhaustive search: this handle holds the rank's row block (set_id_base = its first row)."""
        nq = q.shape[0]
        if _is_torch(q):
            import torch
            d = torch.empty((nq, k), dtype=torch.int32 if self.metric == 2 else torch.float32, device=q.device)
            i = torch.empty((nq, k), dtype=torch.int64, device=q.device)
            comm.check(lib().cvtmi_flat_search_sharded_dev(self.h, comm.h, _ptr(q), C.c_int64(nq), C.c_int(k), _ptr(d), _ptr(i), _stream()))
            return d, i
        q = _np(q, self._dt())
        d = np.empty((nq, k), dtype=np.int32 if self.metric == 2 else np.float32)
        i = np.empty((nq, k), dtype=np.int64)
        comm.check(lib().cvtmi_flat_search_sharded(self.h, comm.h, _ptr(q), C.c_int64(nq), C.c_int(k), _ptr(d), _ptr(i)))
        return d, i

    def last_search(self):
        """(how the last search was answered: 0 exact / streaming kernels, 1 sample + matrix-core filter, 2 fp32 stream, 3 fp32 threshold filter, 4 uint8
        threshold filter; largest candidate list (0 for the threshold filters: nothing of theirs is read back))"""
        f = C.c_int(0); m = C.c_int64(0)
        _check(lib().cvtmi_flat_last_search(self.h, C.byref(f), C.byref(m)))
        return f.value, m.value


class HnswIndex:
    """cvtmi_hnsw_*: batched search over a graph file written by the reference's HierarchicalNSW::saveIndex."""
    def __init__(self, index_bytes, metric, D):
        self.h = C.c_void_p()
        self.D = D
        buf = np.frombuffer(index_bytes, dtype=np.uint8)
        _check(lib().cvtmi_hnsw_load(_ptr(buf), C.c_int64(buf.size), C.c_int(metric), C.c_int(D), C.byref(self.h)))

    def close(self):
        if self.h:
            lib().cvtmi_hnsw_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def ntotal(self):
        lib().cvtmi_hnsw_ntotal.restype = C.c_int64
        return int(lib().cvtmi_hnsw_ntotal(self.h))

    def search(self, q, k, ef):
        nq = q.shape[0]
        if _is_torch(q):
            import torch
            assert q.is_contiguous() and q.dtype == torch.float32
            d = torch.empty((nq, k), dtype=torch.float32, device=q.device)
            lab = torch.empty((nq, k), dtype=torch.int64, device=q.device)
            _check(lib().cvtmi_hnsw_search_dev(self.h, _ptr(q), C.c_int64(nq), C.c_int(k), C.c_int(ef), _ptr(d), _ptr(lab), _stream()))
            return d, lab
        q = _np(q, np.float32)
        d = np.empty((nq, k), dtype=np.float32); lab = np.empty((nq, k), dtype=np.int64)
        _check(lib().cvtmi_hnsw_search(self.h, _ptr(q), C.c_int64(nq), C.c_int(k), C.c_int(ef), _ptr(d), _ptr(lab)))
        return d, lab

    def search_adc(self, opq, q, k, ef, rotate=True):
        """Same graph, distances = ADC over the PQ codes held by the OpqIndex `opq` (one row per node)."""
        nq = q.shape[0]
        if _is_torch(q):
            import torch
            assert q.is_contiguous() and q.dtype == torch.float32
            d = torch.empty((nq, k), dtype=torch.float32, device=q.device)
            lab = torch.empty((nq, k), dtype=torch.int64, device=q.device)
            _check(lib().cvtmi_hnsw_search_adc_dev(self.h, opq.h, _ptr(q), C.c_int64(nq), C.c_int(int(rotate)), C.c_int(k), C.c_int(ef),
                                                   _ptr(d), _ptr(lab), _stream()))
            return d, lab
        q = _np(q, np.float32)
        d = np.empty((nq, k), dtype=np.float32); lab = np.empty((nq, k), dtype=np.int64)
        _check(lib().cvtmi_hnsw_search_adc(self.h, opq.h, _ptr(q), C.c_int64(nq), C.c_int(int(rotate)), C.c_int(k), C.c_int(ef),
                                           _ptr(d), _ptr(lab)))
        return d, lab


def _hnsw_search_adc_rerank(self, opq, q, k, ef, rerank=None, rotate=True):
    """ADC traversal + exact fp32 re-rank of its `rerank` best nodes (default: ef)."""
    rerank = ef if rerank is None else rerank
    nq = q.shape[0]
    if _is_torch(q):
        import torch
        assert q.is_contiguous() and q.dtype == torch.float32
        d = torch.empty((nq, k), dtype=torch.float32, device=q.device)
        lab = torch.empty((nq, k), dtype=torch.int64, device=q.device)
        _check(lib().cvtmi_hnsw_search_adc_rerank_dev(self.h, opq.h, _ptr(q), C.c_int64(nq), C.c_int(int(rotate)), C.c_int(k), C.c_int(ef),
                                                      C.c_int(rerank), _ptr(d), _ptr(lab), _stream()))
        return d, lab
    q = _np(q, np.float32)
    d = np.empty((nq, k), dtype=np.float32); lab = np.empty((nq, k), dtype=np.int64)
    _check(lib().cvtmi_hnsw_search_adc_rerank(self.h, opq.h, _ptr(q), C.c_int64(nq), C.c_int(int(rotate)), C.c_int(k), C.c_int(ef),
                                              C.c_int(rerank), _ptr(d), _ptr(lab)))
    return d, lab


HnswIndex.search_adc_rerank = _hnsw_search_adc_rerank


class _PinnedOwner:
    """a cvtmi_host_alloc block; freed when the last numpy view of it goes away (pinned_empty ties the two together)"""

    def __init__(self, nbytes):
        self.p = C.c_void_p(0)
        _check(lib().cvtmi_host_alloc(C.c_size_t(nbytes), C.byref(self.p)))
        self.nbytes = nbytes

    def __del__(self):
        try:
            if self.p and self.p.value:
                lib().cvtmi_host_free(self.p)
                self.p = C.c_void_p(0)
        except Exception:
            pass


def pinned_empty(shape, dtype):
    """numpy array in page-locked host memory (cvtmi_host_alloc): the host-pointer entries move such arrays without a staging
    copy.  The block lives exactly as long as an array views it: the ctypes buffer every view keeps as its base carries the
    owner, so the last view's death frees the page-locked block (no module-level table, nothing leaks)."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    own = _PinnedOwner(max(n, 16))
    buf = (C.c_char * max(n, 16)).from_address(own.p.value)
    buf._cvtmi_owner = own   # numpy keeps `buf` alive as the base of the array and of every view / reshape of it
    arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    arr.flags.writeable = True
    return arr


def set_tuning(name, value):
    """library-wide tuning / measurement hooks (cvtmi_set_tuning): no effect on results"""
    _check(lib().cvtmi_set_tuning(name.encode(), C.c_int64(int(value))))


def kmeans(x, k, niter=0, seed=1):
    """cvtmi_kmeans: (centroids [k][d], assign [n], iterations)."""
    n, d = x.shape
    it = C.c_int(0)
    if _is_torch(x):
        import torch
        assert x.is_contiguous() and x.dtype == torch.float32
        cent = torch.empty((k, d), dtype=torch.float32, device=x.device)
        assign = torch.empty((n,), dtype=torch.int32, device=x.device)
        _check(lib().cvtmi_kmeans_dev(_ptr(x), C.c_int64(d), C.c_int64(n), C.c_int(d), C.c_int(k), C.c_int(niter),
                                      C.c_uint64(seed), _ptr(cent), _ptr(assign), C.byref(it), _stream()))
        return cent, assign, it.value
    x = _np(x, np.float32)
    cent = np.empty((k, d), dtype=np.float32); assign = np.empty(n, dtype=np.int32)
    _check(lib().cvtmi_kmeans(_ptr(x), C.c_int64(n), C.c_int(d), C.c_int(k), C.c_int(niter), C.c_uint64(seed),
                              _ptr(cent), _ptr(assign), C.byref(it)))
    return cent, assign, it.value


def opq_train(x, coarseK, M, K, niter=0, seed=1):
    """cvtmi_opq_train on already permuted / rotated rows: (coarse [coarseK][D], books [M][K][D/M])."""
    n, D = x.shape
    if _is_torch(x):
        import torch
        assert x.is_contiguous() and x.dtype == torch.float32
        coarse = torch.empty((coarseK, D), dtype=torch.float32, device=x.device)
        books = torch.empty((M, K, D // M), dtype=torch.float32, device=x.device)
        _check(lib().cvtmi_opq_train_dev(_ptr(x), C.c_int64(n), C.c_int(D), C.c_int(coarseK), C.c_int(M), C.c_int(K),
                                         C.c_int(niter), C.c_uint64(seed), _ptr(coarse), _ptr(books), _stream()))
        return coarse, books
    x = _np(x, np.float32)
    coarse = np.empty((coarseK, D), dtype=np.float32); books = np.empty((M, K, D // M), dtype=np.float32)
    _check(lib().cvtmi_opq_train(_ptr(x), C.c_int64(n), C.c_int(D), C.c_int(coarseK), C.c_int(M), C.c_int(K), C.c_int(niter),
                                 C.c_uint64(seed), _ptr(coarse), _ptr(books)))
    return coarse, books


def opq_learn_rotation(x, M, K, outer, niter=0, seed=1):
    """cvtmi_opq_learn_rotation: (R [D][D], books [M][K][D/M]) learned from the rows x (numpy -> numpy, torch CUDA -> torch)"""
    n, D = x.shape
    if _is_torch(x):
        import torch
        R = torch.empty((D, D), dtype=torch.float32, device=x.device)
        books = torch.empty((M, K, D // M), dtype=torch.float32, device=x.device)
        _check(lib().cvtmi_opq_learn_rotation_dev(_ptr(x), C.c_int64(n), C.c_int(D), C.c_int(M), C.c_int(K), C.c_int(outer), C.c_int(niter),
                                                  C.c_uint64(seed), _ptr(R), _ptr(books), _stream()))
        return R, books
    x = _np(x, np.float32)
    R = np.empty((D, D), dtype=np.float32); books = np.empty((M, K, D // M), dtype=np.float32)
    _check(lib().cvtmi_opq_learn_rotation(_ptr(x), C.c_int64(n), C.c_int(D), C.c_int(M), C.c_int(K), C.c_int(outer), C.c_int(niter),
                                          C.c_uint64(seed), _ptr(R), _ptr(books)))
    return R, books


def sq8_train(x, l2norm=True):
    n, d = x.shape
    if _is_torch(x):
        import torch
        vmin = torch.empty(d, dtype=torch.float32, device=x.device); vdiff = torch.empty_like(vmin)
        _check(lib().cvtmi_sq8_train_dev(_ptr(x), C.c_int64(n), C.c_int(d), C.c_int(int(l2norm)), _ptr(vmin), _ptr(vdiff),
                                         _stream()))
        return vmin, vdiff
    x = _np(x, np.float32)
    vmin = np.empty(d, dtype=np.float32); vdiff = np.empty(d, dtype=np.float32)
    _check(lib().cvtmi_sq8_train(_ptr(x), C.c_int64(n), C.c_int(d), C.c_int(int(l2norm)), _ptr(vmin), _ptr(vdiff)))
    return vmin, vdiff


def sq8_encode(vmin, vdiff, x, l2norm=True):
    """Returns codes; x is normalised IN PLACE when l2norm is True / 1 (reference behaviour), left alone when l2norm == 2."""
    n, d = x.shape
    if _is_torch(x):
        import torch
        codes = torch.empty((n, d), dtype=torch.uint8, device=x.device)
        _check(lib().cvtmi_sq8_encode_dev(_ptr(vmin), _ptr(vdiff), C.c_int(d), _ptr(x), C.c_int64(n), C.c_int(int(l2norm)),
                                          _ptr(codes), _stream()))
        return codes
    assert x.dtype == np.float32 and x.flags["C_CONTIGUOUS"]
    vmin = _np(vmin, np.float32); vdiff = _np(vdiff, np.float32)
    codes = np.empty((n, d), dtype=np.uint8)
    _check(lib().cvtmi_sq8_encode(_ptr(vmin), _ptr(vdiff), C.c_int(d), _ptr(x), C.c_int64(n), C.c_int(int(l2norm)),
                                  _ptr(codes)))
    return codes


def pca_project(mean, vectors, x, l2norm=True):
    """cvtk::PCAUtils::reduceDim (pca_utils.cc:25-35): (x - mean) * vectors^T, rows L2-normalised.  numpy in ->
    numpy out (host entry); torch CUDA tensors in -> torch tensor out (device entry, current stream)."""
    n, din = x.shape
    dout = vectors.shape[0]
    if _is_torch(x):
        import torch
        y = torch.empty((n, dout), dtype=torch.float32, device=x.device)
        _check(lib().cvtmi_pca_project_dev(_ptr(mean), _ptr(vectors), C.c_int(din), C.c_int(dout), _ptr(x), C.c_int64(n),
                                           C.c_int(1 if l2norm else 0), _ptr(y), _stream()))
        return y
    x = _np(x, np.float32); vectors = _np(vectors, np.float32); mean = _np(mean, np.float32)
    y = np.empty((n, dout), dtype=np.float32)
    _check(lib().cvtmi_pca_project(_ptr(mean), _ptr(vectors), C.c_int(din), C.c_int(dout), _ptr(x), C.c_int64(n),
                                   C.c_int(1 if l2norm else 0), _ptr(y)))
    return y


def sq8_decode(vmin, vdiff, codes):
    n, d = codes.shape
    if _is_torch(codes):
        import torch
        x = torch.empty((n, d), dtype=torch.float32, device=codes.device)
        _check(lib().cvtmi_sq8_decode_dev(_ptr(vmin), _ptr(vdiff), C.c_int(d), _ptr(codes), C.c_int64(n), _ptr(x), _stream()))
        return x
    codes = _np(codes, np.uint8); vmin = _np(vmin, np.float32); vdiff = _np(vdiff, np.float32)
    x = np.empty((n, d), dtype=np.float32)
    _check(lib().cvtmi_sq8_decode(_ptr(vmin), _ptr(vdiff), C.c_int(d), _ptr(codes), C.c_int64(n), _ptr(x)))
    return x


def sq8_decode_faiss(vmin, vdiff, codes):
    """Int8Decode(uint8_t*) / Int8DecodeFaiss arithmetic: the fp32 8-bit codec of faiss (cvtmi_sq8_decode_faiss)"""
    n, d = codes.shape
    if _is_torch(codes):
        import torch
        x = torch.empty((n, d), dtype=torch.float32, device=codes.device)
        _check(lib().cvtmi_sq8_decode_faiss_dev(_ptr(vmin), _ptr(vdiff), C.c_int(d), _ptr(codes), C.c_int64(n), _ptr(x), _stream()))
        return x
    codes = _np(codes, np.uint8); vmin = _np(vmin, np.float32); vdiff = _np(vdiff, np.float32)
    x = np.empty((n, d), dtype=np.float32)
    _check(lib().cvtmi_sq8_decode_faiss(_ptr(vmin), _ptr(vdiff), C.c_int(d), _ptr(codes), C.c_int64(n), _ptr(x)))
    return x
