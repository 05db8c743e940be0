"""GPU: the reference's own query semantics (IVFOPQ::Query, opq/src/IVFOPQ.cpp:213-320) at the reference's real shape:
coarseK = 8192 lists, nk = 3 probes, >= 1 M entries -- device-built list-ordered copy (stable counting sort), coarse top-nk,
per-list tables and scans -- against the oracle, and at a smaller size against the reference itself run live
(oracle/_ref/libref_opq.so = the reference's IVFOPQ.cpp compiled in place)."""
import os
import time

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import cvt_amd
    return cvt_amd


def _host_csr(lists, videos, codes, L):
    keep = (lists >= 0) & (lists < L)
    order = np.argsort(lists[keep], kind="stable")
    off = np.zeros(L + 1, np.int64)
    np.add.at(off, lists[keep] + 1, 1)
    return np.cumsum(off), videos[keep][order], codes[keep][order]


def test_ivf_query_reference_shape(amd, orc):
    import torch
    D, M, K, L, nk = 128, 16, 256, 8192, 3
    n, n_videos, nq = 1_000_000, 5000, 300
    rng = np.random.default_rng(8192)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(5)
    # scales chosen so that residuals are ~0.03 per dimension: ADC scores of true neighbours sit well below the 1.0 clamp
    cen = torch.randn((L, D), generator=g, device=dev) * 0.08
    x = cen[torch.randint(0, L, (n,), generator=g, device=dev)] + 0.03 * torch.randn((n, D), generator=g, device=dev)
    x[1000:1040] = x[1000]                                   # duplicates: equal scores inside a list
    coarse = cen.cpu().numpy()
    books = (rng.normal(size=(M, K, D // M)) * 0.03).astype(np.float32)
    perm = rng.permutation(D).astype(np.int32)
    idx = amd.OpqIndex(coarse, books, perm=perm)
    xr = idx.rotate(x)
    lists, codes = idx.encode(xr)
    videos = torch.from_numpy(rng.integers(0, n_videos, size=n).astype(np.int32)).to(dev)
    half = n // 2 + 17
    idx.add_codes(codes[:half], lists[:half], videos[:half])  # two appends: the list-ordered copy is rebuilt after each
    q = x[torch.randint(0, n, (nq,), generator=g, device=dev)] + 0.01 * torch.randn((nq, D), generator=g, device=dev)
    ms_half = idx.query_video(q, nk, n_videos, rotate=True)
    idx.add_codes(codes[half:], lists[half:], videos[half:])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ms = idx.query_video(q, nk, n_videos, rotate=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    lists_h, codes_h, vid_h = lists.cpu().numpy(), codes.cpu().numpy(), videos.cpu().numpy()
    # coarse assignment of a sample against the oracle (the full 1 M x 8192 x 128 chain is minutes of CPU)
    xr_h = xr.cpu().numpy()
    samp = rng.integers(0, n, size=1500)
    assert np.array_equal(orc.coarse_assign(xr_h[samp], coarse), lists_h[samp])
    _, codes_o = orc.pq_encode(xr_h[samp], coarse, books)
    assert np.array_equal(codes_o, codes_h[samp])
    # the device-built list-ordered copy == a stable host sort (SaveIndex order)
    off, vid_csr, codes_csr = idx.get_entries()
    off_h, vid_e, codes_e = _host_csr(lists_h, vid_h, codes_h, L)
    assert np.array_equal(off, off_h) and np.array_equal(vid_csr, vid_e) and np.array_equal(codes_csr, codes_e)
    # scores: bit-exact against the oracle's Query
    qr = orc.reorder(perm, q.cpu().numpy())
    oms = orc.query_video(qr, coarse, books, nk, off_h, codes_e, vid_e, n_videos)
    assert np.array_equal(bits(ms.cpu().numpy()), bits(oms))
    off2, vid2, codes2 = _host_csr(lists_h[:half], vid_h[:half], codes_h[:half], L)
    oms2 = orc.query_video(qr, coarse, books, nk, off2, codes2, vid2, n_videos)
    assert np.array_equal(bits(ms_half.cpu().numpy()), bits(oms2))
    assert (oms < 1.0).sum() > nq                             # the probes do find entries below the 1.0 clamp
    # coarse probing through the exact kernels and through the matrix-core filter: same scores
    for variant in (1, 2):
        amd.set_tuning("probe_variant", variant)
        msv = idx.query_video(q, nk, n_videos, rotate=True)
        assert np.array_equal(bits(msv.cpu().numpy()), bits(oms)), variant
        for nkk in (1, 8, 40):                                     # other probe counts, incl. more than the filter's candidate list takes
            a = idx.query_video(q[:260].contiguous(), nkk, n_videos, rotate=True)
            amd.set_tuning("probe_variant", 1)
            b = idx.query_video(q[:260].contiguous(), nkk, n_videos, rotate=True)
            amd.set_tuning("probe_variant", variant)
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (variant, nkk)
    amd.set_tuning("probe_variant", 0)
    # the library's own choice below 1536 frames: the centroids cut into ranges over all CUs, one merge per frame (coarse_probe_split_kernel)
    # -- batch sizes on both sides of the old kernels' switch points, probe counts from 1 to more than a range's share
    for nf in (1, 9, 64, 255, 300):
        for nkk in (1, 3, 40):
            a = idx.query_video(q[:nf].contiguous(), nkk, n_videos, rotate=True)
            amd.set_tuning("probe_variant", 1)
            b = idx.query_video(q[:nf].contiguous(), nkk, n_videos, rotate=True)
            amd.set_tuning("probe_variant", 0)
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (nf, nkk)
    # the host-pointer entry gives the same
    msh = idx.query_video(q[:7].cpu().numpy(), nk, n_videos, rotate=True)
    assert np.array_equal(bits(msh), bits(oms[:7]))
    # Query only reads the index (IVFOPQ.cpp:213-320): four host threads query one handle at once -- host arrays in and out, each call on a
    # scratch set leased from the handle -- and every call returns what it returns alone
    import threading
    qh = q.cpu().numpy()
    jobs = [(0, 9), (9, 18), (100, 164), (200, 201)]
    want = {j: idx.query_video(qh[j[0]:j[1]], nk, n_videos, rotate=True) for j in jobs}
    bad, go = [], threading.Barrier(len(jobs))

    def worker(j):
        go.wait()
        for _ in range(30):
            if not np.array_equal(bits(idx.query_video(qh[j[0]:j[1]], nk, n_videos, rotate=True)), bits(want[j])):
                bad.append(j)
                return
    ts = [threading.Thread(target=worker, args=(j,)) for j in jobs]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not bad, bad
    for j in jobs:
        assert np.array_equal(bits(want[j]), bits(oms[j[0]:j[1]]))
    tq0 = time.perf_counter()
    for _ in range(200):
        idx.query_video(qh[:9], nk, n_videos, rotate=True)
    print("ivf query, host pointers, 9 frames: %.1f us per call" % ((time.perf_counter() - tq0) / 200 * 1e6))
    print("ivf query: %d frames x nk=%d over %d entries in %d lists: %.3f ms" % (nq, nk, n, L, (t1 - t0) * 1e3))
    idx.close()


def test_ivf_query_against_reference_live(amd, orc):
    """The same path against the reference itself (its own Add / Query loops, compiled in place) at a size its single
    thread finishes in seconds: coarseK = 512, 24 K entries in 40 videos, nk = 3, skewed list sizes."""
    from oracle import binding as ob
    if not ob.ref_available():
        pytest.skip("oracle/_ref not built")
    D, M, K, L, nk = 128, 16, 256, 512, 3
    rng = np.random.default_rng(99)
    coarse = (rng.normal(size=(L, D)) * 0.08).astype(np.float32)
    books = (rng.normal(size=(M, K, D // M)) * 0.03).astype(np.float32)
    perm = rng.permutation(D).astype(np.int32)
    vids = []
    for v in range(40):
        nv = int(rng.integers(200, 1000))
        c = coarse[rng.integers(0, 8 if v % 3 == 0 else L, size=nv)]   # every third video crowds 8 lists: long lists
        vids.append((c[:, np.argsort(perm)] + 0.03 * rng.normal(size=(nv, D))).astype(np.float32))
    ref = ob.RefOPQ(coarse, books, perm)
    assert ref.index(vids) == len(vids)
    q = np.concatenate([v[:6] for v in vids[:20]]).astype(np.float32) + (0.01 * rng.normal(size=(120, D))).astype(np.float32)
    rms = ref.query(q, nk, len(vids))
    r_off, r_vid, r_codes = ref.dump()
    idx = amd.OpqIndex(coarse, books, perm=perm)
    for v, x in enumerate(vids):
        lists, codes = idx.encode(idx.rotate(x))
        idx.add_codes(codes, lists, np.full(x.shape[0], v, np.int32))
    off, vid, codes = idx.get_entries()
    assert np.array_equal(off, r_off) and np.array_equal(vid, r_vid) and np.array_equal(codes, r_codes)
    ms = idx.query_video(q, nk, len(vids), rotate=True)
    assert np.array_equal(bits(ms), bits(rms))
    assert (rms < 1.0).sum() > 120                            # scores below the clamp do occur
    import torch
    msd = idx.query_video(torch.from_numpy(q).cuda(), nk, len(vids), rotate=True)   # 120 frames: the batched coarse kernel
    assert np.array_equal(bits(msd.cpu().numpy()), bits(rms))
    ref.close(); idx.close()
