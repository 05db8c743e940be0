#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity sweep over the main entry points (development aid; the fixed cases live in
tests/).  Exits non-zero on the first mismatch and prints the configuration."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch, cvt_amd
from oracle import binding as ob
ob.build()
orc = ob.Oracle()
bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
seed = int(os.environ.get("SEED", 1)); iters = int(os.environ.get("ITERS", 40))
rng = np.random.default_rng(seed)
for it in range(iters):
    M = int(rng.choice([16, 16, 16, 8, 4])); K = int(rng.choice([256, 256, 200, 64, 17])); step = int(rng.choice([8, 4, 2, 16]))
    D = M * step
    n = int(rng.integers(1, 60000)); nq = int(rng.integers(1, 90)); k = int(rng.integers(1, 129))
    if it % 3 == 1:    # the small-batch path (1 .. 32 queries, >= 65536 rows) and the big-k kernels (k > 128)
        n = int(rng.integers(65536, 300000)); nq = int(rng.integers(1, 33)); k = int(rng.choice([1, 10, 100, 128, 129, 300]))
    elif it % 3 == 2:  # enough query groups for the persistent grid to cut a group into segments
        n = int(rng.integers(20000, 200000)); nq = int(rng.integers(60, 700)); k = int(rng.choice([1, 10, 100, 128, 200]))
    scale = float(rng.choice([1.0, 1e-3, 30.0]))
    books = (rng.normal(size=(M, K, step)) * scale).astype(np.float32)
    codes = rng.integers(0, K, size=(n, M), dtype=np.uint8)
    if rng.random() < 0.3:
        codes[rng.integers(0, n, size=max(1, n // 10))] = codes[0]          # many exact ties
    q = (rng.normal(size=(nq, D)) * scale).astype(np.float32)
    if rng.random() < 0.2:
        q[0] = np.inf if rng.random() < 0.5 else np.nan
    idx = cvt_amd.OpqIndex(np.zeros((1, D), np.float32), books)
    idx.add_codes(codes)
    for variant in (7, 6, 3, 4, 5, 1, 0):
        idx.set_param("scan_variant", variant); idx.set_param("splits", int(rng.choice([0, 0, 1, 2, 5])))
        idx.set_param("prerotate", int(rng.integers(0, 2)))
        d, i = idx.search(q, k, rotate=False)
        od, oi = orc.adc_search(q, books, codes, k)
        fin = np.isfinite(od)
        if not (np.array_equal(i[fin], oi[fin]) and np.array_equal(bits(d)[fin], bits(od)[fin])):
            print("MISMATCH adc", dict(it=it, M=M, K=K, D=D, n=n, nq=nq, k=k, scale=scale, variant=variant)); sys.exit(1)
    if it % 4 == 0 and M == 16:
        # large batches (round 5): enough query groups for the scan's two-region plans -- the first region's groups store their final lists
        # in place, the merge visits the tail only -- through device pointers, pageable host arrays (pipelined pieces) and page-locked
        # result arrays (written by the kernels); the oracle checks a sample of the queries
        nL = int(rng.integers(3000, 40000)); nqL = int(rng.integers(4100, 9500)); kL = int(rng.choice([1, 10, 37, 100]))
        cL = rng.integers(0, K, size=(nL, M), dtype=np.uint8)
        cL[nL // 2] = cL[0]; cL[nL - 1] = cL[0]
        qL = (rng.normal(size=(nqL, D)) * scale).astype(np.float32)
        big = cvt_amd.OpqIndex(np.zeros((1, D), np.float32), books)
        big.add_codes(cL)
        big.set_param("scan_variant", 3)
        ref_d = ref_i = None
        for tail in (-1, 0, int(rng.choice([2, 4, 8]))):
            cvt_amd.set_tuning("scan_tail_splits", tail)
            d, i = big.search(torch.from_numpy(qL).cuda(), kL, rotate=False)
            d, i = d.cpu().numpy(), i.cpu().numpy()
            if ref_d is None:
                ref_d, ref_i = d, i
                sel = np.unique(np.r_[0:4, rng.integers(0, nqL, size=24), nqL - 4:nqL])
                od, oi = orc.adc_search(qL[sel], books, cL, kL)
                if not (np.array_equal(i[sel], oi) and np.array_equal(bits(d[sel]), bits(od))):
                    print("MISMATCH adc large", dict(it=it, K=K, D=D, n=nL, nq=nqL, k=kL, scale=scale)); sys.exit(1)
            elif not (np.array_equal(i, ref_i) and np.array_equal(bits(d), bits(ref_d))):
                print("MISMATCH adc tail plan", dict(it=it, K=K, D=D, n=nL, nq=nqL, k=kL, tail=tail)); sys.exit(1)
            for zc in (0, 1):
                cvt_amd.set_tuning("opq_host_zero_copy", zc)
                outs = [(np.zeros((nqL, kL), np.float32), np.zeros((nqL, kL), np.int64)),
                        (cvt_amd.pinned_empty((nqL, kL), np.float32), cvt_amd.pinned_empty((nqL, kL), np.int64))]
                for o in outs:
                    o[1][:] = -9
                    big.search(qL, kL, rotate=False, out=o)
                    if not (np.array_equal(o[1], ref_i) and np.array_equal(bits(o[0]), bits(ref_d))):
                        print("MISMATCH adc host pointers", dict(it=it, K=K, D=D, n=nL, nq=nqL, k=kL, tail=tail, zero_copy=zc)); sys.exit(1)
        cvt_amd.set_tuning("scan_tail_splits", 0); cvt_amd.set_tuning("opq_host_zero_copy", 1)
        big.close()
    # flat fp32 + u8
    Df = int(rng.choice([128, 64, 36, 20, 7, 512])); nf = int(rng.integers(1, 30000)); kf = int(rng.integers(1, 129)); nqf = int(rng.integers(1, 300))
    for metric in (0, 1):
        x = rng.normal(size=(nf, Df)).astype(np.float32); qq = rng.normal(size=(nqf, Df)).astype(np.float32)
        x[nf // 2] = x[0]
        fi = cvt_amd.FlatIndex(metric, Df); fi.add(x[: nf // 3 + 1]); fi.add(x[nf // 3 + 1:])
        d, i = fi.search(qq, kf)
        od, _, oi = orc.flat_search(metric, x, qq, kf)
        if not (np.array_equal(i, oi) and np.array_equal(bits(d), bits(od))):
            print("MISMATCH flat", dict(it=it, metric=metric, D=Df, n=nf, nq=nqf, k=kf)); sys.exit(1)
    Du = int(rng.choice([512, 128, 96, 32, 21])); ku = int(rng.choice([1, 5, 10, 16, 24, 40, 80, 100]))
    xu = rng.integers(0, 256, size=(nf, Du), dtype=np.uint8); qu = rng.integers(0, 256, size=(nqf, Du), dtype=np.uint8)
    xu[nf // 2] = xu[0]; qu[0] = xu[0]
    fi = cvt_amd.FlatIndex(2, Du); fi.add(xu)
    d, i = fi.search(qu, ku)
    _, odi, oi = orc.flat_search(2, xu, qu, ku)
    if not (np.array_equal(i, oi) and np.array_equal(d, odi)):
        print("MISMATCH u8", dict(it=it, D=Du, n=nf, nq=nqf, k=ku)); sys.exit(1)
    # encode (coarse + PQ), IVF query with per-video min, SQ8, k-means
    De = int(rng.choice([128, 64, 32])); Me = int(rng.choice([16, 8, 4])); Ke = int(rng.choice([256, 100])); cK = int(rng.choice([1, 7, 40]))
    ne = int(rng.integers(50, 5000)); img = int(rng.integers(1, 30))
    xe = rng.normal(size=(ne, De)).astype(np.float32)
    coarse = np.zeros((1, De), np.float32) if cK == 1 else rng.normal(size=(cK, De)).astype(np.float32)
    be = (rng.normal(size=(Me, Ke, De // Me))).astype(np.float32)
    ie = cvt_amd.OpqIndex(coarse, be)
    lists, ce = ie.encode(xe)
    ol, oc = orc.pq_encode(xe, coarse, be)
    if not (np.array_equal(ce, oc) and (cK == 1 or np.array_equal(lists, ol))):
        print("MISMATCH encode", dict(it=it, D=De, M=Me, K=Ke, coarseK=cK, n=ne)); sys.exit(1)
    vid = rng.integers(0, img, size=ne).astype(np.int32)
    ie.add_codes(ce, list_id=ol if cK > 1 else None, video_id=vid)
    qe = rng.normal(size=(int(rng.integers(1, 12)), De)).astype(np.float32) * 0.3
    nk = int(rng.integers(1, min(cK, 5) + 1))
    ms = ie.query_video(qe, nk, img, rotate=False)
    order = np.argsort(ol if cK > 1 else np.zeros(ne, np.int32), kind="stable")
    off = np.concatenate([[0], np.cumsum(np.bincount((ol if cK > 1 else np.zeros(ne, np.int64))[order], minlength=cK))]).astype(np.int64)
    oms = orc.query_video(qe, coarse, be, nk, off, oc[order], vid[order], img)
    if not np.array_equal(bits(ms), bits(oms)):
        print("MISMATCH query_video", dict(it=it, D=De, M=Me, K=Ke, coarseK=cK, n=ne, nk=nk, img=img)); sys.exit(1)
    ds = int(rng.choice([512, 64, 300, 20])); ns = int(rng.integers(1, 3000))
    xs = (rng.normal(size=(ns, ds)) * np.exp2(rng.integers(-8, 8, size=(ns, ds)))).astype(np.float32)
    l2 = bool(rng.integers(0, 2))
    vm, vd = cvt_amd.sq8_train(xs, l2norm=l2); ovm, ovd = orc.sq8_train(xs, l2norm=l2)
    xg = xs.copy(); cs = cvt_amd.sq8_encode(vm, vd, xg, l2norm=l2); ocs, ox = orc.sq8_encode(vm, vd, xs, l2norm=l2)
    if not (np.array_equal(bits(vm), bits(ovm)) and np.array_equal(bits(vd), bits(ovd)) and np.array_equal(cs, ocs) and np.array_equal(bits(xg), bits(ox))
            and np.array_equal(bits(cvt_amd.sq8_decode(vm, vd, cs)), bits(orc.sq8_decode(vm, vd, cs)))):
        print("MISMATCH sq8", dict(it=it, d=ds, n=ns, l2=l2)); sys.exit(1)
    kk = int(rng.integers(1, min(64, ne))); gc, ga, git = cvt_amd.kmeans(xe, kk, 4, it + 1); oc2, oa2, oit = orc.kmeans(xe, kk, 4, it + 1)
    if not (git == oit and np.array_equal(ga, oa2) and np.array_equal(bits(gc), bits(oc2))):
        print("MISMATCH kmeans", dict(it=it, n=ne, d=De, k=kk)); sys.exit(1)
    print("iter %d ok (adc M=%d K=%d n=%d nq=%d k=%d | flat D=%d n=%d nq=%d k=%d | u8 D=%d k=%d)" % (it, M, K, n, nq, k, Df, nf, nqf, kf, Du, ku), flush=True)
print("fuzz: %d iterations, no mismatch" % iters)
