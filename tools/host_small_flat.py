#!/usr/bin/env python3
"""small host-pointer flat searches (the brute_force CLI's call pattern: one searchKnn per query): wall time per call, host pointers
against device pointers + synchronise.  ROWS / D / K / NQS env; TUNE=name=value,..."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cvt_amd
dev = torch.device("cuda", 0)
rows, D, k = int(os.environ.get("ROWS", 1_000_000)), int(os.environ.get("D", 128)), int(os.environ.get("K", 100))
g = torch.Generator(device=dev); g.manual_seed(3)
x = torch.randn((rows, D), generator=g, device=dev)
for metric in (0, 1):
    fi = cvt_amd.FlatIndex(metric, D)
    fi.add(x)
    for nq in [int(v) for v in os.environ.get("NQS", "1,8,64").split(",")]:
        qd = torch.randn((nq, D), generator=g, device=dev)
        qh = qd.cpu().numpy()
        for _ in range(20):
            d0, i0 = fi.search(qd, k)
        torch.cuda.synchronize()
        reps = 300
        t0 = time.perf_counter()
        for _ in range(reps):
            d0, i0 = fi.search(qd, k); torch.cuda.synchronize()
        t_dev = (time.perf_counter() - t0) / reps
        for tune in os.environ.get("TUNES", "flat_small_zero_copy=0;flat_small_zero_copy=1").split(";"):
            for kv in [v for v in tune.split(",") if v]:
                name, value = kv.split("=")
                try:
                    cvt_amd.set_tuning(name, float(value))
                except Exception as e:
                    print("(", e, ")")
            for _ in range(20):
                dh, ih = fi.search(qh, k)
            same = bool(np.array_equal(ih, i0.cpu().numpy()) and np.array_equal(dh.view(np.uint32), d0.cpu().numpy().view(np.uint32)))
            t0 = time.perf_counter()
            for _ in range(reps):
                fi.search(qh, k)
            t_host = (time.perf_counter() - t0) / reps
            print("metric=%d nq=%d: device pointers %.1f us, host pointers %.1f us (%s), same=%s" % (metric, nq, t_dev * 1e6, t_host * 1e6, tune, same), flush=True)
    fi.close()
