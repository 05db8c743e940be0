#!/usr/bin/env python3
"""SQ8 host-pointer entries on one vector per call (the reference's Int8Encode / Int8Decode are called per feature vector): us per call."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, cvt_amd
torch.cuda.init()
rng = np.random.default_rng(0)
for small in [int(v) for v in os.environ.get("SMALL", "0,1").split(",")]:
  cvt_amd.set_tuning("sq8_host_small", small)
  print("== sq8_host_small =", small)
  for d in (64, 512, 2048):
      xs = np.abs(rng.normal(size=(2000, d))).astype(np.float32)
      vmin, vdiff = cvt_amd.sq8_train(xs, l2norm=True)
      for n in (1, 16, 256):
          x = xs[:n].copy()
          for l2 in (True, False):
              for _ in range(5): codes = cvt_amd.sq8_encode(vmin, vdiff, x.copy(), l2norm=l2)
              reps = 200
              xc = [x.copy() for _ in range(reps)]
              t0 = time.perf_counter()
              for i in range(reps): codes = cvt_amd.sq8_encode(vmin, vdiff, xc[i], l2norm=l2)
              te = (time.perf_counter() - t0) / reps
              print("d=%d n=%d l2norm=%d: encode %.1f us per call" % (d, n, l2, te * 1e6), flush=True)
          for _ in range(5): dec = cvt_amd.sq8_decode(vmin, vdiff, codes)
          t0 = time.perf_counter()
          for i in range(200): dec = cvt_amd.sq8_decode(vmin, vdiff, codes)
          print("d=%d n=%d: decode %.1f us per call" % (d, n, (time.perf_counter() - t0) / 200 * 1e6), flush=True)
