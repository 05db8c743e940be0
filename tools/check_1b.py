#!/usr/bin/env python3
"""One-off: 2^30 code rows on one GPU, rows planted across the whole range must come back first (id order)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, cvt_amd
from oracle import binding as ob
orc = ob.Oracle()
D, M, K, n = 128, 16, 256, 1 << 30
rng = np.random.default_rng(5)
books = (rng.normal(size=(M, K, D // M)) * 0.1).astype(np.float32)
q = (rng.normal(size=(16, D)) * 0.1).astype(np.float32)
idx = cvt_amd.OpqIndex(np.zeros((1, D), np.float32), books)
idx.reserve(n)
g = torch.Generator(device="cuda"); g.manual_seed(9)
planted = [7, (1 << 28) - 1, (1 << 28), (1 << 29) + 12345, 3 * (1 << 28) + 99, n - 1]
best = orc.lut(q[0], np.zeros(D, np.float32), books).argmin(axis=1).astype(np.uint8)
done = 0
while done < n:
    m = min(1 << 25, n - done)
    chunk = torch.randint(0, 256, (m, M), generator=g, device="cuda", dtype=torch.uint8)
    for p in planted:
        if done <= p < done + m:
            chunk[p - done] = torch.from_numpy(best).cuda()
    idx.add_codes(chunk); done += m
for nq in (16, 1, 9):
    d, i = idx.search(q[:nq], 10, rotate=False)
    assert list(i[0, :6]) == planted, i[0]
    assert np.all(d[:, 1:] >= d[:, :-1]) and i.min() >= 0 and i.max() < n
print("1B-row check OK:", list(i[0, :6]))
