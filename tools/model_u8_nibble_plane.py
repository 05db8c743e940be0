"""Round 6, VERDICT item 5 ("uint8 batched search: a first pass over 4-bit planes of the rows, exact second pass over the survivors"), modelled on
the bench's config-3 data shape before building anything: how many rows would a HIGH-NIBBLE first pass let through?
x = 16 xh + xl (xl in 0..15).  d = |x|^2 + |q|^2 - 2 q.x and q.x = 16 q.xh + q.xl, so with the row norms kept exactly
    d_mid = |x|^2 + |q|^2 - 2 (16 q.xh + 7.5 sum q)           (the low nibble at its centre)
    |d - d_mid| <= 2 |q . (xl - 7.5)| <= 2 |q| |xl - 7.5|      (Cauchy-Schwarz, |xl - 7.5| kept per row: one float)
A row can be among the k best only if d_mid - band <= tau, tau = the k-th smallest (d_mid + band)."""
import numpy as np
rng = np.random.default_rng(3)
n, d, k, nq = 400_000, 512, 10, 16
def codes(m, seed):
    g = np.random.default_rng(seed)
    f = np.maximum(g.normal(size=(m, d)).astype(np.float32), 0)
    f /= np.linalg.norm(f, axis=1, keepdims=True)
    return f
fx, fq = codes(n, 1), codes(nq, 2)
vmin, vmax = fx.min(0), fx.max(0)
enc = lambda f: np.clip(np.floor(255.0 * (f - vmin) / (vmax - vmin)), 0, 255).astype(np.int64)
x, q = enc(fx), enc(fq)
xh, xl = x >> 4, x & 15
xx = (x * x).sum(1)
lown = np.sqrt(((xl - 7.5) ** 2).sum(1))
print("rows %d x %d-d, top-%d; mean |xl - 7.5| = %.1f" % (n, d, k, lown.mean()))
for i in range(4):
    qi = q[i]
    dist = xx + (qi * qi).sum() - 2 * (x @ qi)
    dmid = xx + (qi * qi).sum() - 2 * (16 * (xh @ qi) + 7.5 * qi.sum())
    band = 2 * np.sqrt((qi * qi).sum()) * lown
    tau = np.partition(dmid + band, k - 1)[k - 1]
    passed = int((dmid - band <= tau).sum())
    dk = np.partition(dist, k - 1)[k - 1]
    exact_pass = int((dist <= tau).sum())
    print("query %d: k-th distance %d, median distance %d, band (mean) +-%d -> first pass lets %d rows through = %.1f %% (rows with d <= tau: %d)" % (
        i, dk, int(np.median(dist)), int(band.mean()), passed, 100.0 * passed / n, exact_pass))
