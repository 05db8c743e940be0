"""CPU model (round 6): pass rate of a saturating, threshold-relative 8-bit lower-bound filter for the ADC scan
on the bench's own data shape (synth.sift_like rows, random rotation, 4-iteration k-means codebooks, M = 16, top-100).

Runs HERE (no GPU): torch on the CPU for data / codebooks / encode, numpy for the scan model.  The random streams differ from
the GPU generator's (same distribution).  Nothing here is product code.

Model of one workgroup's scan of rows 0 .. N-1 in order for one query:
  * exact threshold T(n) = k-th smallest exact distance among the rows seen so far, refreshed every `CK` rows (a compaction);
  * 15-bit scheme of today (adc_scan.hip:554): pass = exact distance < T (the filter is near-exact), as the natural floor;
  * 8-bit scheme: entries e = min(c, floor((LUT - min_m) / u)), unit u = (T_build - bias) / TQ0 fixed when the tables are
    built; a row survives when sum_m e < Tq, Tq = ceil((T - bias) / u) + 1; with the bit-7 start value 128 - Tq the sum must
    stay <= 255, i.e. Tq >= 16 c - 127: when the threshold has fallen below that the tables are REBUILT at the current T.
Prints per scheme: survivors per query (and as a share of rows), rebuilds per scan.

MODE=causes : tables rebuilt at every checkpoint (no range limit): floor loss and clamp separately, by scan phase.
MODE=once   : tables built ONCE at the seed threshold, wider entries (the matrix-core widening of tools/ubench/scan_loop_u8.hip
              allows entries <= 42 in int32 sums, no lower limit on Tq), by scan phase.
The data / codebooks / tables are cached in /tmp/model_cache.npz between modes.
"""
import os
import sys
import time
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvt_amd import synth  # noqa: E402

N = int(os.environ.get("ROWS", 1_000_000))
NQ = int(os.environ.get("NQ", 48))
K = int(os.environ.get("K", 100))
M, KC, D = 16, 256, 128
CK = int(os.environ.get("CK", 16384))
SEED_ROWS = 2048


def kmeans(x, k, iters, seed):
    g = torch.Generator().manual_seed(seed)
    c = x[torch.randperm(x.shape[0], generator=g)[:k]].clone()
    for _ in range(iters):
        d = torch.cdist(x, c)
        a = d.argmin(1)
        for j in range(k):
            s = x[a == j]
            if s.shape[0]:
                c[j] = s.mean(0)
    return c


def cached_inputs():
    cache = "/tmp/model_cache.npz"
    if os.path.exists(cache):
        z = np.load(cache)
        return z["codes"], z["lut"]
    torch.set_num_threads(8)
    R = torch.from_numpy(synth.random_rotation(D, seed=7))
    sample = synth.sift_like(100_000, D, seed=0xC0FFEE) @ R.T
    step = D // M
    books = torch.stack([kmeans(sample[:, m * step:(m + 1) * step].contiguous(), KC, 4, 1234 + m) for m in range(M)])
    codes = np.empty((N, M), np.uint8)
    for a in range(0, N, 1 << 18):
        b = min(N, a + (1 << 18))
        x = synth.sift_like(b - a, D, seed=0xC0FFEE, row_begin=a) @ R.T
        for m in range(M):
            codes[a:b, m] = torch.cdist(x[:, m * step:(m + 1) * step], books[m]).argmin(1).numpy()
    q = synth.sift_like(64, D, seed=0xBEEF) @ R.T
    lut = torch.stack([torch.cdist(q[:, m * step:(m + 1) * step], books[m]) ** 2 for m in range(M)], 1).numpy().astype(np.float32)
    np.savez(cache, codes=codes, lut=lut)
    return codes, lut


PHASES = [0, 65536, 262144, 524288, 1000000]


def by_phase(mode):
    codes, lut = cached_inputs()
    nq = min(NQ, lut.shape[0])
    mn = lut.min(2); bias = mn.sum(1); delta = lut - mn[:, :, None]
    cidx = codes.astype(np.int64)
    tot = {}

    def acc(k, v):
        tot[k] = tot.get(k, 0) + v
    for qi in range(nq):
        d = np.zeros(N, np.float32)
        for m in range(M):
            d += lut[qi, m][cidx[:, m]]
        dl = np.stack([delta[qi, m][cidx[:, m]] for m in range(M)], 1)
        ckpts = [SEED_ROWS] + list(range(CK, N, CK)); edges = ckpts[1:] + [N]
        thr = [np.partition(d[:n], K - 1)[K - 1] for n in ckpts]
        B0 = thr[0] - bias[qi]
        lo = 0
        for T, hi in zip(thr, edges):
            B = T - bias[qi]
            ph = np.searchsorted(PHASES, lo, side="right") - 1
            acc(("exact", ph), int((d[lo:hi] < T).sum()))
            if mode == "causes":
                for tq0 in (120, 60):
                    u = B / tq0
                    fl = np.floor(dl[lo:hi] / u)
                    acc(("floor only tq=%d" % tq0, ph), int((fl.sum(1) < tq0 + 1).sum()))
                    for c in (15, 20, 31):
                        acc(("clamp%d only tq=%d" % (c, tq0), ph), int((np.minimum(c * u, dl[lo:hi]).sum(1) < B).sum()))
                        acc(("floor+clamp%d tq=%d" % (c, tq0), ph), int((np.minimum(c, fl).sum(1) < tq0 + 1).sum()))
            else:
                for c, tq0 in ((42, 250), (42, 336), (42, 500), (42, 650), (15, 120), (31, 248), (63, 500), (63, 800)):
                    u = B0 / tq0
                    tq = np.ceil(B / u) + 1
                    fl = np.minimum(c, np.floor(dl[lo:hi] / u)).sum(1)
                    acc(("once c=%d tq0=%d" % (c, tq0), ph), int((fl < tq).sum()))
            lo = hi
    print("mode %s: rows %d queries %d k %d checkpoint every %d rows" % (mode, N, nq, K, CK))
    print("survivors per query by scan phase (rows 0-64K, 64K-256K, 256K-512K, 512K-1M), total")
    for n in sorted(set(k[0] for k in tot)):
        v = [tot.get((n, p), 0) / nq for p in range(4)]
        print("%-28s %s  total %.0f" % (n, " ".join("%8.0f" % x for x in v), sum(v)))


def main():
    mode = os.environ.get("MODE", "rebuild")
    if mode != "rebuild":
        return by_phase(mode)
    torch.set_num_threads(8)
    t0 = time.time()
    R = torch.from_numpy(synth.random_rotation(D, seed=7))
    sample = synth.sift_like(100_000, D, seed=0xC0FFEE) @ R.T
    step = D // M
    books = torch.stack([kmeans(sample[:, m * step:(m + 1) * step].contiguous(), KC, 4, 1234 + m) for m in range(M)])
    print("books %.0fs" % (time.time() - t0), flush=True)
    codes = np.empty((N, M), np.uint8)
    for a in range(0, N, 1 << 18):
        b = min(N, a + (1 << 18))
        x = synth.sift_like(b - a, D, seed=0xC0FFEE, row_begin=a) @ R.T
        for m in range(M):
            codes[a:b, m] = torch.cdist(x[:, m * step:(m + 1) * step], books[m]).argmin(1).numpy()
    print("codes %.0fs" % (time.time() - t0), flush=True)
    q = synth.sift_like(NQ, D, seed=0xBEEF) @ R.T
    lut = torch.stack([torch.cdist(q[:, m * step:(m + 1) * step], books[m]) ** 2 for m in range(M)], 1).numpy().astype(np.float32)  # [nq][M][256]
    mn = lut.min(2)            # [nq][M]
    bias = mn.sum(1)
    delta = lut - mn[:, :, None]
    cidx = codes.astype(np.int64)

    schemes = []
    for c in (15, 14, 13, 12):
        schemes.append(("u8 c=%d" % c, c, 127, 16 * c - 127))
    res = {name: dict(surv=0, rebuilds=0) for name, *_ in schemes}
    exact_pass = 0
    ratio_seed_final = []
    bias_over_T = []
    for qi in range(NQ):
        d = np.zeros(N, np.float32)
        for m in range(M):
            d += lut[qi, m][cidx[:, m]]
        dl = np.stack([delta[qi, m][cidx[:, m]] for m in range(M)], 1)  # [N][M] entry - min
        # exact thresholds per checkpoint
        ckpts = [SEED_ROWS] + list(range(CK, N, CK))
        thr = []
        for n in ckpts:
            thr.append(np.partition(d[:n], K - 1)[K - 1])
        Tfinal = np.partition(d, K - 1)[K - 1]
        ratio_seed_final.append((Tfinal - bias[qi]) / (thr[0] - bias[qi]))
        bias_over_T.append(bias[qi] / Tfinal)
        # natural floor: rows below the threshold in force when they are scanned
        edges = ckpts[1:] + [N]
        lo = 0
        for T, hi in zip(thr, edges):
            exact_pass += int((d[lo:hi] < T).sum())
            lo = hi
        for name, c, tq0, tqmin in schemes:
            u = None
            lo = 0
            for T, hi in zip(thr, edges):
                B = T - bias[qi]
                if u is None or np.ceil(B / u) + 1 < tqmin:
                    u = B / (tq0 - 2)
                    res[name]["rebuilds"] += 1
                tq = np.ceil(B / u) + 1
                e = np.minimum(c, np.floor(dl[lo:hi] / u)).sum(1)
                res[name]["surv"] += int((e < tq).sum())
                lo = hi
        if qi % 8 == 7:
            print("query %d %.0fs" % (qi + 1, time.time() - t0), flush=True)
    print("rows %d queries %d k %d checkpoint every %d rows" % (N, NQ, K, CK))
    print("(T_final - bias) / (T_seed - bias): mean %.3f min %.3f max %.3f;  bias / T_final mean %.3f" %
          (np.mean(ratio_seed_final), np.min(ratio_seed_final), np.max(ratio_seed_final), np.mean(bias_over_T)))
    print("exact filter (today's 15-bit scheme, idealised): %.1f candidates per query = %.4f %% of rows" % (exact_pass / NQ, 100.0 * exact_pass / NQ / N))
    for name, *_ in schemes:
        r = res[name]
        print("%-10s survivors %.1f per query = %.4f %% of rows, table builds per scan %.2f" % (name, r["surv"] / NQ, 100.0 * r["surv"] / NQ / N, r["rebuilds"] / NQ))


if __name__ == "__main__":
    main()
