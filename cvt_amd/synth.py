"""Synthetic inputs for tests and bench.py: SIFT-shaped 128-d vectors, random
rotations, quickly-trained sub-codebooks.  Deterministic per (seed, row chunk) so that any row shard
can generate exactly its own rows on its own GPU, with no host round trip.
"""
import numpy as np

CHUNK = 1 << 18  # rows per generator chunk; shards are generated chunk by chunk


N_CENTERS = 4096  # synthetic "visual words": rows are a word plus per-row variation, like real descriptors


def _centers(torch, D, seed, device):
    g = torch.Generator(device=device)
    g.manual_seed((seed * 7919 + 17) & 0x7FFFFFFFFFFFFFFF)
    c = torch.randn((N_CENTERS, D), generator=g, device=device).abs_() * 42.0
    z = torch.rand((N_CENTERS, D), generator=g, device=device) < 0.3
    return torch.where(z, torch.zeros_like(c), c)


def _sift_chunk_torch(torch, n, D, seed, chunk_idx, device, centers):
    g = torch.Generator(device=device)
    g.manual_seed((seed * 1000003 + chunk_idx) & 0x7FFFFFFFFFFFFFFF)
    # heavy-tailed non-negative integers, many exact zeros, clipped at 255 like SIFT bins:
    # a cluster centre ("visual word") plus multiplicative and additive per-row variation
    cid = torch.randint(0, N_CENTERS, (n,), generator=g, device=device)
    x = centers[cid] * (0.75 + 0.5 * torch.rand((n, D), generator=g, device=device))
    x = x + torch.randn((n, D), generator=g, device=device).abs_() * 9.0
    z = torch.rand((n, D), generator=g, device=device) < 0.1
    x = torch.where(z, torch.zeros_like(x), x).floor_().clamp_(max=255.0)
    return x


def sift_like(n, D=128, seed=0xC0FFEE, row_begin=0, device="cpu", rootsift=True):
    """rows [row_begin, row_begin+n) of the infinite synthetic SIFT-like matrix, fp32 torch tensor.

    rootsift=True applies the RootSIFT map of the reference's indexer
    (hnsw_sifts_retrieval/siftsIndex.cpp:54-71: L1-normalise, sqrt, L2-normalise) so that vectors are
    unit-norm like the features the reference actually indexes (opq/data)."""
    import torch
    out = torch.empty((n, D), dtype=torch.float32, device=device)
    centers = _centers(torch, D, 0xC0FFEE, device)  # the SAME words for database and queries, whatever `seed`
    r = row_begin
    end = row_begin + n
    while r < end:
        c = r // CHUNK
        c0 = c * CHUNK
        full = _sift_chunk_torch(torch, CHUNK, D, seed, c, device, centers)
        lo, hi = r - c0, min(end, c0 + CHUNK) - c0
        out[r - row_begin:r - row_begin + (hi - lo)] = full[lo:hi]
        r = c0 + hi
    if rootsift:
        out = out / (out.abs().sum(dim=1, keepdim=True) + 1e-7)
        out = out.sqrt_()
        out = out / out.norm(dim=1, keepdim=True).clamp_min(1e-12)
    return out.contiguous()


def random_rotation(D, seed=7):
    """Random orthonormal fp32 matrix (QR of a Gaussian)."""
    rng = np.random.default_rng(seed)
    q, r = np.linalg.qr(rng.normal(size=(D, D)))
    q = q * np.sign(np.diag(r))
    return np.ascontiguousarray(q.astype(np.float32))


def random_permutation(D, seed=7):
    return np.random.default_rng(seed).permutation(D).astype(np.int32)


def train_books(x_rot, M, K=256, iters=6, seed=1234):
    """Sub-space codebooks for a zero-coarse-centroid model (the exhaustive configs): cvtmi_kmeans (the library's
    own Lloyd iteration, csrc/kmeans.hip) per sub-space on a (rotated) device sample -> numpy [M][K][step]."""
    from . import capi
    n, D = x_rot.shape
    step = D // M
    if not x_rot.is_cuda:
        raise RuntimeError("train_books needs a device sample: the k-means runs in libcvtmi (csrc/kmeans.hip), there is no CPU path")
    books = np.empty((M, K, step), dtype=np.float32)
    for m in range(M):
        cen, _, _ = capi.kmeans(x_rot[:, m * step:(m + 1) * step].contiguous(), K, iters, seed)
        books[m] = cen.cpu().numpy()
    return books
