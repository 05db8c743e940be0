"""CPU, world_size 2 / 3 (gloo): the row-shard -> all-gather -> merge plumbing of cvt_amd/sharded.py (46 lines of torch.distributed
glue) gives the same answer as one process over the whole database.  The per-shard search and the merge are the CPU oracle here
(test infrastructure).

What this file does NOT cover: the library's own exchange, csrc/shard.hip (slot layout, the in-place ncclAllGather / the
caller-supplied transport, topk_merge_kernel<true>, the status word).  That is exercised on the GPU box only -- world 1 through
real RCCL, worlds 2 / 3 / 8 through the custom transport, bench.py --gpus 2 / 8: tests/test_gpu_sharded.py.  This file proves the
N > 1 decomposition (shard ranges, id bases, ties across boundaries, k larger than a shard) on the CPU; that one proves the code
that ships."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, tmp):
    import torch
    import torch.distributed as dist
    from cvt_amd import sharded
    from oracle import binding as ob
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = ob.Oracle()
    z = np.load(os.path.join(tmp, "case.npz"))
    books, codes, q, k = z["books"], z["codes"], z["q"], int(z["k"])
    a, b = sharded.shard_range(codes.shape[0], rank, world)

    def local(qq, kk):
        d, i = orc.adc_search(qq.numpy(), books, codes[a:b], kk, id_base=a)
        return torch.from_numpy(d), torch.from_numpy(i)

    def merge(d, i, kk):
        md, mi = orc.merge_topk(d.numpy(), i.numpy(), kk)
        return torch.from_numpy(md), torch.from_numpy(mi)

    s = sharded.ShardedSearch(local, merge, world, rank)
    d, i = s.search(torch.from_numpy(q), k)
    np.savez(os.path.join(tmp, "out_%d.npz" % rank), d=d.numpy(), i=i.numpy())
    dist.destroy_process_group()


def test_shard_range_partitions():
    from cvt_amd import sharded
    for n in (0, 1, 7, 1000, 1_000_003):
        for w in (1, 2, 3, 8):
            spans = [sharded.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[r][1] == spans[r + 1][0] for r in range(w - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_equals_single(tmp_path, orc, world):
    import torch.multiprocessing as mp
    rng = np.random.default_rng(17)
    D, M, K, n, nq, k = 32, 4, 256, 5001, 6, 10
    books = rng.normal(size=(M, K, D // M)).astype(np.float32)
    codes = rng.integers(0, K, size=(n, M), dtype=np.uint8)
    codes[4000] = codes[3]; codes[2500] = codes[3]          # ties across shard boundaries
    q = rng.normal(size=(nq, D)).astype(np.float32)
    q[0, :8] = books[0, codes[3, 0]]
    np.savez(tmp_path / "case.npz", books=books, codes=codes, q=q, k=k)
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    od, oi = orc.adc_search(q, books, codes, k)
    for r in range(world):
        z = np.load(tmp_path / ("out_%d.npz" % r))
        assert np.array_equal(z["i"], oi), r
        assert np.array_equal(z["d"].view(np.uint32), od.view(np.uint32)), r
