"""Row-sharded search across the GPUs of one node: one process per GPU, each holding a
contiguous block of code rows; per query batch every rank scans its shard, then ONE all-gather of the
per-shard (distance, id) top-k over RCCL/xGMI and a k-way merge on every rank.  Same shape as the only
distributed search in the reference tree (FLANN-MPI: local search, id += offset, reduce with
ResultsMerger; retrieval/vlindex/lib/FLANN/mpi/index.h:74-108, :196-226).

The class is transport- and device-agnostic (torch.distributed group + injected local_search / merge
callables) so the same plumbing runs under gloo on CPU in tests/ and under nccl(=RCCL) in bench.py.
"""


def shard_range(n_total, rank, world):
    """Contiguous row block of `rank`: the first (n_total % world) ranks own one extra row."""
    base, rem = divmod(n_total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


class ShardedSearch:
    def __init__(self, local_search, merge, world=1, rank=0, group=None):
        """local_search(q, k) -> (dist [nq][k] f32, ids [nq][k] i64 with GLOBAL ids, -1 = padding);
        merge(dist [nq][L][k], ids [nq][L][k], k) -> (dist [nq][k], ids [nq][k])."""
        self.local_search, self.merge = local_search, merge
        self.world, self.rank, self.group = world, rank, group

    def search(self, q, k):
        d, i = self.local_search(q, k)
        if self.world == 1:
            return d, i
        import torch
        import torch.distributed as dist
        nq = d.shape[0]
        # gloo has no device collectives: stage through the host (debug runs of bench.py on one GPU only;
        # the RCCL path exchanges HBM buffers directly)
        dev = d.device
        stage = d.is_cuda and dist.get_backend(self.group) == "gloo"
        sd, si = (d.cpu(), i.cpu()) if stage else (d, i)
        # flat 1-D buffers: the one shape every backend's all_gather_into_tensor agrees on
        gd = torch.empty(self.world * nq * k, dtype=sd.dtype, device=sd.device)
        gi = torch.empty(self.world * nq * k, dtype=si.dtype, device=si.device)
        # rank order == ascending id range: the merge's tie rule relies on it
        dist.all_gather_into_tensor(gd, sd.contiguous().view(-1), group=self.group)
        dist.all_gather_into_tensor(gi, si.contiguous().view(-1), group=self.group)
        gd = gd.view(self.world, nq, k).permute(1, 0, 2).contiguous().to(dev)
        gi = gi.view(self.world, nq, k).permute(1, 0, 2).contiguous().to(dev)
        return self.merge(gd, gi, k)
