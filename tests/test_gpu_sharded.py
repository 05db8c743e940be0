"""GPU: the row-sharded search INSIDE the library (cvtmi_comm_*, cvtmi_opq_search_sharded*, csrc/shard.hip).

* world 1 through real RCCL ("comm_force_rccl"): ncclGetUniqueId / ncclCommInitRank / ncclAllGather are bound and run.
* world 2, 3 and 8 (BASELINE configs[3]'s rank count, shards of 0 and 1 rows included) as separate processes sharing this
  box's one GPU: RCCL refuses two ranks on one device, so the ranks exchange through the caller-supplied transport (gloo,
  staged through the host) -- the slot layout, the zero-copy local search into the slot and topk_merge_kernel<true> are
  exactly what the RCCL transport feeds.  `bench.py --gpus 8 --backend host` runs the configs[3] bench path at 8 ranks.
Results must equal a single handle holding every row, and the oracle, bit for bit (ids and distance bits)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case(seed, n, nq, D=128, M=16, K=256, dup=0):
    rng = np.random.default_rng(seed)
    books = (rng.normal(size=(M, K, D // M)) * 0.1).astype(np.float32)
    codes = rng.integers(0, K, size=(n, M), dtype=np.uint8)
    if dup and n > 4:  # exact duplicates on both sides of every shard boundary: ties across ranks
        for w in (2, 3, 8):
            for r in range(1, w):
                b = (n // w) * r + min(r, n % w)
                lo, hi = max(0, b - dup), min(n, b + dup)
                codes[lo:hi] = codes[lo]
    q = rng.normal(size=(nq, D)).astype(np.float32) * 0.1
    return books, codes, q


def test_world1_through_rccl(orc):
    import torch
    import cvt_amd
    cvt_amd.set_tuning("comm_force_rccl", 1)
    try:
        comm = cvt_amd.Comm(cvt_amd.Comm.unique_id(), 0, 1)
    finally:
        cvt_amd.set_tuning("comm_force_rccl", 0)
    assert comm.info()["transport"] == "rccl"
    books, codes, q = _case(1, 50_000, 37, dup=3)
    idx = cvt_amd.OpqIndex(np.zeros((1, 128), np.float32), books)
    idx.add_codes(torch.from_numpy(codes).cuda())
    qd = torch.from_numpy(q).cuda()
    for k in (1, 10, 100):
        d0, i0 = idx.search(qd, k, rotate=False)
        d1, i1 = idx.search_sharded(comm, qd, k, rotate=False)
        torch.cuda.synchronize()
        assert torch.equal(i0, i1) and torch.equal(d0.view(torch.int32), d1.view(torch.int32))
    inf = comm.info()
    assert inf["collectives"] == 3, inf  # ONE all-gather per search
    assert inf["bytes_per_rank"] == 16 + ((37 * 100 * 4 + 15) // 16 * 16) + ((37 * 100 * 8 + 15) // 16 * 16)   # status word + lists
    od, oi = orc.adc_search(q, books, codes, 100)
    assert np.array_equal(oi, i1.cpu().numpy()) and np.array_equal(bits(od), bits(d1.cpu().numpy()))
    # the host-pointer entry and the exchange step alone
    d2, i2 = idx.search_sharded(comm, q, 100, rotate=False)
    assert np.array_equal(i2, oi) and np.array_equal(bits(d2), bits(od))
    d3, i3 = comm.merge_topk(d1, i1, 100)
    assert torch.equal(i3, i1)
    comm.close(); idx.close()


WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["CVT_ROOT"])
import cvt_amd
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
z = np.load(os.environ["CVT_CASE"])
books, codes, q, ks = z["books"], z["codes"], z["q"], [int(v) for v in z["ks"]]
n = codes.shape[0]
a, b = cvt_amd.shard_range(n, rank, world)
comm = cvt_amd.Comm.over_torch_group(rank, world)
idx = cvt_amd.OpqIndex(np.zeros((1, books.shape[0] * books.shape[2]), np.float32), books)
if b > a:
    idx.add_codes(torch.from_numpy(codes[a:b]).cuda())
idx.set_id_base(a)
qd = torch.from_numpy(q).cuda()
out = {}
for k in ks:
    d, i = idx.search_sharded(comm, qd, k, rotate=False)
    torch.cuda.synchronize()
    out["d%d" % k] = d.cpu().numpy(); out["i%d" % k] = i.cpu().numpy()
inf = comm.info()
assert inf["transport"] == "custom" and inf["collectives"] == len(ks), inf
np.savez(os.environ["CVT_OUT"] + ".%d.npz" % rank, **out)
comm.close(); idx.close()
dist.destroy_process_group()
"""


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("world,n,nq", [(2, 200_000, 64), (3, 100_001, 33), (3, 2, 5), (2, 300, 9),
                                        (8, 200_003, 40), (8, 9, 5), (8, 5, 3)])   # 8 ranks: shards of 25 000, of 2 / 1 and of 1 / 0 rows
def test_multi_rank_one_gpu_matches_single_handle(world, n, nq, tmp_path, orc):
    import torch
    import cvt_amd
    books, codes, q = _case(100 + world + n, n, nq, dup=2)
    ks = [1, 10, 100] + ([300, 1000] if n in (100_001, 300) else [])   # beyond 128: the exact kernels, incl. k > rows per shard
    case = str(tmp_path / "case.npz")
    np.savez(case, books=books, codes=codes, q=q, ks=np.array(ks))
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   CVT_ROOT=ROOT, CVT_CASE=case, CVT_OUT=str(tmp_path / "out"))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for pp in procs:
                pp.kill()
            raise
        logs.append(o.decode(errors="replace"))
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    one = cvt_amd.OpqIndex(np.zeros((1, 128), np.float32), books)
    one.add_codes(torch.from_numpy(codes).cuda())
    for k in ks:
        d0, i0 = one.search(torch.from_numpy(q).cuda(), k, rotate=False)
        d0, i0 = d0.cpu().numpy(), i0.cpu().numpy()
        od, oi = orc.adc_search(q, books, codes, k)
        kk = min(k, n)
        assert np.array_equal(oi[:, :kk], i0[:, :kk]) and np.array_equal(bits(od[:, :kk]), bits(d0[:, :kk]))
        for r in range(world):
            z = np.load(str(tmp_path / "out") + ".%d.npz" % r)
            assert np.array_equal(z["i%d" % k], i0), (world, n, k, r)
            assert np.array_equal(bits(z["d%d" % k]), bits(d0)), (world, n, k, r)
    one.close()


def test_one_process_all_devices(orc):
    """cvtmi_comm_create_all (ncclCommInitAll) + the *_sharded_all searches: one process, one handle and one communicator per
    device, grouped all-gathers -- on this box over its single GPU (ndev = 1 through real RCCL); OPQ and flat (fp32 + uint8)"""
    import torch
    import cvt_amd
    nd = torch.cuda.device_count()
    comms = cvt_amd.Comm.create_all(nd)
    assert [c.info()["transport"] for c in comms] == ["rccl"] * nd and comms[0].info()["world"] == nd
    books, codes, q = _case(7, 60_000, 21, dup=2)
    idxs = []
    for d in range(nd):
        a, b = cvt_amd.shard_range(codes.shape[0], d, nd)
        with torch.cuda.device(d):
            ix = cvt_amd.OpqIndex(np.zeros((1, 128), np.float32), books)
            ix.add_codes(torch.from_numpy(codes[a:b]).cuda(d)); ix.set_id_base(a)
        idxs.append(ix)
    dd, ii = cvt_amd.search_sharded_all(idxs, comms, q, 100, rotate=False)
    od, oi = orc.adc_search(q, books, codes, 100)
    assert np.array_equal(ii, oi) and np.array_equal(bits(dd), bits(od))
    assert comms[0].info()["collectives"] == 1
    rng = np.random.default_rng(3)
    for metric, D in ((2, 64), (0, 32)):
        n = 30_000
        x = rng.integers(0, 256, size=(n, D), dtype=np.uint8) if metric == 2 else rng.normal(size=(n, D)).astype(np.float32)
        qq = x[rng.integers(0, n, 9)].copy()
        fl = []
        for d in range(nd):
            a, b = cvt_amd.shard_range(n, d, nd)
            with torch.cuda.device(d):
                f = cvt_amd.FlatIndex(metric, D); f.add(x[a:b]); f.set_id_base(a)
            fl.append(f)
        fd, fi = cvt_amd.search_sharded_all(fl, comms, qq, 10)
        odf, odi, oif = orc.flat_search(metric, x, qq, 10)
        assert np.array_equal(fi, oif)
        assert np.array_equal(fd, odi) if metric == 2 else np.array_equal(bits(fd), bits(odf))
        for f in fl:
            f.close()
    # a device whose local search fails: every caller gets CVTMI_ECOMM, nobody hangs
    cvt_amd.set_tuning("comm_inject_failure", nd - 1)
    try:
        with pytest.raises(cvt_amd.CvtmiError, match="error -6"):
            cvt_amd.search_sharded_all(idxs, comms, q, 10, rotate=False)
    finally:
        cvt_amd.set_tuning("comm_inject_failure", -1)
    dd2, ii2 = cvt_amd.search_sharded_all(idxs, comms, q, 100, rotate=False)   # and the communicators keep working
    assert np.array_equal(ii2, oi)
    for c in comms:
        c.close()
    for ix in idxs:
        ix.close()


FLAT_WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["CVT_ROOT"])
import cvt_amd
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
z = np.load(os.environ["CVT_CASE"])
x, q, metric, k = z["x"], z["q"], int(z["metric"]), int(z["k"])
a, b = cvt_amd.shard_range(x.shape[0], rank, world)
comm = cvt_amd.Comm.over_torch_group(rank, world)
ix = cvt_amd.FlatIndex(metric, x.shape[1])
if b > a:
    ix.add(torch.from_numpy(x[a:b]).cuda())
ix.set_id_base(a)
qd = torch.from_numpy(q).cuda()
d, i = ix.search_sharded(comm, qd, k)
torch.cuda.synchronize()
out = {"d": d.cpu().numpy(), "i": i.cpu().numpy()}
# a rank whose local search fails: nobody waits in the collective, and EVERY rank learns of it (CVTMI_ECOMM, -6):
#  - deferred check (default): the device-pointer call itself returns on the ranks that are fine, with the results voided
#    (+inf / -1) on every rank; cvtmi_comm_status reports the failure once the stream is synchronised -- and only once
#  - immediate check (comm_check_status = 1): every rank gets the error from the failing call
cvt_amd.set_tuning("comm_inject_failure", world - 1)
try:
    fd, fi = ix.search_sharded(comm, qd, k)
    torch.cuda.synchronize()
    voided = bool((fi == -1).all().item())
    own_failed = False
except cvt_amd.CvtmiError as e:
    voided, own_failed = True, "error -6" in str(e)          # the failing rank itself still gets its error at once
if own_failed:
    out["failed"] = np.array(1 if rank == world - 1 else -2)
else:
    try:
        comm.status()
        out["failed"] = np.array(0)
    except cvt_amd.CvtmiError as e:
        out["failed"] = np.array(1 if ("error -6" in str(e) and voided) else -1)
comm.status()                                                 # reported once: nothing left
cvt_amd.set_tuning("comm_check_status", 1)
try:
    ix.search_sharded(comm, qd, k)
    out["failed_now"] = np.array(0)
except cvt_amd.CvtmiError as e:
    out["failed_now"] = np.array(1 if "error -6" in str(e) else -1)
cvt_amd.set_tuning("comm_check_status", 2)
cvt_amd.set_tuning("comm_inject_failure", -1)
d2, i2 = ix.search_sharded(comm, qd, k)          # the communicator is still usable
torch.cuda.synchronize()
out["again"] = np.array(int(torch.equal(i2, i)))
np.savez(os.environ["CVT_OUT"] + ".%d.npz" % rank, **out)
comm.close(); ix.close()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world,metric,D,n", [(2, 2, 512, 50_001), (3, 2, 64, 9_000), (2, 1, 128, 40_000), (3, 0, 36, 7_001)])
def test_flat_row_shards_match_single_handle(world, metric, D, n, tmp_path, orc):
    """cvtmi_flat_search_sharded_dev at world 2 / 3 on one GPU (custom transport): uint8 (int32 distances through the fp32 fields)
    and fp32 metrics, exact duplicates on both sides of every shard boundary, against the checker over all rows; then a rank that
    fails locally makes every rank return CVTMI_ECOMM, and the next search works again"""
    rng = np.random.default_rng(world * 1000 + D)
    nq, k = 13, 20
    if metric == 2:
        x = rng.integers(0, 256, size=(n, D), dtype=np.uint8); q = rng.integers(0, 256, size=(nq, D), dtype=np.uint8)
    else:
        x = rng.normal(size=(n, D)).astype(np.float32); q = rng.normal(size=(nq, D)).astype(np.float32)
    for r in range(1, world):
        b = (n // world) * r + min(r, n % world)
        x[b - 3:b + 3] = x[b - 3]
        q[r] = x[b - 3]
    case = str(tmp_path / "case.npz")
    np.savez(case, x=x, q=q, metric=np.array(metric), k=np.array(k))
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   CVT_ROOT=ROOT, CVT_CASE=case, CVT_OUT=str(tmp_path / "out"))
        procs.append(subprocess.Popen([sys.executable, "-c", FLAT_WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for pp in procs:
                pp.kill()
            raise
        logs.append(o.decode(errors="replace"))
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    od, odi, oi = orc.flat_search(metric, x, q, k)
    for r in range(world):
        z = np.load(str(tmp_path / "out") + ".%d.npz" % r)
        assert np.array_equal(z["i"], oi), (world, metric, r)
        assert np.array_equal(z["d"], odi) if metric == 2 else np.array_equal(bits(z["d"]), bits(od)), (world, metric, r)
        assert int(z["failed"]) == 1 and int(z["failed_now"]) == 1 and int(z["again"]) == 1, (r, int(z["failed"]), int(z["failed_now"]), int(z["again"]))


def _bench_line(extra, timeout=600, gpus=2, large_rows=1 << 22):
    """`python bench.py --gpus N ...` with NO launcher around it and no RANK / WORLD_SIZE in the environment"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--nq", "600", "--steps", "2", "--warmup", "1",
           "--rows", "200000", "--large-rows", str(large_rows), "--oracle-queries", "8", "--comm-timeout", "60"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, stdin=subprocess.DEVNULL)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def _check_evidence(line, gpus=2):
    assert line["n_gpus"] == gpus and line["scaling"] == "strong" and line["value"] > 0
    assert line["transport"] == "custom" and line["rccl_ranks"] == 0   # one GPU here: the ranks exchange through the host
    assert line["collectives_per_search"] == 1.0
    ident = line["identical_to_oracle_sample"]
    assert ident["queries"] == 8 and ident["ids_identical"] and ident["distances_bit_identical"]
    assert ident["recall_at_1_identical_to_cpu"]
    assert 0.0 <= line["recall_at_1"] <= 1.0
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == gpus and cb["value"] > 0
    assert "roofline" in line and line["roofline"]["bound"] == "lds"
    assert line["sift1b"]["comm"]["world"] == gpus
    one = line["same_workload_on_one_gpu"]      # the N = 1 point of the same workload, measured in the same run by rank 0 alone
    assert one["value"] > 0 and one["sharded_result_identical"] and line["speedup_over_one_gpu"] > 0


def test_bench_gpus2_self_launch_host_transport():
    """bench.py --gpus N must start by itself and its line must prove what it measured"""
    line = _bench_line(["--backend", "host"])
    _check_evidence(line)
    assert "error" not in line


def test_bench_gpus8_configs3_shape_on_one_gpu():
    """BASELINE configs[3] at its own rank count before an 8-GPU node runs it -- `bench.py --gpus 8` over the host
    transport on this one GPU: 2^26 SIFT-shaped rows in 8 row shards (8 M rows = 128 MB of codes each), ONE all-gather of per-shard
    top-k per search, every rank's merged list equal to the oracle sample and to one handle holding all the rows.
    (Reference shape of the exchange: retrieval/vlindex/lib/FLANN/mpi/index.h:196-226.)"""
    line = _bench_line(["--backend", "host"], timeout=1500, gpus=8, large_rows=1 << 26)
    _check_evidence(line, gpus=8)
    assert "error" not in line
    assert line["config"]["rows"] == 1 << 26 and line["config"]["rows_per_gpu"] == 1 << 23


def test_bench_gpus2_without_second_gpu_still_prints_a_line():
    """asked for RCCL with more ranks than devices: the line appears, says why, and carries the host-transport result"""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices visible: RCCL itself runs (the driver's multi-GPU bench covers it)")
    line = _bench_line([])
    _check_evidence(line)
    assert "RCCL needs one device per rank" in line["error"]


def test_bench_gpus2_failing_rank_still_prints_a_line():
    """a rank whose LOCAL search fails (comm_inject_failure): nobody hangs, the line appears with the reason, and its identity evidence
    says what happened -- the voided results are NOT the oracle's"""
    line = _bench_line(["--backend", "host", "--tune", "comm_inject_failure=1"])
    assert "injected failure of rank 1" in line["error"] and "voided" in line["error"]
    assert line["identical_to_oracle_sample"]["ids_identical"] is False
    assert line["same_workload_on_one_gpu"]["sharded_result_identical"] is False
    assert line["collectives_per_search"] == 1.0     # the failing rank entered every collective
