#!/usr/bin/env python3
"""bench.py -- queries/sec of the OPQ-ADC search hot path on MI355X.

One "step" = one pass of the hot path over one batch of nq synthetic queries against the HBM-resident
code index: query rotation (fp32 MFMA GEMM) -> per-query distance tables built in LDS -> ADC scan of
every code row -> k smallest (distance, id) per query [-> for N > 1: RCCL all-gather of the per-shard
top-k and a k-way merge on every rank].  Inputs are resident in HBM when the timed region starts.

Workload: BASELINE.json configs[1], SIFT-1M (synthetic SIFT-shaped 128-d rows), OPQ M=16 K=256, top-100,
nq=10000 queries per step and GPU.  At N > 1 (--scaling):
  weak     (default) every rank serves its own batch of nq queries against its replica of the 16 MB code matrix:
           per-GPU work is fixed, value = N * nq queries per step -- how a database this small is served;
  strong   the SAME nq-query batch is split over the ranks (per-GPU batches of nq / N).
Multi-GPU layout (--layout):
  rows     the north-star layout: rank r owns a contiguous row shard, every rank scans all queries, then ONE
           RCCL all-gather of the per-shard top-k and a k-way merge on every rank;
  queries  the code matrix is replicated (16 MB at SIFT-1M) and the query batch is split over the ranks:
           no data-path collective at all;
  auto     queries when the code matrix is < 1 GiB per GPU (replication is free and a 125 K-row shard is too
           small to amortise a workgroup's fixed cost), rows otherwise (SIFT-1B: 2 GB per GPU).
At N > 1 the JSON line also carries the throughput of the row-sharded path measured in the same run, and at
every N the SIFT-1B probe ("row_sharded_large"): --large-rows code rows in total (default 2^30 = 16 GiB of codes) sharded by
row over the ranks, 1024 queries on every rank, RCCL all-gather of the per-shard top-k + merge -- the north-star
layout at a shard size where the scan streams from HBM (strong scaling in rows: compare its value across N).

    python bench.py                       # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--nq", type=int, default=10_000)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--M", type=int, default=16)
    ap.add_argument("--qtile", type=int, default=0)
    ap.add_argument("--splits", type=int, default=0)
    ap.add_argument("--variant", type=int, default=-1, help="scan kernel variant (cvtmi.h), -1 = library default")
    ap.add_argument("--layout", choices=["auto", "rows", "queries"], default="auto")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="N > 1 with the queries layout: per-GPU batch fixed (weak) or the one batch split (strong)")
    ap.add_argument("--large-rows", type=int, default=1 << 30, help="total rows of the row-sharded SIFT-1B probe (0 = skip): 16 GiB of codes in all")
    ap.add_argument("--large-nq", type=int, default=1024)
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="gloo = debug: several ranks on one GPU")
    ap.add_argument("--cpu-sample", type=int, default=256, help="queries timed on the CPU baseline (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=-1, help="threads of the all-cores CPU leg (-1 = all logical cores, 0 = skip)")
    ap.add_argument("--recall-sample", type=int, default=1000)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import cvt_amd
    from cvt_amd import sharded, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with --nproc-per-node equal to --gpus (got WORLD_SIZE=%d)" % world
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path"
    if args.backend == "gloo":
        local_rank = 0  # debug: every rank on GPU 0, collectives staged through the host
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group("gloo")
    cvt_amd.lib()

    D, M, K, k, nq = 128, args.M, 256, args.k, args.nq
    zero_coarse = np.zeros((1, D), np.float32)
    R = synth.random_rotation(D, seed=7)

    # ---- model: rank 0 trains the sub-codebooks on a 100K-row sample, everyone gets the same bytes ----
    books_t = torch.empty((M, K, D // M), dtype=torch.float32, device=dev)
    if rank == 0:
        tmp = cvt_amd.OpqIndex(zero_coarse, np.zeros((M, K, D // M), np.float32), R=R)
        sample = tmp.rotate(synth.sift_like(100_000, D, seed=0xC0FFEE, device=dev))
        books_t.copy_(torch.from_numpy(synth.train_books(sample, M, K, iters=4)))
        tmp.close(); del sample
    if world > 1:
        if args.backend == "nccl":
            dist.broadcast(books_t, src=0)
        else:
            bc = books_t.cpu(); dist.broadcast(bc, src=0); books_t.copy_(bc)
    books = books_t.cpu().numpy()

    layout = args.layout
    if layout == "auto":
        layout = "queries" if args.rows * M < (1 << 30) else "rows"
    if world == 1:
        layout = "rows"

    # ---- index build on device: generate -> rotate (MFMA GEMM) -> encode -> append; only codes stay ----
    def build_index(r0, r1):
        ix = cvt_amd.OpqIndex(zero_coarse, books, R=R)
        ix.reserve(r1 - r0); ix.set_id_base(r0)
        rows_done, t_acc = 0, 0.0
        for a in range(r0, r1, synth.CHUNK):
            b = min(r1, a + synth.CHUNK)
            x = synth.sift_like(b - a, D, seed=0xC0FFEE, row_begin=a, device=dev)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            _, codes = ix.encode(ix.rotate(x))
            ix.add_codes(codes)
            torch.cuda.synchronize(); t_acc += time.perf_counter() - t0  # data generation excluded
            rows_done += b - a
        ix.set_param("qtile", args.qtile); ix.set_param("splits", args.splits); ix.set_param("profile", 1)
        if args.variant >= 0: ix.set_param("scan_variant", args.variant)
        return ix, rows_done, t_acc

    q = synth.sift_like(nq, D, seed=0xBEEF, device=dev)  # identical on every rank

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            out = fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = fn()
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, out

    r0, r1 = sharded.shard_range(args.rows, rank, world)
    extra = {}
    if layout == "rows":
        idx, enc_rows, enc_time = build_index(r0, r1)
        searcher = sharded.ShardedSearch(lambda qq, kk: idx.search(qq, kk, rotate=True), cvt_amd.topk_merge, world, rank)
        for _ in range(args.warmup):
            searcher.search(q, k)
        barrier(); idx.last_scan()  # drop the warm-up launches from the kernel-time statistics
        elapsed, out = timed(lambda: searcher.search(q, k), args.steps, 0)
        rows_here = r1 - r0
        par = "row-sharded x%d + RCCL all-gather of per-shard top-k + merge" % world if world > 1 else "1 GPU"
    else:
        idx, enc_rows, enc_time = build_index(0, args.rows)  # replica of the whole code matrix
        if args.scaling == "weak":
            q_mine = synth.sift_like(nq, D, seed=0xBEEF + 7919 * rank, device=dev) if rank else q
        else:
            q0, q1 = sharded.shard_range(nq, rank, world)
            q_mine = q[q0:q1].contiguous()
        for _ in range(args.warmup):
            idx.search(q_mine, k, rotate=True)
        barrier(); idx.last_scan()
        elapsed, out = timed(lambda: idx.search(q_mine, k, rotate=True), args.steps, 0)
        rows_here = args.rows
        par = ("code matrix replicated x%d, every rank serves its own batch of %d queries, no data-path collective" % (world, nq)
               if args.scaling == "weak" else
               "code matrix replicated x%d, one batch of %d queries split over the ranks, no data-path collective" % (world, nq))
        # the north-star layout measured in the same run (row shard of the replica + all-gather + merge)
        shard = cvt_amd.OpqIndex(zero_coarse, books, R=R)
        codes_t = torch.from_numpy(idx.get_entries()[2][r0:r1]).to(dev)
        shard.add_codes(codes_t); shard.set_id_base(r0)
        rs = sharded.ShardedSearch(lambda qq, kk: shard.search(qq, kk, rotate=True), cvt_amd.topk_merge, world, rank)
        el2, out2 = timed(lambda: rs.search(q, k), args.steps, args.warmup)
        extra["row_sharded"] = {"value": round(nq * args.steps / el2, 1), "unit": "queries/s",
                                "ms_per_step": round(el2 / args.steps * 1e3, 4),
                                "what": "same database row-sharded x%d, all queries on every rank, RCCL all-gather of "
                                        "per-shard top-%d + merge" % (world, k)}
    scan = idx.last_scan()  # mean HIP-event duration of the scan kernel over the timed steps

    # ---- SIFT-1B-shaped probe: a large code matrix row-sharded over the ranks (north-star layout) ----
    if args.large_rows > 0:
        try:
            l0, l1 = sharded.shard_range(args.large_rows, rank, world)
            # codes + the scan's rotated copy + generation chunks must fit; every rank takes the same decision
            free_b, _ = torch.cuda.mem_get_info(dev)
            fits = torch.tensor([1 if free_b > (l1 - l0) * M * 2.3 + (2 << 30) else 0], dtype=torch.int32,
                                device=dev if args.backend == "nccl" else "cpu")
            if world > 1:
                dist.all_reduce(fits, op=dist.ReduceOp.MIN)
            if int(fits.item()) == 0:
                raise MemoryError("not enough free HBM for %d code rows per GPU" % (l1 - l0))
            g = torch.Generator(device=dev); g.manual_seed(0x51F7 + rank)
            big = cvt_amd.OpqIndex(zero_coarse, books, R=R)
            big.reserve(l1 - l0); big.set_id_base(l0)
            for a in range(l0, l1, 1 << 24):
                b = min(l1, a + (1 << 24))
                big.add_codes(torch.randint(0, 256, (b - a, M), generator=g, device=dev, dtype=torch.uint8))
            big.set_param("profile", 1)
            if args.variant >= 0: big.set_param("scan_variant", args.variant)
            ql = q[:min(args.large_nq, nq)].contiguous()
            ls = sharded.ShardedSearch(lambda qq, kk: big.search(qq, kk, rotate=True), cvt_amd.topk_merge, world, rank)
            for _ in range(2):
                ls.search(ql, k)
            barrier(); big.last_scan()
            lsteps = max(2, min(args.steps, 5))
            el3, _ = timed(lambda: ls.search(ql, k), lsteps, 0)
            sc3 = big.last_scan()
            extra["row_sharded_large"] = {
                "value": round(ql.shape[0] * lsteps / el3, 1), "unit": "queries/s", "ms_per_step": round(el3 / lsteps * 1e3, 4),
                "rows_total": args.large_rows, "rows_per_gpu": l1 - l0, "nq": int(ql.shape[0]), "scaling": "strong (rows)",
                "scan_kernel_ms": round(sc3["ms"], 4),
                "scan_algorithmic_GBps": round(sc3["code_bytes"] / (sc3["ms"] * 1e-3) / 1e9, 1),
                "scan_frac_of_hbm_peak": round(sc3["code_bytes"] / (sc3["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "what": "uniform random codes, %d rows row-sharded x%d, %d queries on every rank, top-%d, %s" % (
                    args.large_rows, world, ql.shape[0], k,
                    "RCCL all-gather of per-shard top-k + merge" if world > 1 else "single shard")}
            big.close(); del big
        except MemoryError as e:  # decided identically on every rank (all-reduce above): skip, keep the headline line
            extra["row_sharded_large"] = {"error": str(e)}

    result = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        weak = world > 1 and layout == "queries" and args.scaling == "weak"
        nq_step = nq * world if weak else nq
        qps = nq_step * args.steps / elapsed
        achieved = scan["code_bytes"] / (scan["ms"] * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "scan_traffic.json")
        default_shape = world == 1 and args.rows == 1_000_000 and nq == 10_000 and k == 100 and M == 16 and not args.qtile and not args.splits
        if default_shape and os.path.exists(pmc):  # PMC passes were taken on exactly this launch shape
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        result = {
            "metric": "queries/sec, OPQ-ADC top-%d over 128-d SIFT-1M" % k,
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak" if (weak or world == 1) else "strong", "vs_baseline": None,
            "dtype": "u8 codes / f32 distances", "data": "synthetic",
            "config": {"workload": "SIFT-1M synthetic 128-d, OPQ M=%d K=256 (dense 128x128 rotation), ADC scan + top-%d, "
                                   "nq=%d queries per step and GPU" % (M, k, nq),
                       "rows": args.rows, "rows_per_gpu": rows_here, "nq_per_step": nq_step, "nq_per_gpu": nq if (weak or world == 1 or layout == "rows") else nq // world, "k": k, "M": M,
                       "parallelism": par, "layout": layout,
                       "qtile": scan["qtile"], "row_splits": scan["splits"]},
            "roofline": {"bound": "hbm", "kernel": "adc_scan16q_kernel (M=%d, %d queries per pass)" % (M, scan["qtile"]) if (M == 16 and scan["qtile"] == 8) else "adc_scan kernel (M=%d, %d queries per pass)" % (M, scan["qtile"]),
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": scan["code_bytes"], "kernel_ms": round(scan["ms"], 4),
                         "lds_lookups_per_s": round(scan["code_bytes"] * scan["qtile"] / (scan["ms"] * 1e-3) / 1e12, 2),
                         "lds_lookups_per_s_unit": "T table look-ups/s (LDS ceiling of this kernel, 256 B/clk/CU: 78 at 2.35 GHz, 66 at the ~2.0 GHz it sustains)"},
            "encode": {"rows_per_s": round(enc_rows / enc_time, 1), "what": "rotate (MFMA GEMM) + PQ encode + append, this rank"},
        }
        result.update(extra)

    # ---- outside the timed region: recall@1 and the CPU baseline (rank 0, N = 1 only) ----
    if rank == 0 and world == 1:
        d_gpu, i_gpu = out
        ns = min(args.recall_sample, nq)
        if ns > 0:
            best = torch.full((ns,), float("inf"), device=dev); arg = torch.zeros((ns,), dtype=torch.int64, device=dev)
            qs = q[:ns]
            for a in range(0, args.rows, synth.CHUNK):
                b = min(args.rows, a + synth.CHUNK)
                x = synth.sift_like(b - a, D, seed=0xC0FFEE, row_begin=a, device=dev)
                dd = torch.cdist(qs, x)
                m, j = dd.min(dim=1)
                upd = m < best
                best = torch.where(upd, m, best); arg = torch.where(upd, j + a, arg)
            result["recall_at_1"] = round(float((i_gpu[:ns, 0] == arg).float().mean().item()), 4)
            result["recall_at_1_what"] = "ADC top-1 == exact fp32 L2 nearest neighbour, first %d queries" % ns
        if args.cpu_sample > 0:
            from oracle import binding as ob  # the CPU restatement of the reference path (checker + baseline)
            ob.build(o3=True)
            orc = ob.Oracle(o3=True)
            cs = min(args.cpu_sample, nq)
            codes_h = np.empty((args.rows, M), dtype=np.uint8)
            _, _, codes_h = idx.get_entries()
            q_rot = orc.rotate_fma(R, q[:cs].cpu().numpy())
            t0 = time.perf_counter()
            od, oi = orc.adc_search(q_rot, books, codes_h, k)
            t_cpu = time.perf_counter() - t0
            same_ids = bool(np.array_equal(oi, i_gpu[:cs].cpu().numpy()))
            same_d = bool(np.array_equal(od.view(np.uint32), d_gpu[:cs].cpu().numpy().view(np.uint32)))
            result["cpu_baseline"] = {
                "value": round(cs / t_cpu, 2), "unit": "queries/s", "cores": 1, "kind": "port",
                "sample": "%d of the %d queries against the full %d-row code matrix, LUT + scan + top-%d "
                          "(oracle/cvt_oracle.c -O3, 1 thread); host: %s" % (cs, nq, args.rows, k, _cpu_model()),
                "gpu_topk_ids_identical": same_ids, "gpu_distances_bit_identical": same_d}
            result["recall_at_1_identical_to_cpu"] = bool(np.array_equal(oi[:, 0], i_gpu[:cs, 0].cpu().numpy()))
            # the same loop on all host cores: queries split over threads (ctypes releases the GIL), bounded sample
            nth = os.cpu_count() if args.cpu_threads < 0 else args.cpu_threads
            if nth and nth > 1:
                from concurrent.futures import ThreadPoolExecutor
                per = 8
                qs_mt = min(nq, nth * per)
                q_rot_mt = orc.rotate_fma(R, q[:qs_mt].cpu().numpy())
                chunks = [(a, min(qs_mt, a + per)) for a in range(0, qs_mt, per)]
                def work(ab):
                    return orc.adc_search(q_rot_mt[ab[0]:ab[1]], books, codes_h, k)
                with ThreadPoolExecutor(max_workers=nth) as ex:
                    t0 = time.perf_counter()
                    parts = list(ex.map(work, chunks))
                    t_mt = time.perf_counter() - t0
                oi_mt = np.concatenate([p[1] for p in parts])
                result["cpu_baseline_all_cores"] = {
                    "value": round(qs_mt / t_cpu_fix(t_mt), 2), "unit": "queries/s", "cores": nth, "kind": "port",
                    "sample": "%d queries in chunks of %d over %d threads, same loop as cpu_baseline" % (qs_mt, per, nth),
                    "gpu_topk_ids_identical": bool(np.array_equal(oi_mt, i_gpu[:qs_mt].cpu().numpy()))}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


def t_cpu_fix(t):
    return max(t, 1e-9)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip() + " (%d logical cores)" % os.cpu_count()
    except Exception:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
