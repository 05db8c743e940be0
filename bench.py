#!/usr/bin/env python3
"""bench.py -- queries/sec of the OPQ-ADC search hot path on MI355X.

One "step" = one pass of the hot path over one batch of nq synthetic queries against the HBM-resident code index:
query rotation (fp32 MFMA GEMM) -> per-query distance tables -> ADC scan of every code row -> k smallest
(distance, id) per query [-> N > 1: ONE RCCL all-gather of the per-shard top-k + k-way merge on every rank, issued by
libcvtmi itself (cvtmi_opq_search_sharded_dev, csrc/shard.hip)].  Inputs are resident in HBM when the timed region starts.

Headline workload:
  N = 1   BASELINE.json configs[1]: SIFT-1M (synthetic SIFT-shaped 128-d rows), OPQ M=16 K=256, top-100, nq = 10 000.
          The same line carries "sift1b": the N > 1 headline workload on this one GPU (the N = 1 point of its curve),
          and "secondary": rotation / encode / SQ8 / config 3 (10 M x 512-d uint8 flat search) / config 5 (HNSW).
  N > 1   BASELINE.json configs[3], the north-star multi-GPU case: SIFT-1B-shaped -- --large-rows (2^30) synthetic rows,
          generated, rotated and encoded on device, ROW-SHARDED over the N ranks (2 GB of codes per GPU at N = 8), the
          same nq = 10 000 queries on every rank, all-gather + merge: "scaling": "strong" (total work fixed).  A 16 MB
          database is not a multi-GPU workload; what N GPUs do with it (row-sharded, and as N replicas) is reported
          under "sift1m_row_sharded" / "sift1m_replicas", never as `value`.

    python bench.py                       # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0      # HBM3E
F32_MFMA_PEAK_TF = 157.3   # dense fp32 matrix
I8_MFMA_PEAK_TOPS = 5033.0  # dense int8 matrix (= the dense fp8 rate)
LDS_BYTES_PER_CLK_CU = 256.0
N_CU, CLK_GHZ = 256, 2.4


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
    ap.add_argument("--large-rows", type=int, default=1 << 30, help="total rows of the SIFT-1B-shaped workload (0 = skip at N = 1)")
    ap.add_argument("--large-nq", type=int, default=10_000)
    ap.add_argument("--large-data", choices=["sift", "random"], default="sift",
                    help="sift = synthetic rows rotated + encoded on device; random = uniform random code bytes (quick)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="gloo = debug: several ranks on one GPU, exchange staged through the host")
    ap.add_argument("--cpu-sample", type=int, default=256, help="queries timed on the CPU baseline (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=-1, help="threads of the all-cores CPU leg (-1 = all logical cores, 0 = skip)")
    ap.add_argument("--recall-sample", type=int, default=1000)
    ap.add_argument("--secondary", type=int, default=1, help="N = 1: also measure rotation / encode / SQ8 / config 3 / config 5 (0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import cvt_amd
    from cvt_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with --nproc-per-node equal to --gpus (got WORLD_SIZE=%d)" % world
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path"
    if args.backend == "gloo":
        local_rank = 0  # debug: every rank on GPU 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cvt_amd.lib()
    comm = None
    if world > 1:
        # torch.distributed: launcher glue (barriers, max-over-ranks clock, bootstrap of the id).  The data-path collective
        # is the library's own ncclAllGather on its own communicator.
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
            box = [cvt_amd.Comm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            comm = cvt_amd.Comm(box[0], rank, world)
        else:
            dist.init_process_group("gloo")
            comm = cvt_amd.Comm.over_torch_group(rank, world)

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

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps, warmup):
        out = None
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

    # ---- index build on device: generate -> rotate (MFMA GEMM) -> encode -> append; only codes stay ----
    def build_index(r0, r1, data="sift", seed=0xC0FFEE):
        ix = cvt_amd.OpqIndex(zero_coarse, books, R=R)
        ix.reserve(r1 - r0); ix.set_id_base(r0)
        rows_done, t_acc = 0, 0.0
        if data == "random":
            g = torch.Generator(device=dev); g.manual_seed(0x51F7 + rank)
            for a in range(r0, r1, 1 << 24):
                b = min(r1, a + (1 << 24))
                ix.add_codes(torch.randint(0, 256, (b - a, M), generator=g, device=dev, dtype=torch.uint8))
        else:
            step = synth.CHUNK * 4
            for a in range(r0, r1, step):
                b = min(r1, a + step)
                x = synth.sift_like(b - a, D, seed=seed, row_begin=a, device=dev)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                _, codes = ix.rotate_encode(x)
                ix.add_codes(codes)
                torch.cuda.synchronize(); t_acc += time.perf_counter() - t0  # data generation excluded
                rows_done += b - a
        ix.set_param("qtile", args.qtile); ix.set_param("splits", args.splits); ix.set_param("profile", 1)
        if args.variant >= 0:
            ix.set_param("scan_variant", args.variant)
        return ix, rows_done, t_acc

    def searcher(ix):
        if comm is not None:
            return lambda qq: ix.search_sharded(comm, qq, k, rotate=True)
        return lambda qq: ix.search(qq, k, rotate=True)

    def scan_roofline(sc):
        ach = sc["code_bytes"] / (sc["ms"] * 1e-3) / 1e9
        return {"kernel_ms": round(sc["ms"], 4), "algorithmic_bytes_per_launch": sc["code_bytes"], "achieved_GBps": round(ach, 1),
                "frac_of_hbm_peak": round(ach / HBM_PEAK_GBS, 4), "queries_per_pass": sc["qtile"], "row_splits": sc["splits"]}

    q = synth.sift_like(nq, D, seed=0xBEEF, device=dev)  # identical on every rank

    # ---- the SIFT-1B-shaped workload, row-sharded over the ranks (N > 1: the headline; N = 1: the first point of its curve) ----
    def run_sift1b(steps, warmup):
        l0, l1 = cvt_amd.shard_range(args.large_rows, rank, world)
        free_b, _ = torch.cuda.mem_get_info(dev)
        # codes + the scan's rotated copy + generation chunks + tables must fit; every rank takes the same decision
        fits = torch.tensor([1 if free_b > (l1 - l0) * M * 2.3 + (3 << 30) else 0], dtype=torch.int32,
                            device=dev if args.backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(fits, op=dist.ReduceOp.MIN)
        if int(fits.item()) == 0:
            return {"error": "not enough free HBM for %d code rows per GPU" % (l1 - l0)}, None
        t0 = time.perf_counter()
        big, enc_rows, enc_t = build_index(l0, l1, data=args.large_data, seed=0xC0FFEE)
        torch.cuda.synchronize(); t_build = time.perf_counter() - t0
        ql = q[:min(args.large_nq, nq)].contiguous()
        fn = searcher(big)
        for _ in range(warmup):
            fn(ql)
        barrier(); big.last_scan()
        el, out = timed(lambda: fn(ql), steps, 0)
        sc = big.last_scan()
        res = {"value": round(ql.shape[0] * steps / el, 1), "unit": "queries/s", "ms_per_step": round(el / steps * 1e3, 4), "steps": steps,
               "rows_total": args.large_rows, "rows_per_gpu": l1 - l0, "nq": int(ql.shape[0]), "k": k, "n_gpus": world,
               "scan": scan_roofline(sc),
               "index_build_s_this_rank": round(t_build, 2),
               "encode_rows_per_s_this_rank": round(enc_rows / enc_t, 1) if enc_t > 0 else None,
               "data": ("synthetic SIFT-shaped rows generated, rotated and PQ-encoded on device" if args.large_data == "sift"
                        else "uniform random code bytes"),
               "what": "%d rows row-sharded x%d (%.2f GB of codes per GPU), the same %d queries on every rank, top-%d, %s" % (
                   args.large_rows, world, (l1 - l0) * M / 1e9, ql.shape[0], k,
                   "ONE ncclAllGather of the per-shard top-k inside libcvtmi + merge" if world > 1 else "single shard")}
        if comm is not None:
            res["comm"] = comm.info()
        big.close()
        return res, sc

    result, extra = None, {}
    if world == 1:
        # ================= N = 1: SIFT-1M, configs[1] =================
        idx, enc_rows, enc_time = build_index(0, args.rows)
        fn = searcher(idx)
        for _ in range(args.warmup):
            fn(q)
        barrier(); idx.last_scan()  # drop the warm-up launches from the kernel-time statistics
        elapsed, out = timed(lambda: fn(q), args.steps, 0)
        scan = idx.last_scan()  # mean HIP-event duration of the scan kernel over the timed steps
        ms_per_step = elapsed / args.steps * 1e3
        rf = scan_roofline(scan)
        lookups = scan["code_bytes"] * scan["qtile"] / (scan["ms"] * 1e-3)  # one table look-up per code byte and query
        lds_bytes = lookups * 2.0                                          # 2 bytes of LDS read per look-up (15-bit tables, 8 queries per 16-byte read)
        lds_peak = LDS_BYTES_PER_CLK_CU * N_CU * CLK_GHZ * 1e9
        traffic, traffic_src = _pmc_traffic(args, nq, k, M)
        result = {
            "metric": "queries/sec, OPQ-ADC top-%d over 128-d %s" % (k, "SIFT-1M" if args.rows == 1_000_000 else "%d synthetic rows" % args.rows),
            "value": round(nq * args.steps / elapsed, 1), "unit": "queries/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 codes / f32 distances", "data": "synthetic",
            "config": {"workload": "%s synthetic 128-d, OPQ M=%d K=256 (dense 128x128 rotation), ADC scan + top-%d, "
                                   "nq=%d queries per step" % ("SIFT-1M" if args.rows == 1_000_000 else "%d-row" % args.rows, M, k, nq),
                       "rows": args.rows, "rows_per_gpu": args.rows, "nq_per_step": nq, "k": k, "M": M, "parallelism": "1 GPU",
                       "qtile": scan["qtile"], "row_splits": scan["splits"]},
            "roofline": {"bound": "hbm", "kernel": "adc_scan kernel (M=%d, %d queries per pass)" % (M, scan["qtile"]),
                         "achieved": rf["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rf["frac_of_hbm_peak"],
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": scan["code_bytes"], "kernel_ms": rf["kernel_ms"],
                         "operative_bound": "lds+valu",
                         "operative_note": "the kernel is bound by its LDS table look-ups at every size: the query groups of a row split share each "
                                           "row chunk through L2 / Infinity Cache, so HBM-side traffic (PMC) is a fraction of the algorithmic bytes "
                                           "(0.10 at SIFT-1M, 0.03 at a 128 M-row shard: profiles/r02_scan_traffic*.json); `achieved` is the "
                                           "algorithmic rate SURVEY 8(d) prescribes",
                         "lds_lookups_per_s": round(lookups / 1e12, 2),
                         "lds_frac": round(lds_bytes / lds_peak, 4),
                         "lds_frac_what": "table look-ups/s x 2 B per look-up / (256 B/clk/CU x 256 CU x 2.4 GHz)"},
            "encode": {"rows_per_s": round(enc_rows / enc_time, 1), "what": "rotate (MFMA GEMM) + PQ encode (cvtmi_opq_rotate_encode) + append of the 1 M rows"},
        }
        if args.large_rows > 0:
            try:
                res, _ = run_sift1b(max(1, min(args.steps, 2)), 1)
                result["sift1b"] = res
            except cvt_amd.CvtmiError as e:
                result["sift1b"] = {"error": str(e)}
    else:
        # ================= N > 1: SIFT-1B-shaped, row-sharded, configs[3] =================
        res, sc = run_sift1b(args.steps, args.warmup)
        if "error" in res:
            raise SystemExit("bench.py: " + res["error"])
        # the small database on N GPUs, for the record: row-sharded through the same library path, and as N replicas
        r0, r1 = cvt_amd.shard_range(args.rows, rank, world)
        shard, _, _ = build_index(r0, r1)
        fn = searcher(shard)
        el2, _ = timed(lambda: fn(q), args.steps, args.warmup)
        extra["sift1m_row_sharded"] = {"value": round(nq * args.steps / el2, 1), "unit": "queries/s", "ms_per_step": round(el2 / args.steps * 1e3, 4),
                                       "what": "the 1 M-row database row-sharded x%d (%d rows per GPU), all %d queries on every rank, all-gather + merge: "
                                               "strong scaling of a 16 MB problem" % (world, r1 - r0, nq)}
        shard.close()
        rep, _, _ = build_index(0, args.rows)
        q_mine = synth.sift_like(nq, D, seed=0xBEEF + 7919 * rank, device=dev) if rank else q
        el3, _ = timed(lambda: rep.search(q_mine, k, rotate=True), args.steps, args.warmup)
        extra["sift1m_replicas"] = {"value": round(world * nq * args.steps / el3, 1), "unit": "queries/s", "ms_per_step": round(el3 / args.steps * 1e3, 4),
                                    "what": "code matrix replicated x%d, every rank serves its own batch of %d queries, no collective: "
                                            "N independent replicas (Nx by construction)" % (world, nq)}
        rep.close()
        if rank == 0:
            rf = res["scan"]
            result = {
                "metric": "queries/sec, OPQ-ADC top-%d over 128-d SIFT-1B-shaped rows, row-sharded x%d" % (k, world),
                "value": res["value"], "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "u8 codes / f32 distances", "data": "synthetic",
                "config": {"workload": "SIFT-1B-shaped: %d synthetic 128-d rows, OPQ M=%d K=256, row-sharded over %d GPUs, ADC scan + top-%d, nq=%d "
                                       "queries per step (the same batch on every rank)" % (args.large_rows, M, world, k, res["nq"]),
                           "rows": args.large_rows, "rows_per_gpu": res["rows_per_gpu"], "nq_per_step": res["nq"], "k": k, "M": M,
                           "parallelism": "row-sharded x%d, ONE ncclAllGather (RCCL, issued inside libcvtmi) of per-shard top-%d + merge on every rank" % (world, k),
                           "n1_point_of_this_curve": "the N = 1 line's \"sift1b\".value (same rows, same queries, one GPU)"},
                "roofline": {"bound": "hbm", "kernel": "adc_scan kernel (M=%d, %d queries per pass), rank 0's shard" % (M, rf["queries_per_pass"]),
                             "achieved": rf["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rf["frac_of_hbm_peak"], "traffic": None,
                             "algorithmic_bytes_per_launch": rf["algorithmic_bytes_per_launch"], "kernel_ms": rf["kernel_ms"]},
                "sift1b": res,
            }
            result.update(extra)

    # ---- outside the timed region: recall@1, the CPU baseline and the secondary kernels (rank 0, N = 1 only) ----
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
                    "value": round(qs_mt / max(t_mt, 1e-9), 2), "unit": "queries/s", "cores": nth, "kind": "port",
                    "sample": "%d queries in chunks of %d over %d threads, same loop as cpu_baseline" % (qs_mt, per, nth),
                    "gpu_topk_ids_identical": bool(np.array_equal(oi_mt, i_gpu[:qs_mt].cpu().numpy()))}
        idx.close()
        if args.secondary:
            try:
                result["secondary"] = secondary(args, dev, books, R)
            except Exception as e:  # the headline line must survive a failing side measurement
                result["secondary"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()


def _pmc_traffic(args, nq, k, M):
    """HBM bytes per scan launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs, of this
    same command; tools/profile_round.sh).  Counters cannot be collected inside the timed run, so the JSON says where the
    number comes from; null when the launch shape differs from the profiled one."""
    default_shape = args.rows == 1_000_000 and nq == 10_000 and k == 100 and M == 16 and not args.qtile and not args.splits and args.variant < 0
    if not default_shape:
        return None, None
    for tag in ("r02", "r01"):
        p = os.path.join(ROOT, "profiles", "%s_scan_traffic.json" % tag)
        if os.path.exists(p):
            try:
                d = json.load(open(p))
                return d.get("hbm_bytes_per_launch"), "profiles/%s_scan_traffic.json: %s; kernel %s" % (
                    tag, d.get("note", "rocprofv3 --pmc passes"), d.get("kernel", "?"))
            except Exception:
                pass
    return None, None


def _ev_ms(torch, fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()  # the library launches on torch's current stream (capi._stream), so torch events bracket its kernels
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def secondary(args, dev, books, R):
    """The other rows of SURVEY.md 8 / BASELINE configs, one short measurement each (N = 1, outside the timed region)."""
    import torch
    import cvt_amd
    from cvt_amd import synth
    from concurrent.futures import ThreadPoolExecutor
    sec = {}
    D, M = 128, books.shape[0]
    zero = np.zeros((1, D), np.float32)
    # ---- a-R rotation (fp32 MFMA GEMM) and a-E encode, 2^20 rows ----
    n = 1 << 20
    ix = cvt_amd.OpqIndex(zero, books, R=R)
    x = synth.sift_like(n, D, seed=0xC0FFEE, device=dev)
    ms = _ev_ms(torch, lambda: ix.rotate(x))
    tf = 2.0 * n * D * D / (ms * 1e-3) / 1e12
    sec["rotation"] = {"rows": n, "ms": round(ms, 4), "tflops": round(tf, 1), "frac_of_f32_mfma_peak": round(tf / F32_MFMA_PEAK_TF, 4),
                       "hbm_GBps": round(n * D * 8 / (ms * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(n * D * 8 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                       "what": "Y = X R^T, dense 128 x 128 fp32 R on v_mfma_f32_32x32x2_f32 (2 D^2 flop and 1 KB moved per row)"}
    xr = ix.rotate(x)
    ms = _ev_ms(torch, lambda: ix.encode(xr))
    sec["encode"] = {"rows": n, "ms": round(ms, 4), "rows_per_s": round(n / (ms * 1e-3), 1), "what": "PQ encode of rotated rows, M=%d K=256 (bf16 matrix-core filter + exact chain)" % M}
    try:
        ms = _ev_ms(torch, lambda: ix.rotate_encode(x))
        sec["rotate_encode"] = {"rows": n, "ms": round(ms, 4), "rows_per_s": round(n / (ms * 1e-3), 1),
                                      "what": "raw rows -> codes in one call: rotation and encode chunk by chunk through a cache-resident scratch (cvtmi_opq_rotate_encode)"}
    except AttributeError:
        pass
    # the reference's own rotation is a permutation of the dimensions (reorder_, IVFOPQ.cpp:424-439): the encode kernel gathers through it
    ixp = cvt_amd.OpqIndex(zero, books, perm=synth.random_permutation(D, seed=5))
    ms_two = _ev_ms(torch, lambda: ixp.encode(ixp.rotate(x)))
    ms = _ev_ms(torch, lambda: ixp.rotate_encode(x))
    sec["reorder_encode"] = {"rows": n, "ms": round(ms, 4), "rows_per_s": round(n / (ms * 1e-3), 1),
                             "rows_per_s_permute_then_encode": round(n / (ms_two * 1e-3), 1),
                             "what": "raw rows -> codes for a permutation model (the reference's IVFOPQ::reorder + Add): one kernel, the rows are read through the permutation"}
    ixp.close()
    ix.close(); del x, xr
    # ---- a-Q / a-T SQ8: train + encode, 2 M x 512-d ----
    d3 = 512
    g = torch.Generator(device=dev); g.manual_seed(3)
    feats = torch.randn((1 << 21, d3), generator=g, device=dev).clamp_(min=0)  # "CNN-like": ReLU'd Gaussian
    nb = feats.numel() * 4
    ms_t = _ev_ms(torch, lambda: cvt_amd.sq8_train(feats, l2norm=True), reps=3, warm=1)
    vmin, vdiff = cvt_amd.sq8_train(feats, l2norm=True)
    ms_e = _ev_ms(torch, lambda: cvt_amd.sq8_encode(vmin, vdiff, feats.clone(), l2norm=False), reps=3, warm=1)
    ms_c = _ev_ms(torch, lambda: feats.clone(), reps=3, warm=1)
    ms_e = max(ms_e - ms_c, 1e-3)
    sec["sq8"] = {"rows": 1 << 21, "d": d3,
                  "train_l2norm_GBps": round(nb / (ms_t * 1e-3) / 1e9, 1), "train_frac_of_hbm_peak": round(nb / (ms_t * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                  "encode_GBps": round(nb * 1.25 / (ms_e * 1e-3) / 1e9, 1), "encode_frac_of_hbm_peak": round(nb * 1.25 / (ms_e * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                  "what": "per-dimension min / max-min over L2-normalised rows (4d bytes per row); encode 4d in + d out per row"}
    del feats
    # ---- config 3: 10 M x 512-d uint8 codes (SQ8 of CNN-like features), brute-force L2 top-10 ----
    n3, k3 = 10_000_000, 10
    flat = cvt_amd.FlatIndex(cvt_amd.L2U8, d3)
    chunk = 1 << 20
    rows_h = np.empty((n3, d3), dtype=np.uint8) if args.cpu_sample > 0 else None  # host copy for the CPU baseline only
    for a in range(0, n3, chunk):
        b = min(n3, a + chunk)
        g.manual_seed(1000 + a // chunk)
        f = torch.randn((b - a, d3), generator=g, device=dev).clamp_(min=0)
        c = cvt_amd.sq8_encode(vmin, vdiff, f, l2norm=True)
        flat.add(c)
        if rows_h is not None:
            rows_h[a:b] = c.cpu().numpy()
    g.manual_seed(77)
    qf = torch.randn((4096, d3), generator=g, device=dev).clamp_(min=0)
    q3 = cvt_amd.sq8_encode(vmin, vdiff, qf, l2norm=True)
    c3 = {"rows": n3, "d": d3, "k": k3, "cases": {}}
    outs = {}
    for nq3 in (1, 8, 64, 1000, 4096):
        qq = q3[:nq3].contiguous()
        ms = _ev_ms(torch, lambda: flat.search(qq, k3), reps=3, warm=1)
        outs[nq3] = flat.search(qq, k3)
        ops = 2.0 * nq3 * n3 * d3 / (ms * 1e-3)
        c = {"ms": round(ms, 4), "queries_per_s": round(nq3 / (ms * 1e-3), 1)}
        if nq3 <= 128:   # one stream over the rows (flat_u8_mstream_kernel): bound by HBM
            gb = n3 * d3 / (ms * 1e-3) / 1e9
            c["roofline"] = {"bound": "hbm", "achieved": round(gb, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gb / HBM_PEAK_GBS, 4)}
        else:
            c["roofline"] = {"bound": "mfma", "achieved": round(ops / 1e12, 1), "peak": I8_MFMA_PEAK_TOPS, "unit": "TOP/s (int8, 2 per MAC)",
                             "frac": round(ops / 1e12 / I8_MFMA_PEAK_TOPS, 4)}
        c3["cases"]["nq=%d" % nq3] = c
    if args.cpu_sample > 0:
        from oracle import binding as ob
        orc = ob.Oracle(o3=True)
        cs = 4
        qh = q3[:cs].cpu().numpy()
        t0 = time.perf_counter()
        _, od, oi = orc.flat_search(ob.L2U8, rows_h, qh, k3)
        t_cpu = time.perf_counter() - t0
        gd, gi = outs[1000]
        c3["cpu_baseline"] = {"value": round(cs / t_cpu, 3), "unit": "queries/s", "cores": 1, "kind": "port",
                              "sample": "%d queries against all %d rows (oracle L2SqrI loop, -O3, 1 thread)" % (cs, n3),
                              "gpu_topk_ids_identical": bool(np.array_equal(oi, gi[:cs].cpu().numpy())),
                              "gpu_distances_identical": bool(np.array_equal(np.asarray(od).astype(np.int64), gd[:cs].cpu().numpy().astype(np.int64)))}
        del rows_h
    flat.close()
    sec["flat_u8_c3"] = c3
    # ---- f-4: the reference's own Query shape -- coarseK = 8192 lists, nk = 3 probes, per-video minimum scores ----
    try:
        sec["ivf_query"] = _ivf_query(dev, torch, cvt_amd, books)
    except Exception as e:
        sec["ivf_query"] = {"error": "%s: %s" % (type(e).__name__, e)}
    # ---- config 5: HNSW graph in HBM, batched 10 K queries, fp32 vectors and OPQ codes ----
    try:
        sec["hnsw_c5"] = _hnsw_c5(args, dev, torch, cvt_amd, synth, R)
    except Exception as e:
        sec["hnsw_c5"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return sec


def _ivf_query(dev, torch, cvt_amd, books):
    D, L, nk, n, n_videos, nq = 128, 8192, 3, 1 << 20, 4096, 10_000
    g = torch.Generator(device=dev); g.manual_seed(11)
    cen = torch.randn((L, D), generator=g, device=dev) * 0.08
    x = cen[torch.randint(0, L, (n,), generator=g, device=dev)] + 0.03 * torch.randn((n, D), generator=g, device=dev)
    q = x[torch.randint(0, n, (nq,), generator=g, device=dev)] + 0.01 * torch.randn((nq, D), generator=g, device=dev)
    ix = cvt_amd.OpqIndex(cen.cpu().numpy(), (books * 0.3).astype(np.float32))
    ms_enc = _ev_ms(torch, lambda: ix.encode(x), reps=2, warm=1)
    lists, codes = ix.encode(x)
    ix.add_codes(codes, lists, torch.randint(0, n_videos, (n,), generator=g, device=dev, dtype=torch.int32))
    ms_first = _ev_ms(torch, lambda: ix.query_video(q[:64].contiguous(), nk, n_videos, rotate=False), reps=1, warm=0)  # builds the list-ordered copy
    ms_q = _ev_ms(torch, lambda: ix.query_video(q, nk, n_videos, rotate=False), reps=3, warm=1)
    ms_9 = _ev_ms(torch, lambda: ix.query_video(q[:9].contiguous(), nk, n_videos, rotate=False), reps=5, warm=1)
    ix.close()
    return {"entries": n, "lists": L, "nprobe": nk, "videos": n_videos,
            "encode_rows_per_s": round(n / (ms_enc * 1e-3), 1), "encode_what": "coarse argmin over 8192 centroids (matrix-core filter + exact resolution) + PQ encode",
            "first_query_ms_incl_list_build": round(ms_first, 3),
            "frames": nq, "ms": round(ms_q, 3), "frames_per_s": round(nq / (ms_q * 1e-3), 1),
            "ms_9_frames": round(ms_9, 3),
            "what": "IVFOPQ::Query semantics (IVFOPQ.cpp:213-320): coarse top-3 of 8192, residual tables, list scans, per-video min clamped at 1.0; "
                    "dense [frames][videos] score matrix out"}


def _hnsw_c5(args, dev, torch, cvt_amd, synth, R):
    n, D, nq, Mg, efc = 100_000, 128, 10_000, 32, 80
    rng = np.random.default_rng(5)
    cen = rng.normal(size=(1000, D)).astype(np.float32)
    x = cen[rng.integers(0, 1000, n)] + 0.6 * rng.normal(size=(n, D)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    q = x[rng.integers(0, n, nq)] + 0.15 * rng.normal(size=(nq, D)).astype(np.float32)
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    tmpd = tempfile.mkdtemp()
    rows_p, idx_p = os.path.join(tmpd, "rows.bin"), os.path.join(tmpd, "graph.hnsw")
    x.astype(np.float32).tofile(rows_p)
    t0 = time.perf_counter()
    subprocess.run([os.path.join(ROOT, "cvt_amd", "bin", "hnsw_build"), rows_p, str(D), str(Mg), str(efc), idx_p, "ip"], check=True,
                   capture_output=True)
    t_build = time.perf_counter() - t0
    blob = open(idx_p, "rb").read()
    ix = cvt_amd.HnswIndex(blob, cvt_amd.IP, D)
    qd = torch.from_numpy(q).to(dev)
    exact = torch.argmax(qd @ torch.from_numpy(x).to(dev).T, dim=1)
    res = {"nodes": n, "d": D, "M": Mg, "ef_construction": efc, "nq": nq, "graph_build_s_host": round(t_build, 1),
           "what": "one wave per query over a graph built on the host (hnsw_build CLI = the reference's addPoint order, M=32 efC=80, makeIdx.cpp:303-304)",
           "fp32": {}, "adc": {}}
    for kk, ef in ((5, 1000), (10, 64)):
        ms = _ev_ms(torch, lambda: ix.search(qd, kk, ef), reps=2, warm=1)
        _, lab = ix.search(qd, kk, ef)
        res["fp32"]["ef=%d" % ef] = {"ms": round(ms, 3), "queries_per_s": round(nq / (ms * 1e-3), 1), "k": kk,
                                     "recall_at_1": round(float((lab[:, 0] == exact).float().mean().item()), 4)}
    tmp = cvt_amd.OpqIndex(np.zeros((1, D), np.float32), np.zeros((16, 256, D // 16), np.float32), R=R)
    xr = tmp.rotate(torch.from_numpy(x).to(dev))
    _, books5 = cvt_amd.opq_train(xr[:50_000].contiguous(), 1, 16, 256, 8, 1)
    opq = cvt_amd.OpqIndex(np.zeros((1, D), np.float32), books5.cpu().numpy(), R=R)
    _, codes = opq.encode(xr)
    opq.add_codes(codes)
    for kk, ef in ((5, 1000), (10, 64)):
        ms = _ev_ms(torch, lambda: ix.search_adc(opq, qd, kk, ef), reps=2, warm=1)
        _, lab = ix.search_adc(opq, qd, kk, ef)
        res["adc"]["ef=%d" % ef] = {"ms": round(ms, 3), "queries_per_s": round(nq / (ms * 1e-3), 1), "k": kk,
                                    "recall_at_1": round(float((lab[:, 0] == exact).float().mean().item()), 4),
                                    "recall_at_k": round(float((lab == exact[:, None]).any(dim=1).float().mean().item()), 4)}
        rr = getattr(ix, "search_adc_rerank", None)
        if rr is not None:
            ms = _ev_ms(torch, lambda: rr(opq, qd, kk, ef), reps=2, warm=1)
            _, lab = rr(opq, qd, kk, ef)
            res["adc"]["ef=%d+rerank" % ef] = {"ms": round(ms, 3), "queries_per_s": round(nq / (ms * 1e-3), 1), "k": kk,
                                               "recall_at_1": round(float((lab[:, 0] == exact).float().mean().item()), 4)}
    if args.cpu_sample > 0:
        from oracle import binding as ob
        if ob.ref_available():
            rh = ob.RefHnsw()
            cs = 200
            t0 = time.perf_counter(); rd, rl = rh.search(0, D, idx_p, q[:cs], 5, 1000); t_cpu = time.perf_counter() - t0
            gd, gl = ix.search(qd[:cs].contiguous(), 5, 1000)
            res["cpu_baseline"] = {"value": round(cs / t_cpu, 1), "unit": "queries/s", "cores": 1, "kind": "reference",
                                   "sample": "%d of the queries, the reference's own searchKnn (oracle/_ref/libref_hnsw.so), ef=1000, same graph file" % cs,
                                   "gpu_labels_identical": bool(np.array_equal(rl, gl.cpu().numpy())),
                                   "gpu_distances_bit_identical": bool(np.array_equal(rd.view(np.uint32), gd.cpu().numpy().view(np.uint32)))}
    ix.close(); opq.close(); tmp.close()
    for p in (rows_p, idx_p):
        if os.path.exists(p):
            os.remove(p)
    os.rmdir(tmpd)
    return res


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
