#!/usr/bin/env python3
"""bench.py -- queries/sec of the OPQ-ADC search hot path on MI355X.

One "step" = one pass of the hot path over one batch of nq synthetic queries against the HBM-resident code index:
query rotation (fp32 MFMA GEMM) -> per-query distance tables -> ADC scan of every code row -> k smallest
(distance, id) per query [-> N > 1: ONE RCCL all-gather of the per-shard top-k + k-way merge on every rank, issued
by libcvtmi itself (cvtmi_opq_search_sharded_dev, csrc/shard.hip)].  Inputs are resident in HBM when the timed
region starts.

Headline workload:
  N = 1   BASELINE.json configs[1]: SIFT-1M (synthetic SIFT-shaped 128-d rows), OPQ M=16 K=256, top-100, nq = 10 000.
          The same line carries "sift1b": the N > 1 headline workload on this one GPU (the N = 1 point of its curve),
          and "secondary": rotation / encode / SQ8 / fp32 flat search / config 3 (10 M x 512-d uint8 flat search) /
          the IVF query shape / config 5 (HNSW).
  N > 1   BASELINE.json configs[3], the north-star multi-GPU case: SIFT-1B-shaped -- --large-rows (2^30) synthetic
          rows, generated, rotated and encoded on device, ROW-SHARDED over the N ranks (2 GB of codes per GPU at
          N = 8), the same nq = 10 000 queries on every rank, all-gather + merge: "scaling": "strong" (total work
          fixed).  A 16 MB database is not a multi-GPU workload; what N GPUs do with it (row-sharded, and as N
          replicas) is reported under "sift1m_row_sharded" / "sift1m_replicas", never as `value`.

Ranks find each other over cvt_amd/rendezvous.py (plain TCP around the launcher's RANK / WORLD_SIZE / MASTER_*):
barriers, the max-over-ranks clock and the hand-over of the communicator id need no torch.distributed, and every rank reports
"index built, ready" BEFORE anyone enters the blocking communicator creation.  torch is used for device memory only.

    python bench.py                       # 1 GPU
    python bench.py --gpus N              # no launcher: bench.py starts its N ranks itself (self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

The N > 1 line proves what it measured: `rccl_ranks` / `collectives_per_search` from the library's own counters
(cvtmi_comm_info), `identical_to_oracle_sample` (every rank scans ITS shard with the CPU oracle for a few queries, rank 0 merges
the per-shard lists with the oracle's merge and compares ids and distance bits with the GPUs' merged answer -- the shape of
FLANN-MPI's local search + offset + reduce, retrieval/vlindex/lib/FLANN/mpi/index.h:196-226, :74-108), `recall_at_1` and a
`cpu_baseline` from that same oracle run.  If the RCCL communicator cannot be created (fewer devices than ranks, creation
fails or times out) the line still appears, with "error" and the result of the same library path over the host transport.
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
HBM_PEAK_GBS = 8000.0           # HBM3E (spec; a float4 copy reaches 6.29 TB/s)
F32_MFMA_PEAK_TF = 157.3        # dense fp32 matrix
BF16_MFMA_PEAK_TF = 2500.0      # dense bf16 matrix
I8_MFMA_PEAK_TOPS = 5033.0      # dense int8 matrix, spec (= the dense fp8 rate)
I8_MFMA_MEASURED_TOPS = 3944.0  # the guide's measured ceiling of the int8 matrix instruction
LDS_BYTES_PER_CLK_CU = 256.0
N_CU, CLK_GHZ = 256, 2.4
LDS_PEAK_TBS = LDS_BYTES_PER_CLK_CU * N_CU * CLK_GHZ / 1e3   # 157.3 TB/s: ds_read_b128 at the nominal clock, every CU
LDS_MEASURED_TBS = 150.0        # the guide's aggregate with every CU streaming ds_read_b64/b128
D, K = 128, 256


def parse_args():
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
    ap.add_argument("--large-rows", type=int, default=1 << 30,
                    help="total rows of the SIFT-1B-shaped workload (0 = skip at N = 1)")
    ap.add_argument("--large-nq", type=int, default=10_000)
    ap.add_argument("--large-data", choices=["sift", "random"], default="sift",
                    help="sift = synthetic rows rotated + encoded on device; random = uniform random code bytes "
                         "(quick)")
    ap.add_argument("--backend", choices=["nccl", "host", "gloo"], default="nccl",
                    help="host (alias gloo) = debug: several ranks on ONE GPU, exchange staged through the host")
    ap.add_argument("--comm-timeout", type=float, default=180.0,
                    help="seconds the RCCL communicator creation may take before the ranks fall back to the host transport")
    ap.add_argument("--n1-point", type=int, default=1,
                    help="N > 1: rank 0 also runs the SAME workload alone on its GPU afterwards (all rows, same queries, no collective) -> "
                         "'same_workload_on_one_gpu' and 'speedup_over_one_gpu' in the line (0 = skip)")
    ap.add_argument("--oracle-queries", type=int, default=8,
                    help="N > 1: queries every rank scans over its own shard with the CPU oracle (identity check + cpu_baseline; 0 = skip)")
    ap.add_argument("--cpu-sample", type=int, default=256, help="queries timed on the CPU baseline (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=-1,
                    help="threads of the all-cores CPU leg (-1 = all logical cores, 0 = skip)")
    ap.add_argument("--recall-sample", type=int, default=1000)
    ap.add_argument("--ref-rows", type=int, default=1_000_000,
                    help="rows of the index the reference itself builds for cpu_baseline: the headline's own size by default, ~12 s of its Add (0 = skip)")
    ap.add_argument("--ref-queries", type=int, default=1024, help="queries of the reference leg (~4 s at 1 M rows)")
    ap.add_argument("--hnsw-nodes", type=int, default=1_000_000, help="nodes of the config-5 graph (secondary.hnsw_c5)")
    ap.add_argument("--only", default="", help="development: comma list of secondary sections to run (rotation_encode, opq_m8, opq_rotation_learning, "
                                               "sq8, flat_f32, flat_u8_c3, ivf_query, hnsw_c5); default all")
    ap.add_argument("--host-api", type=int, default=1, help="0 = skip the host-pointer leg (its 4096-query pieces are scan launches too)")
    ap.add_argument("--tune", default="", help="development: cvtmi_set_tuning pairs, name=value,name=value")
    ap.add_argument("--secondary", type=int, default=1,
                    help="N = 1: also measure rotation / encode / SQ8 / flat searches / config 5 (0 = skip)")
    return ap.parse_args()


class Ctx:
    """what every measurement needs: arguments, device, rank / world, the rendezvous, the model"""

    def __init__(self, args):
        import torch
        import cvt_amd
        from cvt_amd import synth
        from cvt_amd.rendezvous import Rendezvous
        self.args, self.torch, self.cvt, self.synth = args, torch, cvt_amd, synth
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        assert self.world == args.gpus, "launch with --nproc-per-node equal to --gpus (got WORLD_SIZE=%d)" % self.world
        assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path"
        ndev = torch.cuda.device_count()
        self.host_transport = args.backend != "nccl"
        self.error = None          # why the N > 1 line is not the RCCL measurement it was asked for (it is printed either way)
        self.hung_thread = False   # a communicator creation that never returned: the process leaves through os._exit
        if not self.host_transport and ndev < self.world:
            # RCCL refuses two ranks on one device: every rank sees the same device count and takes the same decision
            self.error = ("%d ranks asked for, %d device(s) visible: RCCL needs one device per rank; the figures are the same library "
                          "path (local scan -> all-gather -> merge) over the host transport, several ranks per device" % (self.world, ndev))
            self.host_transport = True
        if self.host_transport:
            local_rank = local_rank % max(ndev, 1)   # debug / fall-back: ranks share the visible devices
        torch.cuda.set_device(local_rank)
        self.dev = torch.device("cuda", local_rank)
        cvt_amd.lib()
        self.rv = Rendezvous(self.rank, self.world)
        self.comm = None
        self.M, self.k, self.nq = args.M, args.k, args.nq
        self.zero_coarse = np.zeros((1, D), np.float32)
        self.R = synth.random_rotation(D, seed=7)
        self.books = None
        self.nn = None
        self.last_out = None
        self.exact_nn = None
        self.search_error = None

    def barrier(self):
        self.rv.barrier()
        self.torch.cuda.synchronize()

    def timed(self, fn, steps, warmup):
        """warm-up, then exactly `steps` calls between two (barrier + synchronize), MAX over ranks"""
        out = None
        for _ in range(warmup):
            out = fn()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = fn()
        self.barrier()
        return self.rv.max(time.perf_counter() - t0), out

    def _create_rccl(self):
        """ncclCommInitRank inside the library, on a helper thread so that a creation that never returns (a peer died on its
        way in) costs --comm-timeout seconds instead of the job; returns None or the reason it did not happen"""
        import threading
        msg = None
        if self.rank == 0:
            try:
                msg = (True, self.cvt.Comm.unique_id())
            except Exception as e:   # RCCL not loadable: the others are waiting for this broadcast
                msg = (False, "%s: %s" % (type(e).__name__, e))
        ok, uid = self.rv.bcast(msg)
        if not ok:
            return "rank 0 could not get a communicator id: %s" % uid
        box = {}

        def work():
            try:
                self.torch.cuda.set_device(self.dev)   # the current device is per-thread state
                box["comm"] = self.cvt.Comm(uid, self.rank, self.world)
            except Exception as e:
                box["err"] = "%s: %s" % (type(e).__name__, e)

        t = threading.Thread(target=work, daemon=True)
        t.start(); t.join(self.args.comm_timeout)
        if t.is_alive():
            self.hung_thread = True
            return "rank %d: communicator creation did not return within %g s" % (self.rank, self.args.comm_timeout)
        if "err" in box:
            return "rank %d: %s" % (self.rank, box["err"])
        self.comm = box["comm"]
        return None

    def make_comm(self, ready, what=""):
        """the library's communicator -- only after EVERY rank has said it is ready (the creation is a blocking collective).
        RCCL first; when any rank cannot create it, every rank falls back to the host transport (the same library path with the
        all-gather staged through the rendezvous) and the JSON line carries "error"."""
        ok, bad = self.rv.all_ok(ready, what)
        if not ok:
            raise SystemExit("bench.py: rank(s) not ready, nobody creates the communicator: %s" % bad)
        if self.world == 1:
            return
        if not self.host_transport:
            why = self._create_rccl()
            ok, bad = self.rv.all_ok(why is None, why or "")
            if ok:
                return
            self.error = ("RCCL communicator not created (%s); the figures are the same library path over the host transport"
                          % "; ".join(w for _, w in bad))
            self.comm = None       # a half-created communicator is leaked on purpose: destroying it is a collective too
            self.host_transport = True
        self.comm = self.cvt.Comm.over_rendezvous(self.rv)

    def train_books(self):
        """rank 0 trains the sub-codebooks on a 100K-row sample, everyone gets the same bytes"""
        msg = None
        if self.rank == 0:
            try:
                M = self.M
                tmp = self.cvt.OpqIndex(self.zero_coarse, np.zeros((M, K, D // M), np.float32), R=self.R)
                sample = tmp.rotate(self.synth.sift_like(100_000, D, seed=0xC0FFEE, device=self.dev))
                msg = (True, self.synth.train_books(sample, M, K, iters=4).astype(np.float32).tobytes())
                tmp.close()
            # the others are waiting for this broadcast: they get the failure instead of the bytes
            except Exception as e:
                msg = (False, "%s: %s" % (type(e).__name__, e))
        ok, blob = self.rv.bcast(msg)
        if not ok:
            raise SystemExit("bench.py: rank 0 could not train the codebooks: %s" % blob)
        self.books = np.frombuffer(blob, dtype=np.float32).reshape(self.M, K, D // self.M).copy()

    def build_index(self, r0, r1, data="sift", seed=0xC0FFEE, nn_q=None):
        """rows [r0, r1) on device: generate -> rotate (MFMA GEMM) -> encode -> append; only codes stay.  nn_q (a few
        queries): the exact fp32 L2 nearest neighbour of each among THESE rows is tracked while the rows exist (they are
        never stored) -> self.nn = (distance, row id), the ground truth of recall_at_1 at sizes no fp32 matrix fits"""
        args, torch, synth = self.args, self.torch, self.synth
        ix = self.cvt.OpqIndex(self.zero_coarse, self.books, R=self.R)
        ix.reserve(r1 - r0); ix.set_id_base(r0)
        rows_done, t_acc = 0, 0.0
        self.nn = None
        if data == "random":
            g = torch.Generator(device=self.dev); g.manual_seed(0x51F7 + self.rank)
            for a in range(r0, r1, 1 << 24):
                b = min(r1, a + (1 << 24))
                ix.add_codes(torch.randint(0, 256, (b - a, self.M), generator=g, device=self.dev, dtype=torch.uint8))
        else:
            if nn_q is not None and nn_q.shape[0] > 0:
                self.nn = (torch.full((nn_q.shape[0],), float("inf"), device=self.dev),
                           torch.full((nn_q.shape[0],), -1, dtype=torch.int64, device=self.dev))
            step = synth.CHUNK * 4
            for a in range(r0, r1, step):
                b = min(r1, a + step)
                x = synth.sift_like(b - a, D, seed=seed, row_begin=a, device=self.dev)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                _, codes = ix.rotate_encode(x)
                ix.add_codes(codes)
                torch.cuda.synchronize(); t_acc += time.perf_counter() - t0  # data generation excluded
                rows_done += b - a
                if self.nn is not None:
                    m, j = torch.cdist(nn_q, x).min(dim=1)
                    upd = m < self.nn[0]
                    self.nn = (torch.where(upd, m, self.nn[0]), torch.where(upd, j + a, self.nn[1]))
        ix.set_param("qtile", args.qtile); ix.set_param("splits", args.splits); ix.set_param("profile", 1)
        if args.variant >= 0:
            ix.set_param("scan_variant", args.variant)
        return ix, rows_done, t_acc

    def searcher(self, ix):
        if self.comm is None:
            return lambda qq: ix.search(qq, self.k, rotate=True)

        def sharded(qq):
            # The rank whose LOCAL search fails gets the error from its own call (it has entered the collective all the same, so its
            # peers are not left waiting: their results are voided on the device and their status check says so).  The bench keeps
            # going on every rank -- the line has to appear, with the reason in it.
            try:
                return ix.search_sharded(self.comm, qq, self.k, rotate=True)
            except self.cvt.CvtmiError as e:
                self.search_error = str(e)
                torch = self.torch
                return (torch.full((qq.shape[0], self.k), float("inf"), device=self.dev),
                        torch.full((qq.shape[0], self.k), -1, dtype=torch.int64, device=self.dev))
        return sharded


def scan_roofline(sc, traffic=None):
    """Roofline of one ADC-scan launch.  The kernel's bound is the LDS look-up rate (one 2-byte table entry per code byte and
    query: 15-bit tables, 8 queries per 16-byte read): that is `achieved` / `peak` / `frac`, and no input can push it past 1.
    SURVEY 8(d)'s algorithmic code bytes (passes x rows x M, 8 queries per pass) are kept as `algorithmic_hbm_equiv` -- an
    equivalent rate, NOT HBM utilisation: the query groups of a row split read each row chunk through L2 / Infinity Cache, so one
    HBM read of the codes serves `effective_queries_per_hbm_read` queries (algorithmic bytes x 8 / PMC bytes), far more than 8."""
    sec = sc["ms"] * 1e-3
    lookups = sc["code_bytes"] * sc["qtile"] / sec          # table look-ups per second
    lds_tbs = lookups * 2.0 / 1e12
    alg = sc["code_bytes"] / sec / 1e9
    return {"bound": "lds", "achieved": round(lds_tbs, 2), "peak": round(LDS_PEAK_TBS, 1), "unit": "TB/s",
            "frac": round(lds_tbs / LDS_PEAK_TBS, 4), "frac_of_measured_lds_peak": round(lds_tbs / LDS_MEASURED_TBS, 4),
            "what": "table look-ups/s x 2 B per look-up against 256 B/clk/CU x 256 CU x 2.4 GHz (the guide measures ~150 TB/s)",
            "lds_lookups_per_s_T": round(lookups / 1e12, 2), "kernel_ms": round(sc["ms"], 4),
            "kernel_ms_what": "mean over the timed steps, HIP events on the launch stream inside the library (they bracket the "
                              "launch: 1-2 % above the kernel's own duration)",
            "queries_per_pass": sc["qtile"], "row_splits": sc["splits"],
            "traffic": traffic,
            "algorithmic_hbm_equiv": {
                "bytes_per_launch": sc["code_bytes"], "achieved_GBps": round(alg, 1), "frac_of_hbm_peak": round(alg / HBM_PEAK_GBS, 4),
                "effective_queries_per_hbm_read": (round(sc["code_bytes"] * sc["qtile"] / traffic, 1) if traffic else None),
                "hbm_frac_measured": (round(traffic / sec / 1e9 / HBM_PEAK_GBS, 4) if traffic else None),
                "what": "SURVEY 8(d): passes x rows x M code bytes per launch / kernel time; may exceed the HBM peak because the "
                        "passes share their reads on chip -- hbm_frac_measured (PMC bytes / kernel time / 8 TB/s) is what HBM sees"}}


def shard_evidence(ctx, ix, r0, q, out, res):
    """What the row-sharded line has to prove (BASELINE.md parity gates: "multi-GPU result identical", "recall@1 identical
    to the CPU reference"), measured, not asserted: (i) every rank scans ITS OWN shard with the CPU oracle for a few of the
    queries (orc_adc_search, ids = local row + shard offset), rank 0 merges the per-shard lists with the oracle's merge
    (orc_merge_topk: (distance, id) order) and compares ids and distance bits with what the GPUs' all-gather + merge
    returned -- FLANN-MPI's local search / offset / reduce (mpi/index.h:196-226, :74-108) restated on the CPU;
    (ii) the same oracle run, all ranks side by side, is the CPU baseline of this workload; (iii) recall@1 against the
    exact fp32 neighbour tracked while the rows were generated."""
    args, k, world = ctx.args, ctx.k, ctx.world
    d_gpu, i_gpu = out
    nqs = min(args.oracle_queries, q.shape[0]) if args.cpu_sample > 0 else 0
    if nqs > 0:
        from oracle import binding as ob
        if ctx.rank == 0:
            ob.build(o3=True)
        ctx.rv.barrier()
        orc = ob.Oracle(o3=True)
        _, _, codes_h = ix.get_entries()
        q_rot = orc.rotate_fma(ctx.R, q[:nqs].cpu().numpy())
        t0 = time.perf_counter()
        od, oi = orc.adc_search(q_rot, ctx.books, codes_h, k, id_base=r0)
        t_cpu = time.perf_counter() - t0
        del codes_h
        parts = ctx.rv.gather((od.tobytes(), oi.tobytes(), t_cpu))
        if ctx.rank == 0:
            ods = np.stack([np.frombuffer(p[0], np.float32).reshape(nqs, k) for p in parts], axis=1)   # [nq][world][k]
            ois = np.stack([np.frombuffer(p[1], np.int64).reshape(nqs, k) for p in parts], axis=1)
            md, mi = orc.merge_topk(ods, ois, k)
            gd, gi = d_gpu[:nqs].cpu().numpy(), i_gpu[:nqs].cpu().numpy()
            t_max = max(p[2] for p in parts)
            res["identical_to_oracle_sample"] = {
                "queries": nqs, "ids_identical": bool(np.array_equal(mi, gi)),
                "distances_bit_identical": bool(np.array_equal(md.view(np.uint32), gd.view(np.uint32))),
                "recall_at_1_identical_to_cpu": bool(np.array_equal(mi[:, 0], gi[:, 0])),
                "what": "every rank: oracle ADC scan + top-%d of its own shard (ids offset by the shard's first row); rank 0: "
                        "oracle merge of the %d per-shard lists vs the GPUs' all-gather + merge" % (k, world)}
            res["cpu_baseline"] = {
                "value": round(nqs / t_max, 3), "unit": "queries/s", "cores": world, "kind": "port",
                "sample": "%d of the queries against all %d rows: %d processes side by side, one thread and one shard each "
                          "(oracle/cvt_oracle.c -O3: tables + scan + top-%d), slowest rank %.1f s, the merge excluded; host: %s" % (
                              nqs, res["rows_total"], world, k, t_max, _cpu_model())}
    if ctx.nn is not None:
        ns = ctx.nn[0].shape[0]
        parts = ctx.rv.gather((ctx.nn[0].cpu().numpy().tobytes(), ctx.nn[1].cpu().numpy().tobytes()))
        if ctx.rank == 0:
            bd = np.stack([np.frombuffer(p[0], np.float32) for p in parts])      # [world][ns]
            bi = np.stack([np.frombuffer(p[1], np.int64) for p in parts])
            exact = bi[np.argmin(bd, axis=0), np.arange(ns)]
            res["recall_at_1"] = round(float((i_gpu[:ns, 0].cpu().numpy() == exact).mean()), 4)
            res["recall_at_1_what"] = ("ADC top-1 == exact fp32 L2 nearest neighbour among all %d rows (tracked chunk by chunk "
                                       "while the rows were generated), first %d queries" % (res["rows_total"], ns))


def run_sift1b(ctx, q, steps, warmup):
    """the SIFT-1B-shaped workload, row-sharded over the ranks (N > 1: the headline; N = 1: the first
    point of its curve)"""
    args, torch, cvt = ctx.args, ctx.torch, ctx.cvt
    l0, l1 = cvt.shard_range(args.large_rows, ctx.rank, ctx.world)
    free_b, _ = torch.cuda.mem_get_info(ctx.dev)
    # codes + the scan's rotated copy + generation chunks + tables must fit; every rank takes the same decision
    sharing = 1 if not ctx.host_transport else -(-ctx.world // max(1, torch.cuda.device_count()))   # ranks per device
    fits = ctx.rv.min(1 if free_b > sharing * ((l1 - l0) * ctx.M * 2.3 + (3 << 30)) else 0)
    if fits == 0:
        return {"error": "not enough free HBM for %d code rows per GPU" % (l1 - l0)}, None
    ql = q[:min(args.large_nq, ctx.nq)].contiguous()
    ns = min(args.recall_sample, 256, ql.shape[0])
    t0 = time.perf_counter()
    big, enc_rows, enc_t = ctx.build_index(l0, l1, data=args.large_data, seed=0xC0FFEE, nn_q=ql[:ns] if ns > 0 else None)
    torch.cuda.synchronize(); t_build = time.perf_counter() - t0
    fn = ctx.searcher(big)
    for _ in range(warmup):
        fn(ql)
    def last_scan():   # (a rank whose local search failed before its scan has nothing profiled: the line survives that too)
        try:
            return big.last_scan()
        except cvt.CvtmiError:
            return None
    ctx.barrier(); last_scan()
    c0 = ctx.comm.info() if ctx.comm is not None else None
    el, out = ctx.timed(lambda: fn(ql), steps, 0)
    ctx.last_out = out   # (headline_multi compares the sharded answer with the same workload on ONE GPU)
    sc = last_scan()
    res = {"value": round(ql.shape[0] * steps / el, 1), "unit": "queries/s", "ms_per_step": round(el / steps * 1e3, 4),
           "steps": steps, "rows_total": args.large_rows, "rows_per_gpu": l1 - l0, "nq": int(ql.shape[0]), "k": ctx.k,
           "n_gpus": ctx.world, "scan": scan_roofline(sc, _pmc_traffic_large(args, l1 - l0, int(ql.shape[0]), ctx.k)) if sc else None,
           "index_build_s_this_rank": round(t_build, 2),
           "encode_rows_per_s_this_rank": round(enc_rows / enc_t, 1) if enc_t > 0 else None,
           "data": ("synthetic SIFT-shaped rows generated, rotated and PQ-encoded on device"
                    if args.large_data == "sift" else "uniform random code bytes"),
           "what": "%d rows row-sharded x%d (%.2f GB of codes per GPU), the same %d queries on every rank, top-%d, "
                   "%s" % (
               args.large_rows, ctx.world, (l1 - l0) * ctx.M / 1e9, ql.shape[0], ctx.k,
               "ONE all-gather of the per-shard top-k inside libcvtmi + merge" if ctx.world > 1 else "single shard")}
    if ctx.comm is not None:
        try:
            ctx.comm.status()   # deferred status check of the sharded searches: a rank that failed locally surfaces here
        except cvt.CvtmiError as e:   # the line still appears, and says so
            res["error"] = "a rank's local search failed during the timed steps (its results were voided): %s" % e
        own = [w for w in ctx.rv.allgather(ctx.search_error or "") if w]   # (the failing rank's own call returned the error to it)
        if own:
            res["error"] = "; ".join(([res["error"]] if "error" in res else []) + own)
        c1 = ctx.comm.info()
        res["comm"] = c1
        res["transport"] = c1["transport"]
        res["rccl_ranks"] = c1["world"] if c1["transport"] == "rccl" else 0
        res["collectives_per_search"] = round((c1["collectives"] - c0["collectives"]) / max(steps, 1), 3)
    if ctx.world > 1:
        shard_evidence(ctx, big, l0, ql, out, res)
    elif ctx.nn is not None:
        ns = ctx.nn[0].shape[0]
        res["recall_at_1"] = round(float((out[1][:ns, 0] == ctx.nn[1]).float().mean().item()), 4)
        res["recall_at_1_what"] = "ADC top-1 == exact fp32 L2 nearest neighbour among all %d rows, first %d queries" % (args.large_rows, ns)
    big.close()
    return res, sc


def _pmc_traffic(args, nq, k, M):
    """HBM bytes per scan launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
        separate runs of
    this same command; tools/profile_round.sh).  Counters cannot be collected inside the timed run, so the JSON says
    where the number comes from; null when the launch shape differs from the profiled one."""
    default_shape = (args.rows == 1_000_000 and nq == 10_000 and k == 100 and M == 16 and not args.qtile
                     and not args.splits and args.variant < 0)
    if not default_shape:
        return None, None
    for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
        p = os.path.join(ROOT, "profiles", "%s_scan_traffic.json" % tag)
        if os.path.exists(p):
            try:
                d = json.load(open(p))
                return d.get("hbm_bytes_per_launch"), "profiles/%s_scan_traffic.json: %s; kernel %s" % (
                    tag, d.get("note", "rocprofv3 --pmc passes"), d.get("kernel", "?"))
            except Exception:
                pass
    return None, None


def _pmc_traffic_large(args, rows, nq, k):
    """the same for the HBM-resident shard shape the round's profile covers (128 M rows x 2048 queries), else null"""
    for tag in ("r06", "r05", "r04", "r03", "r02"):
        p = os.path.join(ROOT, "profiles", "%s_scan_traffic_128m.json" % tag)
        if os.path.exists(p):
            try:
                d = json.load(open(p))
                if d.get("rows") == rows and d.get("nq") == nq and k == 100:
                    return d.get("hbm_bytes_per_launch")
            except Exception:
                pass
    return None


def headline_n1(ctx, q):
    """N = 1: SIFT-1M, configs[1]"""
    args, nq, k, M = ctx.args, ctx.nq, ctx.k, ctx.M
    idx, enc_rows, enc_time = ctx.build_index(0, args.rows)
    fn = ctx.searcher(idx)
    for _ in range(args.warmup):
        fn(q)
    ctx.barrier(); idx.last_scan()  # drop the warm-up launches from the kernel-time statistics
    elapsed, out = ctx.timed(lambda: fn(q), args.steps, 0)
    scan = idx.last_scan()  # mean HIP-event duration of the scan kernel over the timed steps
    traffic, traffic_src = _pmc_traffic(args, nq, k, M)
    name = "SIFT-1M" if args.rows == 1_000_000 else "%d synthetic rows" % args.rows
    roof = scan_roofline(scan, traffic)
    roof["kernel"] = "adc_scan kernel (M=%d, %d queries per pass)" % (M, scan["qtile"])
    roof["traffic_source"] = traffic_src
    result = {
        "metric": "queries/sec, OPQ-ADC top-%d over 128-d %s" % (k, name),
        "value": round(nq * args.steps / elapsed, 1), "unit": "queries/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8 codes / f32 distances", "data": "synthetic",
        "config": {"workload": "%s synthetic 128-d, OPQ M=%d K=256 (dense 128x128 rotation), ADC scan + top-%d, nq=%d "
                               "queries per step" % (name if args.rows == 1_000_000 else "%d-row" % args.rows, M, k,
                                                     nq),
                   "rows": args.rows, "rows_per_gpu": args.rows, "nq_per_step": nq, "k": k, "M": M,
                   "parallelism": "1 GPU", "qtile": scan["qtile"], "row_splits": scan["splits"]},
        "roofline": roof,
        "encode": {"rows_per_s": round(enc_rows / enc_time, 1),
                   "what": "rotate (MFMA GEMM) + PQ encode (cvtmi_opq_rotate_encode) + append of the 1 M rows"},
    }
    if not args.host_api:   # (profiling runs: every scan launch of the process is then a headline launch)
        return result, idx, out
    # the reference's API takes host pointers: the same step through cvtmi_opq_search (queries up, results down over
    # PCIe)
    qh = q.cpu().numpy()
    # the caller's result arrays, reused across calls (fresh ones cost a page fault per 4 KB)
    out_h = (np.zeros((nq, k), np.float32), np.zeros((nq, k), np.int64))
    idx.search(qh, k, rotate=True, out=out_h)
    t0 = time.perf_counter()
    reps = max(1, min(10, args.steps))
    for _ in range(reps):
        idx.search(qh, k, rotate=True, out=out_h)
    el_h = (time.perf_counter() - t0) / reps
    what = ("cvtmi_opq_search with host buffers (the reference's call shape): %.1f MB of queries in, %.1f MB of results out per step over "
            "PCIe, in pieces of 4096 queries that alternate between two scratch sets / streams of the handle (upload of piece i+1 and "
            "download of piece i-1 beside the scan of piece i); pageable numpy arrays, reused across calls -- reported beside `value`, "
            "never as it" % (nq * D * 4 / 1e6, nq * k * 12 / 1e6))
    result["host_pointer_api"] = {"value": round(nq / el_h, 1), "unit": "queries/s",
                                  "ms_per_step": round(el_h * 1e3, 4), "what": what}
    try:   # the same with page-locked arrays from the library's allocator: no staging copy on either side
        qp = ctx.cvt.pinned_empty((nq, D), np.float32); qp[:] = qh
        outp = (ctx.cvt.pinned_empty((nq, k), np.float32), ctx.cvt.pinned_empty((nq, k), np.int64))
        idx.search(qp, k, rotate=True, out=outp)
        t0 = time.perf_counter()
        for _ in range(reps):
            idx.search(qp, k, rotate=True, out=outp)
        el_p = (time.perf_counter() - t0) / reps
        result["host_pointer_api"]["page_locked_arrays"] = {
            "value": round(nq / el_p, 1), "unit": "queries/s", "ms_per_step": round(el_p * 1e3, 4),
            "identical": bool(np.array_equal(outp[1], out_h[1]) and np.array_equal(outp[0].view(np.uint32), out_h[0].view(np.uint32))),
            "what": "queries and result arrays from cvtmi_host_alloc: the queries go up as they are, the kernels write the lists straight into "
                    "the caller's arrays (device-visible host memory) -- ONE launch chain, the one the device-pointer entry issues, no "
                    "device copy of the results and no copy back (round 5; opq_host_zero_copy)"}
    except Exception as e:
        result["host_pointer_api"]["page_locked_arrays"] = {"error": "%s: %s" % (type(e).__name__, e)}
    # the reference's own call pattern is a handful of query frames per Query (opq/src/multi_frame_index_test.cpp:45-54): wall time of a
    # call of 1 / 8 queries on the same index, device pointers (call + synchronise) and host pointers (the call returns the results)
    small = {"what": "wall time per search call of 1 / 8 queries, top-%d over the same %d rows: four launches (tables incl. the rotation, "
                     "sampled histogram, candidate lists, selection), no partial lists, no merge; host pointers: queries read and results "
                     "written through the handle's pinned staging area by the kernels" % (k, args.rows)}
    torch = ctx.torch
    for n_small in (1, 8):
        qs = q[:n_small].contiguous()
        qsh = qs.cpu().numpy()
        outs = (np.zeros((n_small, k), np.float32), np.zeros((n_small, k), np.int64))
        for _ in range(20):
            idx.search(qs, k, rotate=True); idx.search(qsh, k, rotate=True, out=outs)
        torch.cuda.synchronize()
        reps = 200
        t0 = time.perf_counter()
        for _ in range(reps):
            d_s, i_s = idx.search(qs, k, rotate=True)
            torch.cuda.synchronize()
        t_dev = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            idx.search(qsh, k, rotate=True, out=outs)
        t_host = (time.perf_counter() - t0) / reps
        small["nq=%d" % n_small] = {
            "device_pointers_ms": round(t_dev * 1e3, 4), "host_pointers_ms": round(t_host * 1e3, 4),
            "identical_to_the_batch": bool(np.array_equal(outs[1], out[1][:n_small].cpu().numpy()) and
                                           np.array_equal(outs[0].view(np.uint32), out[0][:n_small].cpu().numpy().view(np.uint32)) and
                                           torch.equal(i_s, out[1][:n_small]))}
    result["small_batch_latency"] = small
    return result, idx, out


def headline_multi(ctx, q):
    """N > 1: SIFT-1B-shaped, row-sharded, configs[3]"""
    args, cvt, nq, k, M, world = ctx.args, ctx.cvt, ctx.nq, ctx.k, ctx.M, ctx.world
    res, _ = run_sift1b(ctx, q, args.steps, args.warmup)
    if "value" not in res:   # nothing was measured (not enough HBM: every rank took the same decision): the line still appears, with the reason
        return {"metric": "queries/sec, OPQ-ADC top-%d over 128-d SIFT-1B-shaped rows, row-sharded x%d" % (k, world), "value": None,
                "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
                "scaling": "strong", "error": res["error"]} if ctx.rank == 0 else None
    extra = {}
    # The N = 1 point of THIS curve, measured in THIS run: rank 0 alone, every row on its one GPU, the same queries, no collective --
    # so the line carries its own speed-up (the N = 1 line's headline is BASELINE configs[1], another workload).  The others wait.
    if args.n1_point:
        one = None
        if ctx.rank == 0:
            try:
                free_b, _ = ctx.torch.cuda.mem_get_info(ctx.dev)
                if free_b > args.large_rows * M * 2.3 + (3 << 30):
                    t0 = time.perf_counter()
                    solo, _, _ = ctx.build_index(0, args.large_rows, data=args.large_data, seed=0xC0FFEE)
                    ctx.torch.cuda.synchronize(); t_build = time.perf_counter() - t0
                    ql = q[:res["nq"]].contiguous()
                    reps = max(1, min(args.steps, 2))
                    solo.search(ql, k, rotate=True); ctx.torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        d1, i1 = solo.search(ql, k, rotate=True)
                    ctx.torch.cuda.synchronize()
                    el1 = (time.perf_counter() - t0) / reps
                    ds, is_ = ctx.last_out
                    one = {"value": round(res["nq"] / el1, 1), "unit": "queries/s", "ms_per_step": round(el1 * 1e3, 3), "steps": reps,
                           "index_build_s": round(t_build, 2),
                           "sharded_result_identical": bool(ctx.torch.equal(i1, is_) and
                                                            ctx.torch.equal(d1.view(ctx.torch.int32), ds.view(ctx.torch.int32))),
                           "what": "rank 0 alone: all %d rows on one GPU, the same %d queries, cvtmi_opq_search_dev (no communicator)" % (
                               args.large_rows, res["nq"])}
                    solo.close()
                else:
                    one = {"error": "not enough free HBM on rank 0 for all %d rows" % args.large_rows}
            except Exception as e:   # the sharded measurement above must survive this side leg
                one = {"error": "%s: %s" % (type(e).__name__, e)}
        ctx.rv.barrier()
        if ctx.rank == 0 and one is not None:
            extra["same_workload_on_one_gpu"] = one
            if "value" in one:
                extra["speedup_over_one_gpu"] = round(res["value"] / one["value"], 3)
    # the small database on N GPUs, for the record: row-sharded through the same library path, and as N replicas
    r0, r1 = cvt.shard_range(args.rows, ctx.rank, world)
    shard, _, _ = ctx.build_index(r0, r1)
    fn = ctx.searcher(shard)
    el2, _ = ctx.timed(lambda: fn(q), args.steps, args.warmup)
    extra["sift1m_row_sharded"] = {
        "value": round(nq * args.steps / el2, 1), "unit": "queries/s", "ms_per_step": round(el2 / args.steps * 1e3, 4),
        "what": "the 1 M-row database row-sharded x%d (%d rows per GPU), all %d queries on every rank, all-gather + "
                "merge: "
                "strong scaling of a 16 MB problem" % (world, r1 - r0, nq)}
    shard.close()
    rep, _, _ = ctx.build_index(0, args.rows)
    q_mine = ctx.synth.sift_like(nq, D, seed=0xBEEF + 7919 * ctx.rank, device=ctx.dev) if ctx.rank else q
    el3, _ = ctx.timed(lambda: rep.search(q_mine, k, rotate=True), args.steps, args.warmup)
    extra["sift1m_replicas"] = {
        "value": round(world * nq * args.steps / el3, 1), "unit": "queries/s",
            "ms_per_step": round(el3 / args.steps * 1e3, 4),
        "what": "code matrix replicated x%d, every rank serves its own batch of %d queries, no collective: N "
                "independent "
                "replicas (Nx by construction)" % (world, nq)}
    rep.close()
    if ctx.rank != 0:
        return None
    rf = res["scan"] or {"queries_per_pass": None}
    result = {
        "metric": "queries/sec, OPQ-ADC top-%d over 128-d SIFT-1B-shaped rows, row-sharded x%d" % (k, world),
        "value": res["value"], "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u8 codes / f32 distances", "data": "synthetic",
        "config": {"workload": "SIFT-1B-shaped: %d synthetic 128-d rows, OPQ M=%d K=256, row-sharded over %d GPUs, ADC "
                               "scan + top-%d, nq=%d queries per step (the same batch on every rank)" % (
                                   args.large_rows, M, world, k, res["nq"]),
                   "rows": args.large_rows, "rows_per_gpu": res["rows_per_gpu"], "nq_per_step": res["nq"], "k": k,
                       "M": M,
                   "parallelism": "row-sharded x%d, ONE %s of per-shard top-%d + merge on every rank" % (
                       world, "ncclAllGather (RCCL, issued inside libcvtmi)" if res.get("transport") == "rccl"
                       else "all-gather through the HOST transport (cvtmi_comm_create_custom over the TCP rendezvous)", k),
                   "n1_point_of_this_curve": "the N = 1 line's 'sift1b'.value (same rows, same queries, one GPU)"},
        "roofline": dict(rf, kernel="adc_scan kernel (M=%d, %s queries per pass), rank 0's shard" % (M, rf["queries_per_pass"])),
    }
    # the evidence, at the top level of the line: who exchanged what, and that the answer is the CPU reference's
    for key in ("transport", "rccl_ranks", "collectives_per_search", "recall_at_1", "recall_at_1_what",
                "identical_to_oracle_sample", "cpu_baseline"):
        if key in res:
            result[key] = res[key]
    if ctx.error or "error" in res:
        result["error"] = "; ".join(e for e in (ctx.error, res.get("error")) if e)
    result["sift1b"] = res
    result.update(extra)
    return result


def recall_at_1(ctx, q, i_gpu, result):
    torch, synth, args = ctx.torch, ctx.synth, ctx.args
    ns = min(args.recall_sample, ctx.nq)
    if ns <= 0:
        return
    best = torch.full((ns,), float("inf"), device=ctx.dev)
    arg = torch.zeros((ns,), dtype=torch.int64, device=ctx.dev)
    qs = q[:ns]
    for a in range(0, args.rows, synth.CHUNK):
        b = min(args.rows, a + synth.CHUNK)
        x = synth.sift_like(b - a, D, seed=0xC0FFEE, row_begin=a, device=ctx.dev)
        m, j = torch.cdist(qs, x).min(dim=1)
        upd = m < best
        best = torch.where(upd, m, best); arg = torch.where(upd, j + a, arg)
    ctx.exact_nn = arg   # (secondary.opq_rotation_learning measures the same recall under a learned rotation)
    result["recall_at_1"] = round(float((i_gpu[:ns, 0] == arg).float().mean().item()), 4)
    result["recall_at_1_what"] = "ADC top-1 == exact fp32 L2 nearest neighbour, first %d queries" % ns


def cpu_baseline_opq(ctx, idx, q, out, result):
    """the oracle (CPU restatement of the reference path) on a bounded sample: baseline + identity check"""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import binding as ob
    args, nq, k = ctx.args, ctx.nq, ctx.k
    d_gpu, i_gpu = out
    ob.build(o3=True)
    orc = ob.Oracle(o3=True)
    cs = min(args.cpu_sample, nq)
    _, _, codes_h = idx.get_entries()
    q_rot = orc.rotate_fma(ctx.R, q[:cs].cpu().numpy())
    t0 = time.perf_counter()
    od, oi = orc.adc_search(q_rot, ctx.books, codes_h, k)
    t_cpu = time.perf_counter() - t0
    result["cpu_baseline_port"] = {
        "value": round(cs / t_cpu, 2), "unit": "queries/s", "cores": 1, "kind": "port",
        "sample": "%d of the %d queries against the full %d-row code matrix, LUT + scan + top-%d (oracle/cvt_oracle.c "
                  "-O3, 1 thread); host: %s" % (cs, nq, args.rows, k, _cpu_model()),
        "gpu_topk_ids_identical": bool(np.array_equal(oi, i_gpu[:cs].cpu().numpy())),
        "gpu_distances_bit_identical": bool(np.array_equal(od.view(np.uint32),
                                                           d_gpu[:cs].cpu().numpy().view(np.uint32)))}
    result["recall_at_1_identical_to_cpu"] = bool(np.array_equal(oi[:, 0], i_gpu[:cs, 0].cpu().numpy()))
    # the same loop on all host cores: queries split over threads (ctypes releases the GIL), bounded sample
    nth = _usable_cpus() if args.cpu_threads < 0 else args.cpu_threads
    if not nth or nth <= 1:
        return
    per = 8
    qs_mt = min(nq, nth * per)
    q_rot_mt = orc.rotate_fma(ctx.R, q[:qs_mt].cpu().numpy())
    chunks = [(a, min(qs_mt, a + per)) for a in range(0, qs_mt, per)]
    with ThreadPoolExecutor(max_workers=nth) as ex:
        t0 = time.perf_counter()
        parts = list(ex.map(lambda ab: orc.adc_search(q_rot_mt[ab[0]:ab[1]], ctx.books, codes_h, k), chunks))
        t_mt = time.perf_counter() - t0
    oi_mt = np.concatenate([p[1] for p in parts])
    result["cpu_baseline_all_cores"] = {
        "value": round(qs_mt / max(t_mt, 1e-9), 2), "unit": "queries/s", "cores": nth, "kind": "port",
        "sample": "%d queries in chunks of %d over %d threads, same loop as cpu_baseline" % (qs_mt, per, nth),
        "gpu_topk_ids_identical": bool(np.array_equal(oi_mt, i_gpu[:qs_mt].cpu().numpy()))}


def cpu_baseline_reference(ctx, q, result):
    """the reference's OWN code (opq/src/IVFOPQ.cpp compiled in place: oracle/_ref/libref_opq.so) on a bounded sample: IndexDatabase
    over one feature file of --ref-rows rows (Add: coarse argmin + PQ encode, IVFOPQ.cpp:105-170), then QueryThrehold
    (:322-422: tables + the scan over its 56-byte IVFelem entries + the per-video minimum) for --ref-queries queries.  One file =
    one video, so the call returns the minimum score per query -- clamped at 1.0 by the reference (:5, :262) and therefore not
    comparable with the unclamped top-k here: this leg is the timing, the identity check is the port's (cpu_baseline_port)."""
    from oracle import binding as ob
    args = ctx.args
    if not ob.ref_available() or args.ref_rows <= 0:
        result["cpu_baseline"] = dict(result["cpu_baseline_port"], note="oracle/_ref/libref_opq.so not built: the port stands in")
        return
    rows_s, nqs = min(args.rows, args.ref_rows), min(ctx.nq, args.ref_queries)
    x = ctx.synth.sift_like(rows_s, D, seed=0xC0FFEE, device=ctx.dev).cpu().numpy()
    ref = ob.RefOPQ(ctx.zero_coarse, ctx.books, np.arange(D, dtype=np.int32), max_index_num=max(rows_s, 1) + 16)
    try:
        t0 = time.perf_counter(); ref.index([x]); t_index = time.perf_counter() - t0
        t0 = time.perf_counter(); ms = ref.query(q[:nqs].cpu().numpy(), 1, 1); t_q = time.perf_counter() - t0
    finally:
        ref.close()
    result["cpu_baseline"] = {
        "value": round(nqs / t_q * rows_s / args.rows, 2), "unit": "queries/s", "cores": 1, "kind": "reference",
        "sample": "the reference's IVFOPQ::QueryThrehold, %d queries over a %d-row index it built itself from one feature file "
                  "(IndexDatabase), %.2f s; %s; host: %s" % (
                      nqs, rows_s, t_q,
                      "measured at the headline's own row count" if rows_s == args.rows else
                      "value = queries/s x %d / %d rows (its scan is linear in the rows)" % (rows_s, args.rows), _cpu_model()),
        "rows": rows_s, "extrapolated": rows_s != args.rows,
        "queries_per_s_on_the_sample": round(nqs / t_q, 2),
        "reference_index_build_rows_per_s": round(rows_s / t_index, 1),
        "min_score_clamped_at_1": bool(np.all(ms == 1.0)),
        "identity_check": "cpu_baseline_port (same arithmetic restated, unclamped): gpu_topk_ids_identical / gpu_distances_bit_identical"}


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks (this same script, one process per GPU) with the
    environment torch.distributed.run would give them, pass rank 0's stdout (the JSON line) through, and stop everybody as soon
    as one of them fails.  Only the exact processes started here are ever signalled."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    base = dict(os.environ, WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                LOCAL_WORLD_SIZE=str(args.gpus), TORCHELASTIC_RUN_ID="cvtmi-self-%d" % os.getpid(),
                HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(args.gpus):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdin=subprocess.DEVNULL,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        live = list(procs)
        while live and rc == 0:
            time.sleep(0.2)
            for p in list(live):
                c = p.poll()
                if c is not None:
                    live.remove(p)
                    if c != 0:
                        rc = c
    finally:
        for p in procs:
            if p.poll() is None:
                if rc != 0:
                    p.terminate()
                try:
                    p.wait(timeout=30 if rc != 0 else None)
                except subprocess.TimeoutExpired:
                    p.kill(); p.wait()
    return rc


def main():
    args = parse_args()
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))
    ctx = Ctx(args)
    for pair in [v for v in args.tune.split(",") if v]:
        name, value = pair.split("=")
        ctx.cvt.set_tuning(name, float(value))
    ctx.train_books()
    ready, why = True, ""
    try:  # a rank that cannot use its GPU must say so before anybody enters the (blocking) communicator creation
        probe = ctx.cvt.OpqIndex(ctx.zero_coarse, ctx.books, R=ctx.R)
        probe.close()
    except Exception as e:
        ready, why = False, "%s: %s" % (type(e).__name__, e)
    ctx.make_comm(ready, why)
    q = ctx.synth.sift_like(ctx.nq, D, seed=0xBEEF, device=ctx.dev)  # identical on every rank
    if ctx.world == 1:
        result, idx, out = headline_n1(ctx, q)
        if args.large_rows > 0:
            try:
                result["sift1b"] = run_sift1b(ctx, q, max(1, min(args.steps, 2)), 1)[0]
            except ctx.cvt.CvtmiError as e:
                result["sift1b"] = {"error": str(e)}
        # outside the timed region: recall@1, the CPU baseline and the secondary kernels
        recall_at_1(ctx, q, out[1], result)
        if args.cpu_sample > 0:
            cpu_baseline_opq(ctx, idx, q, out, result)
            cpu_baseline_reference(ctx, q, result)
        idx.close()
        if args.secondary:
            try:
                result["secondary"] = secondary(ctx)
            except Exception as e:  # the headline line must survive a failing side measurement
                result["secondary"] = {"error": "%s: %s" % (type(e).__name__, e)}
    else:
        result = headline_multi(ctx, q)
    if ctx.rank == 0:
        print(json.dumps(result), flush=True)
    if ctx.comm is not None:
        try:
            ctx.comm.close()
        except ctx.cvt.CvtmiError as e:
            print("bench.py: rank %d: %s" % (ctx.rank, e), file=sys.stderr)
    ctx.rv.barrier()
    ctx.rv.close()
    if ctx.hung_thread:   # a helper thread is still inside ncclCommInitRank: the interpreter's shutdown would wait for RCCL
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)


# ------------------------------------------------------------------------------------------------------------------
# secondary measurements (N = 1, outside the timed region): the other rows of SURVEY.md 8 / BASELINE configs
# ------------------------------------------------------------------------------------------------------------------
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


def _ev_ms_steady(torch, fn, span_ms=60.0, warm=2):
    """like _ev_ms over enough back-to-back launches to span span_ms: short kernels measured over five launches sit in the first
    milliseconds after an idle gap, where the same kernel runs 10-25 % slower than in a sustained stream of launches (round 6: the
    rotation of 1 M rows 0.37 ms = 0.58 of the matrix peak over 5 launches, 0.30 ms = 0.73 over 400; tools/rotate_ab.py)"""
    first = _ev_ms(torch, fn, reps=5, warm=warm)
    reps = int(max(5, min(400, span_ms / max(first, 1e-3))))
    return _ev_ms(torch, fn, reps=reps, warm=0), first, reps


def _hbm(nbytes, ms):
    gb = nbytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(gb, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gb / HBM_PEAK_GBS, 4)}


def secondary(ctx):
    sec = {}
    only = [v for v in ctx.args.only.split(",") if v]
    if not only or "rotation_encode" in only:
        sec.update(_sec_rotation_encode(ctx))
    vmin, vdiff = _sec_sq8(ctx, sec) if (not only or "sq8" in only or "flat_u8_c3" in only) else (None, None)
    for name, fn in (("opq_m8", lambda: _sec_opq_m8(ctx)), ("opq_rotation_learning", lambda: _sec_rotation_learning(ctx)),
                     ("flat_f32", lambda: _sec_flat_f32(ctx)), ("flat_u8_c3", lambda: _sec_flat_u8_c3(ctx, vmin,
                                                                                                      vdiff)),
                     ("ivf_query", lambda: _ivf_query(ctx)), ("hnsw_c5", lambda: _hnsw_c5(ctx))):
        if only and name not in only:
            continue
        try:
            sec[name] = fn()
        except Exception as e:
            sec[name] = {"error": "%s: %s" % (type(e).__name__, e)}
    return sec


def _sec_rotation_encode(ctx):
    """a-R rotation (fp32 MFMA GEMM) and a-E encode, 2^20 rows"""
    torch, cvt, synth = ctx.torch, ctx.cvt, ctx.synth
    sec, n, M = {}, 1 << 20, ctx.M
    ix = cvt.OpqIndex(ctx.zero_coarse, ctx.books, R=ctx.R)
    x = synth.sift_like(n, D, seed=0xC0FFEE, device=ctx.dev)
    ms, ms5, reps = _ev_ms_steady(torch, lambda: ix.rotate(x))
    tf = 2.0 * n * D * D / (ms * 1e-3) / 1e12
    gbs = n * D * 8 / (ms * 1e-3) / 1e9
    sec["rotation"] = {"rows": n, "ms": round(ms, 4), "launches_timed": reps, "ms_over_the_first_5_launches": round(ms5, 4), "tflops": round(tf, 1),
                       "frac_of_f32_mfma_peak": round(tf / F32_MFMA_PEAK_TF, 4), "hbm_GBps": round(gbs, 1),
                       "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
                       "what": "Y = X R^T, dense 128 x 128 fp32 R on v_mfma_f32_32x32x2_f32 (2 D^2 flop and 1 KB "
                               "moved per row)"}
    # the same kernel over 8 x the rows: the rate of a launch grows with its length (round 6: 1 M rows 0.57-0.60, 8 M rows 0.72)
    x8 = synth.sift_like(8 * n, D, seed=0xC0FFEE, device=ctx.dev)
    ms8 = _ev_ms(torch, lambda: ix.rotate(x8), reps=3, warm=1)
    sec["rotation"]["rows_8M"] = {"rows": 8 * n, "ms": round(ms8, 4), "frac_of_f32_mfma_peak": round(2.0 * 8 * n * D * D / (ms8 * 1e-3) / 1e12 / F32_MFMA_PEAK_TF, 4)}
    del x8
    xr = ix.rotate(x)
    ms, ms5, reps = _ev_ms_steady(torch, lambda: ix.encode(xr))
    sec["encode"] = {"rows": n, "ms": round(ms, 4), "launches_timed": reps, "ms_over_the_first_5_launches": round(ms5, 4), "rows_per_s": round(n / (ms * 1e-3), 1),
                     "what": "PQ encode of rotated rows, M=%d K=256 (bf16 matrix-core filter + exact chain)" % M}
    ms, ms5, reps = _ev_ms_steady(torch, lambda: ix.rotate_encode(x))
    sec["rotate_encode"] = {"rows": n, "ms": round(ms, 4), "launches_timed": reps, "ms_over_the_first_5_launches": round(ms5, 4), "rows_per_s": round(n / (ms * 1e-3), 1),
                            "what": "raw rows -> codes in one call (cvtmi_opq_rotate_encode), dense rotation"}
    # the reference's own rotation is a permutation of the dimensions (reorder_, IVFOPQ.cpp:424-439)
    ixp = cvt.OpqIndex(ctx.zero_coarse, ctx.books, perm=synth.random_permutation(D, seed=5))
    ms_two = _ev_ms(torch, lambda: ixp.encode(ixp.rotate(x)))
    ms = _ev_ms(torch, lambda: ixp.rotate_encode(x))
    sec["reorder_encode"] = {"rows": n, "ms": round(ms, 4), "rows_per_s": round(n / (ms * 1e-3), 1),
                             "rows_per_s_permute_then_encode": round(n / (ms_two * 1e-3), 1),
                             "what": "raw rows -> codes for a permutation model (the reference's IVFOPQ::reorder + "
                                     "Add): one "
                                     "kernel, the rows are read through the permutation"}
    ixp.close(); ix.close()
    return sec


def _sec_opq_m8(ctx):
    """BASELINE configs[0]'s model shape (M = 8, K = 256: the reference's own test model) at the headline's table size, and M = 4:
    the native packed scan (round 6, adc_scan16p: 16 / M rows per 16-byte load), next to the padded rows through the M = 16 kernels
    (round 5) and the row-per-lane kernels it had before -- all three must return the same lists"""
    torch, cvt, synth, args = ctx.torch, ctx.cvt, ctx.synth, ctx.args
    k, nq = ctx.k, ctx.nq
    out = {}
    for Mx in (8, 4):
        tmp = cvt.OpqIndex(ctx.zero_coarse, np.zeros((Mx, K, D // Mx), np.float32), R=ctx.R)
        books = synth.train_books(tmp.rotate(synth.sift_like(100_000, D, seed=0xC0FFEE, device=ctx.dev)), Mx, K, iters=4)
        tmp.close()
        ix = cvt.OpqIndex(ctx.zero_coarse, books, R=ctx.R)
        ix.reserve(args.rows)
        step = synth.CHUNK * 4
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for a in range(0, args.rows, step):
            b = min(args.rows, a + step)
            _, codes = ix.encode(ix.rotate(synth.sift_like(b - a, D, seed=0xC0FFEE, row_begin=a, device=ctx.dev)))
            ix.add_codes(codes)
        torch.cuda.synchronize(); t_build = time.perf_counter() - t0
        q = synth.sift_like(nq, D, seed=0xBEEF, device=ctx.dev)
        res = {"rows": args.rows, "M": Mx, "nq": nq, "k": k, "index_build_s": round(t_build, 3),
               "what": "rotate + encode + ADC top-%d of %d queries over %d rows of %d-byte codes" % (k, nq, args.rows, Mx)}
        ref = None
        try:
            for name, pad, packed in (("packed_rows_native_scan", 1, 1), ("padded_rows_through_the_m16_kernels", 1, 0), ("row_per_lane_kernels", 0, 0)):
                cvt.set_tuning("scan_pad_m", pad); cvt.set_tuning("scan_packed_m", packed)
                ms = _ev_ms(torch, lambda: ix.search(q, k, rotate=True), reps=max(2, min(5, args.steps)), warm=2)
                d, i = ix.search(q, k, rotate=True)
                if ref is None:
                    ref = (d, i)
                res[name] = {"ms": round(ms, 4), "queries_per_s": round(nq / (ms * 1e-3), 1),
                             "identical": bool(torch.equal(i, ref[1]) and torch.equal(d.view(torch.int32), ref[0].view(torch.int32)))}
        finally:
            cvt.set_tuning("scan_pad_m", 1); cvt.set_tuning("scan_packed_m", 1)
        if ctx.exact_nn is not None and ctx.exact_nn.shape[0] >= nq:
            res["recall_at_1"] = round(float((ref[1][:, 0] == ctx.exact_nn[:nq]).float().mean().item()), 4)
        ix.close()
        if Mx == 8:
            out = res
        else:
            out["m4"] = res
    return out


def _sec_rotation_learning(ctx):
    """f-3 (optional): the dense rotation LEARNED (cvtmi_opq_learn_rotation: k-means / orthogonal Procrustes alternation) instead of
    drawn at random as in the headline -- what it costs, and what it buys at the same 16 bytes per row"""
    torch, cvt, synth, args = ctx.torch, ctx.cvt, ctx.synth, ctx.args
    M, k, ns = ctx.M, ctx.k, min(1000, ctx.nq)
    sample = synth.sift_like(100_000, D, seed=0xC0FFEE, device=ctx.dev)
    res = {"sample_rows": 100_000, "outer_iterations": 8, "kmeans_iterations": 4,
           "what": "R, books = cvtmi_opq_learn_rotation(sample): X R^T on the fp32 MFMA GEMM, per-sub-space Lloyd iterations, X^T Y in double, "
                   "R = V U^T by Jacobi on the host; then the %d rows re-encoded and the first %d queries searched, recall@1 = ADC top-1 == exact "
                   "fp32 neighbour -- next to the same under the identity (plain PQ) and under the headline's random rotation" % (args.rows, ns)}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    R, books = cvt.opq_learn_rotation(sample, M, K, 8, 4, 1234)
    torch.cuda.synchronize(); res["learn_s"] = round(time.perf_counter() - t0, 3)
    Rn = R.cpu().numpy()
    res["orthonormality_error"] = float(np.abs(Rn.astype(np.float64) @ Rn.astype(np.float64).T - np.eye(D)).max())
    q = synth.sift_like(ns, D, seed=0xBEEF, device=ctx.dev)

    def evaluate(Rm, bk):
        ix = cvt.OpqIndex(ctx.zero_coarse, bk, R=Rm)
        ix.reserve(args.rows)
        err, step = 0.0, synth.CHUNK * 4
        for a in range(0, args.rows, step):
            b = min(args.rows, a + step)
            x = synth.sift_like(b - a, D, seed=0xC0FFEE, row_begin=a, device=ctx.dev)
            xr = ix.rotate(x)
            _, codes = ix.encode(xr)
            ix.add_codes(codes)
            if a == 0:   # quantisation error on the first chunk
                bt = torch.from_numpy(bk).to(ctx.dev)
                y = torch.cat([bt[m][codes[:, m].long()] for m in range(M)], dim=1)
                err = float(((xr - y) ** 2).sum(dim=1).mean().item())
        _, i = ix.search(q, k, rotate=True)
        ix.close()
        out = {"mean_squared_quantisation_error": round(err, 6)}
        if ctx.exact_nn is not None and ctx.exact_nn.shape[0] >= ns:
            out["recall_at_1"] = round(float((i[:, 0] == ctx.exact_nn[:ns]).float().mean().item()), 4)
            out["recall_at_%d" % k] = round(float((i == ctx.exact_nn[:ns, None]).any(dim=1).float().mean().item()), 4)
        return out
    res["learned"] = evaluate(Rn, books.cpu().numpy())
    _, b_id = cvt.opq_learn_rotation(sample, M, K, 0, 4, 1234)
    res["identity"] = evaluate(np.eye(D, dtype=np.float32), b_id.cpu().numpy())
    res["random_rotation_of_the_headline"] = evaluate(ctx.R, ctx.books)
    return res


def _sec_sq8(ctx, sec):
    """a-Q / a-T SQ8: train, encode, decode on 2 M x 512-d, each timed directly on its own buffers"""
    torch, cvt = ctx.torch, ctx.cvt
    d3, n = 512, 1 << 21
    g = torch.Generator(device=ctx.dev); g.manual_seed(3)
    feats = torch.randn((n, d3), generator=g, device=ctx.dev).clamp_(min=0)  # "CNN-like": ReLU'd Gaussian
    nb = feats.numel() * 4
    ms_t = _ev_ms(torch, lambda: cvt.sq8_train(feats, l2norm=True), reps=3, warm=1)
    vmin, vdiff = cvt.sq8_train(feats, l2norm=True)
    # encode: the reference normalises x in place (int8_quan.cc:76-78); every timed call gets a buffer of raw rows of
    # its own
    copies = [feats] + [feats.clone() for _ in range(3)]
    it = iter(copies)
    ms_wb = _ev_ms(torch, lambda: cvt.sq8_encode(vmin, vdiff, next(it), l2norm=True), reps=3, warm=1)
    del copies, it
    feats = torch.randn((n, d3), generator=g, device=ctx.dev).clamp_(min=0)
    ms_nw = _ev_ms(torch, lambda: cvt.sq8_encode(vmin, vdiff, feats, l2norm=2), reps=3, warm=1)
    ms_plain = _ev_ms(torch, lambda: cvt.sq8_encode(vmin, vdiff, feats, l2norm=False), reps=3, warm=1)
    codes = cvt.sq8_encode(vmin, vdiff, feats, l2norm=2)
    ms_d = _ev_ms(torch, lambda: cvt.sq8_decode(vmin, vdiff, codes), reps=3, warm=1)
    sec["sq8"] = {"rows": n, "d": d3,
                  "train_l2norm": _hbm(nb, ms_t),
                  "encode_l2norm_no_write_back": _hbm(nb * 1.25, ms_nw),
                  "encode_l2norm_write_back": dict(_hbm(nb * 1.25, ms_wb),
                                                   traffic_GBps=round(nb * 2.25 / (ms_wb * 1e-3) / 1e9, 1)),
                  "encode_no_norm": _hbm(nb * 1.25, ms_plain),
                  "decode": _hbm(nb * 1.25, ms_d),
                  "what": "train: per-dimension min / max-min over L2-normalised rows (4d bytes per row); encode: "
                          "4d in + d "
                          "out per row (algorithmic; the reference's in-place normalisation writes another 4d back: "
                          "traffic_GBps); decode: d in + 4d out.  Each number is timed on its own buffers, no "
                          "subtraction"}
    return vmin, vdiff


def _sec_flat_f32(ctx):
    """a-IP / a-L2F: the reference's brute_force CLI shape (brute_force.cpp:14-19, 86): 1 M x 128-d fp32
    rows, top-100"""
    torch, cvt, args = ctx.torch, ctx.cvt, ctx.args
    n, k, res = 1_000_000, 100, {"rows": 1_000_000, "d": D, "k": 100, "cases": {}}
    rng = np.random.default_rng(9)
    cen = rng.normal(size=(2000, D)).astype(np.float32)
    x = cen[rng.integers(0, 2000, n)] + 0.7 * rng.normal(size=(n, D)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)   # unit-norm features, as the reference indexes them
    q = x[rng.integers(0, n, 1024)] + 0.2 * rng.normal(size=(1024, D)).astype(np.float32)
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    xd, qd = torch.from_numpy(x).to(ctx.dev), torch.from_numpy(q).to(ctx.dev)
    outs = {}
    for metric, tag in ((cvt.IP, "ip"), (cvt.L2F, "l2")):
        ix = cvt.FlatIndex(metric, D)
        ix.add(xd)
        for nq in (1, 64, 128, 1000):
            qq = qd[:nq].contiguous()
            ms, ms5, reps = _ev_ms_steady(torch, lambda: ix.search(qq, k), span_ms=40.0)
            c = {"ms": round(ms, 4), "launches_timed": reps, "ms_over_the_first_5_launches": round(ms5, 4), "queries_per_s": round(nq / (ms * 1e-3), 1),
                 "path": ix.last_search()[0]}
            if c["path"] == 3 and nq <= 256:   # threshold filter, one bf16 product: bound by the first-term plane of the operand copy
                c["roofline"] = _hbm(n * D * 2, ms)
                c["roofline"]["note"] = "algorithmic bytes = the first bf16 term of every row once (the sample pass re-reads a third .. a thirty-second of them, by table size and k)"
            elif nq <= 96:   # one stream over the rows: bound by HBM -- round 6: over the first bf16 terms of the operand copy (2 bytes per value)
                c["roofline"] = _hbm(n * D * 2, ms)
                c["roofline"]["note"] = ("algorithmic bytes = the first bf16 term of every row once (\"flat_f32_packed\": the stream kernels read the threshold "
                                         "filter's operand copy; rounds 3-5 streamed the 4-byte rows: %.3f of the HBM peak on those bytes)" % (n * D * 4 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS))
            else:          # bf16 matrix cores: products per (query, row, dimension)
                prods = 1 if c["path"] == 3 else 3
                tf = prods * 2.0 * nq * n * D / (ms * 1e-3) / 1e12
                c["roofline"] = {"bound": "mfma", "achieved": round(tf, 1), "peak": BF16_MFMA_PEAK_TF,
                                 "unit": "TFLOP/s (bf16, %s)" % ("one product per term over all rows (+ the sample pass over a third .. a thirty-second of them); the whole "
                                                                   "pipeline's time, of which the filter pass is half" if prods == 1 else "three two-term products"),
                                 "frac": round(tf / BF16_MFMA_PEAK_TF, 4)}
            res["cases"]["%s nq=%d" % (tag, nq)] = c
        if metric == cvt.IP:
            outs = {nq: ix.search(qd[:nq].contiguous(), k) for nq in (1, 64)}
        ix.close()
    # widths of the reference's CNN features (pca_online / scalar_quantization: 512 / 1024-d): 1 GB of rows each, 1000 queries
    del xd
    res["widths"] = {"rows_bytes": "1 GB per width (2048-d: 2 GB)", "nq": 1000, "k": k, "cases": {}}
    base = res["cases"].get("ip nq=1000", {}).get("ms")
    for d2, relu in ((512, False), (1024, False), (2048, False), (512, True), (1024, True)):
        # relu: the same rows after a ReLU (non-negative, like post-activation CNN features): every score is positive and the scores of unrelated rows
        # lie close together, which is what widens the filter's margin band (DESIGN 4.4 "tight, non-negative rows")
        n2 = max((1 << 30) // (4 * d2), 262144)   # (2048-d: 2 GB of rows -- the pipeline takes tables from 262 144 rows on)
        g = torch.Generator(device=ctx.dev); g.manual_seed(d2)
        cen2 = torch.randn((2000, d2), generator=g, device=ctx.dev)
        x2 = cen2[torch.randint(0, 2000, (n2,), generator=g, device=ctx.dev)] + 0.7 * torch.randn((n2, d2), generator=g, device=ctx.dev)
        if relu:
            x2.clamp_(min=0)
        x2 = x2 / x2.norm(dim=1, keepdim=True)
        q2 = x2[torch.randint(0, n2, (1000,), generator=g, device=ctx.dev)] + 0.2 * torch.randn((1000, d2), generator=g, device=ctx.dev)
        if relu:
            q2.clamp_(min=0)
        q2 = (q2 / q2.norm(dim=1, keepdim=True)).contiguous()
        ix = cvt.FlatIndex(cvt.IP, d2)
        ix.add(x2)
        ms = _ev_ms(torch, lambda: ix.search(q2, k), reps=3, warm=2)
        gb2 = n2 * d2 * 4 / float(1 << 30)
        c = {"rows": n2, "ms": round(ms, 4), "path": ix.last_search()[0], "ms_per_GB_of_rows": round(ms / gb2, 4)}
        if base:
            c["vs_128d_per_byte"] = round((ms / gb2) / (base / (n * D * 4 / float(1 << 30))), 3)
        cvt.set_tuning("flat_variant", 1)
        c["exact_kernels_ms"] = round(_ev_ms(torch, lambda: ix.search(q2, k), reps=1, warm=1), 3)
        cvt.set_tuning("flat_variant", 0)
        res["widths"]["cases"]["ip %dd%s" % (d2, " relu" if relu else "")] = c
        ix.close()
        del x2, q2
    res["path_codes"] = ("0 exact kernels, 1 sample + matrix-core filter pipeline, 2 one stream over the rows "
                         "(flat_f32_stream.hip), 3 threshold filter (flat_f32_tfilter.hip, round 6)")
    if args.cpu_sample > 0:
        from oracle import binding as ob
        if ob.ref_available():
            cs = 16
            rd, rl, t_add, t_cpu = ob.RefFlat().ip_search_timed(x, q[:cs], k)
            gd, gi = outs[64]
            res["cpu_baseline"] = {
                "value": round(cs / t_cpu, 2), "unit": "queries/s", "cores": 1, "kind": "reference",
                "sample": "%d queries, the reference's own BruteforceSearch<float>::searchKnn + InnerProductSpace "
                          "(brutoforce.hpp:73-93) compiled in place (oracle/_ref/libref_bf_ip.so), all %d rows; "
                          "addPoint of "
                          "the rows took %.1f s" % (cs, n, t_add),
                "gpu_labels_identical": bool(np.array_equal(rl, gi[:cs].cpu().numpy())),
                "gpu_distances_bit_identical": bool(np.array_equal(rd.view(np.uint32),
                                                                   gd[:cs].cpu().numpy().view(np.uint32)))}
    return res


def _sec_flat_u8_c3(ctx, vmin, vdiff):
    """config 3: 10 M x 512-d uint8 codes (SQ8 of CNN-like features), brute-force L2 top-10"""
    torch, cvt, args = ctx.torch, ctx.cvt, ctx.args
    d3, n3, k3 = 512, 10_000_000, 10
    g = torch.Generator(device=ctx.dev)
    flat = cvt.FlatIndex(cvt.L2U8, d3)
    chunk = 1 << 20
    rows_h = np.empty((n3, d3), dtype=np.uint8) if args.cpu_sample > 0 else None  # host copy for the CPU baseline only
    for a in range(0, n3, chunk):
        b = min(n3, a + chunk)
        g.manual_seed(1000 + a // chunk)
        f = torch.randn((b - a, d3), generator=g, device=ctx.dev).clamp_(min=0)
        c = cvt.sq8_encode(vmin, vdiff, f, l2norm=True)
        flat.add(c)
        if rows_h is not None:
            rows_h[a:b] = c.cpu().numpy()
    g.manual_seed(77)
    qf = torch.randn((4096, d3), generator=g, device=ctx.dev).clamp_(min=0)
    q3 = cvt.sq8_encode(vmin, vdiff, qf, l2norm=True)
    c3 = {"rows": n3, "d": d3, "k": k3, "cases": {}}
    outs = {}
    for nq3 in (1, 8, 64, 512, 1000, 4096):
        qq = q3[:nq3].contiguous()
        ms = _ev_ms(torch, lambda: flat.search(qq, k3), reps=3, warm=1)
        outs[nq3] = flat.search(qq, k3)
        ops = 2.0 * nq3 * n3 * d3 / (ms * 1e-3)
        c = {"ms": round(ms, 4), "queries_per_s": round(nq3 / (ms * 1e-3), 1), "path": int(flat.last_search()[0])}
        if nq3 <= 128:   # one stream over the rows (flat_u8_mstream_kernel): bound by HBM
            c["roofline"] = _hbm(n3 * d3, ms)
        else:
            c["roofline"] = {"bound": "mfma", "achieved": round(ops / 1e12, 1), "peak": I8_MFMA_PEAK_TOPS,
                             "unit": "TOP/s (int8, 2 per MAC)", "frac": round(ops / 1e12 / I8_MFMA_PEAK_TOPS, 4),
                             "peak_measured": I8_MFMA_MEASURED_TOPS,
                             "frac_of_measured_peak": round(ops / 1e12 / I8_MFMA_MEASURED_TOPS, 4)}
        c3["cases"]["nq=%d" % nq3] = c
    c3["path_codes"] = ("0 streaming passes over the raw rows / exact kernels, 1 sample + matrix-core filter pipeline (round 2-5), 4 threshold filter over the int8 "
                        "operand copy (flat_u8_tfilter.hip, round 6: from 129 queries, and every batch with k > 128); the matrix-core rooflines count the "
                        "algorithmic operations, 2 x queries x rows x d (the filter's sample pass re-scores 1/32 .. 1/3 of the rows on top)")
    # more neighbours (round 6): the stream kernels stop at k = 128, the pipeline of rounds 2-5 at 64; behind them the exact kernels took one
    # query per workgroup
    qq = q3[:1000].contiguous()
    c3["more_neighbours"] = {"nq": 1000, "cases": {}}
    for kk in (100, 129, 1000):
        ms = _ev_ms(torch, lambda: flat.search(qq, kk), reps=3, warm=1)
        c = {"ms": round(ms, 4), "queries_per_s": round(1000 / (ms * 1e-3), 1), "path": int(flat.last_search()[0]),
             "frac_of_measured_i8_peak": round(2.0 * 1000 * n3 * d3 / (ms * 1e-3) / 1e12 / I8_MFMA_MEASURED_TOPS, 4)}
        if kk in (100, 1000):
            got = flat.search(qq, kk)
            cvt.set_tuning("flat_u8_tfilter", 0)
            try:
                c["round_5_paths_ms"] = round(_ev_ms(torch, lambda: flat.search(qq, kk), reps=1, warm=1 if kk <= 128 else 0), 3)
                ref = flat.search(qq, kk)
            finally:
                cvt.set_tuning("flat_u8_tfilter", 1)
            c["identical"] = bool(torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]))
        c3["more_neighbours"]["cases"]["k=%d" % kk] = c
    if args.cpu_sample > 0:
        from oracle import binding as ob
        orc = ob.Oracle(o3=True)
        cs = 4
        qh = q3[:cs].cpu().numpy()
        t0 = time.perf_counter()
        _, od, oi = orc.flat_search(ob.L2U8, rows_h, qh, k3)
        t_cpu = time.perf_counter() - t0
        gd, gi = outs[1000]
        c3["cpu_baseline"] = {
            "value": round(cs / t_cpu, 3), "unit": "queries/s", "cores": 1, "kind": "port",
            "sample": "%d queries against all %d rows (oracle L2SqrI loop, -O3, 1 thread)" % (cs, n3),
            "gpu_topk_ids_identical": bool(np.array_equal(oi, gi[:cs].cpu().numpy())),
            "gpu_distances_identical": bool(np.array_equal(np.asarray(od).astype(np.int64),
                                                           gd[:cs].cpu().numpy().astype(np.int64)))}
    flat.close()
    return c3


def _ivf_query(ctx):
    """f-4: the reference's own Query shape -- coarseK = 8192 lists, nk = 3 probes, per-video minimum scores"""
    torch, cvt, dev = ctx.torch, ctx.cvt, ctx.dev
    L, nk, n, n_videos, nq = 8192, 3, 1 << 20, 4096, 10_000
    g = torch.Generator(device=dev); g.manual_seed(11)
    cen = torch.randn((L, D), generator=g, device=dev) * 0.08
    x = cen[torch.randint(0, L, (n,), generator=g, device=dev)] + 0.03 * torch.randn((n, D), generator=g, device=dev)
    q = x[torch.randint(0, n, (nq,), generator=g, device=dev)] + 0.01 * torch.randn((nq, D), generator=g, device=dev)
    ix = cvt.OpqIndex(cen.cpu().numpy(), (ctx.books * 0.3).astype(np.float32))
    ms_enc = _ev_ms(torch, lambda: ix.encode(x), reps=2, warm=1)
    # the reference builds its index one video at a time: IVFOPQ::Add of a few hundred frames (IVFOPQ.cpp:135-163) -- wall time of one
    # such call through the host-pointer entries (coarse assignment + PQ encode, then the append)
    xv = x[:300].cpu().numpy(); vid = np.zeros(300, np.int32)
    tmp_ix = cvt.OpqIndex(cen.cpu().numpy(), (ctx.books * 0.3).astype(np.float32))
    for _ in range(3):
        lv, cv = tmp_ix.encode(xv); tmp_ix.add_codes(cv, lv, vid)
    t0 = time.perf_counter()
    for _ in range(20):
        lv, cv = tmp_ix.encode(xv); tmp_ix.add_codes(cv, lv, vid)
    ms_video = (time.perf_counter() - t0) / 20 * 1e3
    tmp_ix.close()
    lists, codes = ix.encode(x)
    ix.add_codes(codes, lists, torch.randint(0, n_videos, (n,), generator=g, device=dev, dtype=torch.int32))
    # the first query builds the list-ordered copy
    ms_first = _ev_ms(torch, lambda: ix.query_video(q[:64].contiguous(), nk, n_videos, rotate=False), reps=1, warm=0)
    ms_q = _ev_ms(torch, lambda: ix.query_video(q, nk, n_videos, rotate=False), reps=3, warm=1)
    ms_9 = _ev_ms(torch, lambda: ix.query_video(q[:9].contiguous(), nk, n_videos, rotate=False), reps=5, warm=1)
    ix.close()
    return {"entries": n, "lists": L, "nprobe": nk, "videos": n_videos,
            "encode_rows_per_s": round(n / (ms_enc * 1e-3), 1),
            "encode_what": "coarse argmin over 8192 centroids (matrix-core filter + exact resolution) + PQ encode",
            "add_one_video_300_frames_ms": round(ms_video, 3),
            "add_one_video_what": "cvtmi_opq_encode + cvtmi_opq_add_codes with host pointers on 300 frames (the reference's Add of one video; 7.9 ms "
                                  "before the small-call dispatch of round 5)",
            "first_query_ms_incl_list_build": round(ms_first, 3), "frames": nq, "ms": round(ms_q, 3),
            "frames_per_s": round(nq / (ms_q * 1e-3), 1), "ms_9_frames": round(ms_9, 3),
            "what": "IVFOPQ::Query semantics (IVFOPQ.cpp:213-320): coarse top-3 of 8192, residual tables, list scans, "
                    "per-video min clamped at 1.0; dense [frames][videos] score matrix out"}


def _hnsw_c5(ctx):
    """config 5: HNSW graph in HBM, batched 10 K queries, fp32 vectors and OPQ codes"""
    torch, cvt, args, dev = ctx.torch, ctx.cvt, ctx.args, ctx.dev
    n, nq, Mg, efc = int(args.hnsw_nodes), 10_000, 32, 80
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
    # the graph is built on the host, as in the reference: hnsw_build with all hardware threads = addPoints, the reference's locked
    # parallel insertion (hnswalg.h:594-608); one thread (the byte-identical file) is what tests/test_host_hnsw_build.py pins
    subprocess.run([os.path.join(ROOT, "cvt_amd", "bin", "hnsw_build"), rows_p, str(D), str(Mg), str(efc), idx_p, "ip", "-", "0"],
                   check=True, capture_output=True)
    t_build = time.perf_counter() - t0
    ix = cvt.HnswIndex(open(idx_p, "rb").read(), cvt.IP, D)
    qd = torch.from_numpy(q).to(dev)
    xd = torch.from_numpy(x).to(dev)
    exact = torch.cat([torch.argmax(qd[a:a + 500] @ xd.T, dim=1) for a in range(0, nq, 500)])
    del xd
    res = {"nodes": n, "d": D, "M": Mg, "ef_construction": efc, "nq": nq, "graph_build_s_host": round(t_build, 1),
           "graph_build_threads": _usable_cpus(),
           "what": "one wave per query over a graph built on the host (hnsw_build CLI: the reference's addPoint under its own "
                   "lock discipline on every hardware thread, ids and levels in row order, "
                   "M=32 efC=80, makeIdx.cpp:303-304)", "fp32": {}, "adc": {}}

    def r1(lab):
        return round(float((lab[:, 0] == exact).float().mean().item()), 4)

    def rk(lab):
        return round(float((lab == exact[:, None]).any(dim=1).float().mean().item()), 4)
    for kk, ef in ((5, 1000), (10, 64)):
        ms = _ev_ms(torch, lambda: ix.search(qd, kk, ef), reps=2, warm=1)
        _, lab = ix.search(qd, kk, ef)
        res["fp32"]["ef=%d" % ef] = {"ms": round(ms, 3), "queries_per_s": round(nq / (ms * 1e-3), 1), "k": kk,
                                     "recall_at_1": r1(lab)}
    tmp = cvt.OpqIndex(np.zeros((1, D), np.float32), np.zeros((16, 256, D // 16), np.float32), R=ctx.R)
    xr = tmp.rotate(torch.from_numpy(x).to(dev))
    _, books5 = cvt.opq_train(xr[:50_000].contiguous(), 1, 16, 256, 8, 1)
    opq = cvt.OpqIndex(np.zeros((1, D), np.float32), books5.cpu().numpy(), R=ctx.R)
    _, codes = opq.encode(xr)
    opq.add_codes(codes)
    for kk, ef in ((5, 1000), (10, 64)):
        ms = _ev_ms(torch, lambda: ix.search_adc(opq, qd, kk, ef), reps=2, warm=1)
        _, lab = ix.search_adc(opq, qd, kk, ef)
        res["adc"]["ef=%d" % ef] = {"ms": round(ms, 3), "queries_per_s": round(nq / (ms * 1e-3), 1), "k": kk,
                                    "recall_at_1": r1(lab),
                                    "recall_at_k": rk(lab)}
        ms = _ev_ms(torch, lambda: ix.search_adc_rerank(opq, qd, kk, ef), reps=2, warm=1)
        _, lab = ix.search_adc_rerank(opq, qd, kk, ef)
        res["adc"]["ef=%d+rerank" % ef] = {"ms": round(ms, 3), "queries_per_s": round(nq / (ms * 1e-3), 1), "k": kk,
                                           "recall_at_1": r1(lab)}
    if args.cpu_sample > 0:
        from oracle import binding as ob
        if ob.ref_available():
            rh = ob.RefHnsw()
            cs = 200
            t0 = time.perf_counter(); rd, rl = rh.search(0, D, idx_p, q[:cs], 5, 1000); t_cpu = time.perf_counter() - t0
            gd, gl = ix.search(qd[:cs].contiguous(), 5, 1000)
            res["cpu_baseline"] = {
                "value": round(cs / t_cpu, 1), "unit": "queries/s", "cores": 1, "kind": "reference",
                "sample": "%d of the queries, the reference's own searchKnn (oracle/_ref/libref_hnsw.so), ef=1000, "
                          "same "
                          "graph file" % cs,
                "gpu_labels_identical": bool(np.array_equal(rl, gl.cpu().numpy())),
                "gpu_distances_bit_identical": bool(np.array_equal(rd.view(np.uint32),
                                                                   gd.cpu().numpy().view(np.uint32)))}
    ix.close(); opq.close(); tmp.close()
    for p in (rows_p, idx_p):
        if os.path.exists(p):
            os.remove(p)
    os.rmdir(tmpd)
    return res


def _usable_cpus():
    """hardware threads this process can really use: affinity mask and container CPU quota (the MI355X boxes show 256 logical CPUs and
    cgroup cpu.max = 16 CPUs; the same rule as HierarchicalNSW::usable_threads)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except Exception:
        try:
            qv = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); pv = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if qv > 0 and pv > 0:
                n = min(n, max(1, -(-qv // pv)))
        except Exception:
            pass
    return n


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip() + " (%d logical cores, %d usable under the container's CPU quota)" % (
                    os.cpu_count(), _usable_cpus())
    except Exception:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
