// api.hip -- the C ABI of libcvtmi (include/cvtmi.h): handles, HBM residency, host<->device
// staging for the host-pointer entry points, and dispatch to the HIP kernels.  No compute happens
// on the CPU here; without a HIP device every entry fails with CVTMI_EHIP.
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <shared_mutex>
#include <functional>
#include <mutex>
#include <new>
#include <numeric>
#include <thread>
#include <vector>

#include "host_util.h"
#include "kernels.h"
#include "shard.h"

namespace cvtmi {

static int host_spin_default()
{
    const char *e = getenv("CVTMI_HOST_SPIN_US");   // (measurement aid: the CLIs have no tuning switch)
    return e ? atoi(e) : 200;
}
std::atomic<int> g_host_spin_us{host_spin_default()};   // host_util.h: stream_wait
static thread_local std::string g_err;

void set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

}  // namespace cvtmi

using namespace cvtmi;

// ================================================================ handles =====================
// per-call scratch of an OPQ search: rotated queries, tables (fp32 + the quantised images of adc_scan16h), partial lists, shared
// bounds, spill areas, the item table and the plan it was built from, the staging of the host-pointer entry.  The set remembers
// the stream it was last used on and an event recorded when that call returned: the next lessee on ANOTHER stream waits first.
struct OpqScratch {
    DevBuf s_qrot, s_part_d, s_part_id, s_lut, s_gthr, s_qlut, s_qp, s_spill, s_items, s_probe;
    ScanHPlan hplan;                       // the item table s_items holds ...
    int64_t hplan_n = -1, hplan_nq = -1;   // ... and the (rows, queries, forced splits, planner settings) it was built for
    int hplan_splits = 0, hplan_key = 0;
    DevBuf io_q, io_d, io_i;   // device side of the host-pointer search (cvtmi_opq_search)
    PinBuf io_pin;             // its pinned staging area
    hipStream_t own = nullptr; // stream of the host-pointer entry (created on first use)
    hipEvent_t done = nullptr;
    hipStream_t last = nullptr;
    bool pending = false, busy = false;
    void release_all()
    {
        for (DevBuf *b : { &s_qrot, &s_part_d, &s_part_id, &s_lut, &s_gthr, &s_qlut, &s_qp, &s_spill, &s_items, &s_probe, &io_q, &io_d, &io_i }) b->release();
        io_pin.release();
        if (own) (void)hipStreamDestroy(own);
        if (done) (void)hipEventDestroy(done);
        own = nullptr; done = nullptr;
    }
};

struct cvtmi_opq_s {
    int device = 0;
    HandleSync sync;
    OpqModelDev m{};
    float *d_coarse = nullptr, *d_books = nullptr, *d_R = nullptr;
    int32_t *d_perm = nullptr;
    // resident entries, insertion order
    DevBuf codes, lists, videos;
    DevBuf codes_rot;     // M = 16: rows rotated by (row & 15) bytes for adc_scan16q, built lazily at search time
    int64_t rot_n = 0;    // rows of codes_rot that are up to date
    DevBuf codes16;       // M < 16: the rows padded to 16 bytes with zeros, what the M = 16 scan kernels read (opq_pads; built lazily like codes_rot, which is then its rotation)
    int64_t pad_n = 0;    // rows of codes16 that are up to date
    int rot_kind = 0;     // what codes_rot holds: 0 = 16-byte rows (M = 16, or the padded copy) rotated by row & 15; 1 = the packed rotation of an M = 8 / 4 index (adc_scan_p.hip)
    int64_t n = 0;
    bool has_lists = false, has_videos = false;
    int64_t id_base = 0;
    // list-ordered (CSR) copy for the per-video query path, built lazily
    bool csr_valid = false;
    DevBuf csr_codes, csr_videos, csr_off, csr_scratch, csr_stats;
    int64_t csr_kept = 0, csr_longest = 0;  // entries in the CSR copy (list ids outside [0, coarseK) are dropped), longest list
    int32_t csr_vmin = 0, csr_vmax = -1;    // range of the video ids it holds
    // scratch of the calls that run one at a time (query_video: probe lists, rotated queries)
    DevBuf s_qrot, s_probe, s_rot;
    // Searches (cvtmi_opq_search*) run CONCURRENTLY, as the reference's QueryThrehold de facto may (opq/src/IVFOPQ.cpp:322-422 only
    // reads the index): each leases a scratch set from this pool for the duration of the call (OpqLease) and holds `rw` shared;
    // everything else -- add / reset / reserve, the lazily built copies of the rows, the one-at-a-time entries above -- holds it
    // exclusively (OpqExclusive, on top of the per-handle Serial that orders those calls among themselves).
    std::shared_timed_mutex rw;
    std::mutex pool_mu;
    std::vector<struct OpqScratch *> pool;
    hipEvent_t mutated = nullptr;   // recorded on the stream of the last exclusive call: searches on other streams wait for it
    hipStream_t mut_stream = nullptr;
    bool mut_pending = false;
    // tuning / measurement
    int p_splits = 0, p_qtile = 0, p_profile = 0, p_variant = 7;
    int p_encode = 0;  // 0 = choose, 1 = VALU encode, 2 = matrix-core filter + exact resolution
    int p_prerot = 1;  // adc_scan16q reads a pre-rotated copy of the code rows (+16 bytes of HBM per row)
    int p_tail = 1, p_groups_a = 0, p_splits_b = 0;  // two-region scan plan: on / forced shape (tests)
    int p_lazy = 1, p_share = 1;  // adc_scan16q: lazy selection between checkpoints; row splits share their thresholds
    int p_small = 1;              // 1 .. 8 queries take the small-batch path (adc_scan_h.hip) when the library chooses the scan (scan_variant 7)
    static constexpr int kEvRing = 64;
    hipEvent_t ev0[kEvRing] = {}, ev1[kEvRing] = {};
    int ev_count = 0;  // scan launches recorded since the last cvtmi_opq_last_scan
    int64_t last_bytes = 0;
    int last_qt = 0, last_splits = 0;
};

// per-call scratch of a flat search.  A handle keeps a small pool of these: a search leases one for the duration of the call, so
// searches on one handle overlap -- on the host (several threads inside the library) and on the device (several streams).  The
// set remembers the stream it was last used on and an event recorded when that call returned: the next lessee on ANOTHER
// stream waits for the event first.
struct FlatScratch {
    DevBuf s_part_d, s_part_id, s_gthr, s_stage;
    DevBuf f_stats, f_thr, f_marg, f_cnt, f_cand, f_sd, f_si, f_sd2, f_si2, f_seld, f_seli;   // matrix-core filter pipelines
    DevBuf fs_redo, fs_scratch;                                                                // fp32 stream
    DevBuf io_q, io_d, io_i;                                                                   // staging of the host-pointer entry
    PinBuf io_pin;                  // small calls: [queries | distances | labels] in page-locked memory the kernels write into
    hipStream_t own = nullptr;      // stream of the host-pointer entry (created on first use)
    hipEvent_t done = nullptr;
    hipStream_t last = nullptr;
    bool pending = false, busy = false;
    void release_all()
    {
        for (DevBuf *b : { &s_part_d, &s_part_id, &s_gthr, &s_stage, &f_stats, &f_thr, &f_marg, &f_cnt, &f_cand, &f_sd, &f_si, &f_sd2,
                           &f_si2, &f_seld, &f_seli, &fs_redo, &fs_scratch, &io_q, &io_d, &io_i })
            b->release();
        io_pin.release();
        if (own) (void)hipStreamDestroy(own);
        if (done) (void)hipEventDestroy(done);
        own = nullptr; done = nullptr;
    }
};

struct cvtmi_flat_s {
    int device = 0;
    // searches hold `rw` shared, everything that changes the index (add, reset, the lazily built operand copies) exclusively
    std::shared_timed_mutex rw;
    std::mutex pool_mu;
    std::vector<FlatScratch *> pool;
    hipEvent_t mutated = nullptr;   // recorded on the stream of the last mutation: searches on other streams wait for it
    hipStream_t mut_stream = nullptr;
    bool mut_pending = false;
    int metric = 0, D = 0;
    size_t row_bytes = 0;
    DevBuf data, labels, norms;  // norms: int32 |x-128|^2 per row, uint8 metric with D % 32 == 0 (MFMA path)
    DevBuf add_stage;            // staging of host rows on their way into the blocked layout
    int64_t n = 0;
    int64_t id_base = 0;   // row r reports label id_base + r while labels are implicit (row shards, cvtmi_flat_set_id_base)
    bool identity = true;  // label == row
    // matrix-core filter of the fp32 search (flat_mfma.hip): bf16 operand copy of the rows, built on first use
    DevBuf f_pack, f_bias, f_istats;   // f_istats: [0] max |x|^2, [1] rows with a non-finite value (of the operand copy)
    int64_t f_pack_n = -1;      // rows the copy covers (-1: none)
    DevBuf f_rows;              // fp32: row-major copy of the rows for the threshold filter's exact finish ("flat_f32_rows_copy"; the blocked layout gathers 16 of every 128 bytes it fetches)
    int64_t f_rows_n = -1;      // rows it covers (-1: none)
    bool f_rows_failed = false; // it did not fit once: not tried again on this handle
    int f_pack_nch = 0;         // its K steps per row (the threshold filter of a width between two kernels pads with zeros)
    bool f_nonfinite = false;   // a row holds inf / NaN: the filter is not used
    std::atomic<int> f_last_filtered{0};    // how the last search was answered (0 exact, 1 filter pipeline, 2 fp32 stream, 3 fp32 threshold filter)
    std::atomic<long long> f_last_worst{0};  // its largest candidate list
    // fp32 stream (flat_f32_stream.hip): per-row score bias, statistics of the rows ([0] max |x|^2, [1] non-finite rows)
    DevBuf fs_bias, fs_stats;
    int64_t fs_stats_n = -1;    // index size the host copy of the statistics belongs to
    bool fs_nonfinite = false;
};

// a scratch set for the duration of one call on stream st (nullptr + host = true: the set's own stream)
struct FlatLease {
    cvtmi_flat_s *h;
    FlatScratch *s = nullptr;
    hipStream_t st;
    int open(cvtmi_flat_s *handle, hipStream_t stream, bool host)
    {
        h = handle; st = stream;
        {
            std::lock_guard<std::mutex> g(h->pool_mu);
            FlatScratch *any = nullptr;
            for (FlatScratch *c : h->pool) {
                if (c->busy) continue;
                if (!host && c->pending && c->last == stream) { s = c; break; }   // same stream as before: nothing to wait for
                if (!any) any = c;
            }
            if (!s) s = any;
            if (!s) {
                s = new (std::nothrow) FlatScratch();
                if (!s) return fail(CVTMI_ENOMEM, "flat search: out of host memory");
                h->pool.push_back(s);
            }
            s->busy = true;
        }
        if (host) {
            if (!s->own && hipStreamCreateWithFlags(&s->own, hipStreamNonBlocking) != hipSuccess) { close(false); return fail(CVTMI_EHIP, "hipStreamCreate failed"); }
            st = s->own;
        }
        if (s->pending && s->last != st) (void)hipStreamWaitEvent(st, s->done, 0);
        if (h->mut_pending && h->mut_stream != st) (void)hipStreamWaitEvent(st, h->mutated, 0);
        return CVTMI_OK;
    }
    void close(bool used = true)
    {
        if (!s) return;
        if (used) {
            if (!s->done) (void)hipEventCreateWithFlags(&s->done, hipEventDisableTiming);
            if (s->done && hipEventRecord(s->done, st) == hipSuccess) { s->last = st; s->pending = true; }
        }
        std::lock_guard<std::mutex> g(h->pool_mu);
        s->busy = false;
        s = nullptr;
    }
    ~FlatLease() { close(); }
};

static int use_device(int dev)
{
    int cur = -1;
    CVTMI_HIP(hipGetDevice(&cur));
    if (cur != dev) CVTMI_HIP(hipSetDevice(dev));
    return CVTMI_OK;
}

#define CHECK_H(h) \
    if (!(h)) return fail(CVTMI_EINVAL, "%s: null handle", __func__); \
    CVTMI_TRY(use_device((h)->device))
// + the handle's lock and stream ordering (see Serial) for the rest of the enclosing scope
#define CHECK_H_SERIAL(h, stream) \
    CHECK_H(h); \
    Serial serial_##h((h)->sync, (hipStream_t)(stream)); \
    OpqExclusive excl_##h((h), (hipStream_t)(stream))

// Pure reads of the MODEL (rotation matrix / permutation, codebooks: immutable after cvtmi_opq_create) into the caller's own buffers:
// shared, like a search -- they neither drain the searches in flight nor make later searches wait on their stream.  Entries that hold
// this must not call each other (the shared lock is not recursive): they share the *_impl / launch_* functions instead.
#define CHECK_H_SHARED(h) \
    CHECK_H(h); \
    std::shared_lock<std::shared_timed_mutex> rd_##h((h)->rw)

// a scratch set of an OPQ handle for the duration of one search on stream st (nullptr + host = true: the set's own stream).
// The caller holds h->rw shared.
struct OpqLease {
    cvtmi_opq_s *h = nullptr;
    OpqScratch *s = nullptr;
    hipStream_t st = nullptr;
    bool used = false;
    int open(cvtmi_opq_s *handle, hipStream_t stream, bool host)
    {
        h = handle; st = stream;
        {
            std::lock_guard<std::mutex> g(h->pool_mu);
            OpqScratch *any = nullptr;
            for (OpqScratch *c : h->pool) {
                if (c->busy) continue;
                if (!host && c->pending && c->last == stream) { s = c; break; }   // same stream as before: nothing to wait for
                if (!any) any = c;
            }
            if (!s) s = any;
            if (!s) {
                s = new (std::nothrow) OpqScratch();
                if (!s) return fail(CVTMI_ENOMEM, "opq search: out of host memory");
                h->pool.push_back(s);
            }
            s->busy = true;
        }
        if (host) {
            if (!s->own && hipStreamCreateWithFlags(&s->own, hipStreamNonBlocking) != hipSuccess) { close(); return fail(CVTMI_EHIP, "hipStreamCreate failed"); }
            st = s->own;
        }
        if (s->pending && s->last != st) (void)hipStreamWaitEvent(st, s->done, 0);
        if (h->mut_pending && h->mut_stream != st) (void)hipStreamWaitEvent(st, h->mutated, 0);
        used = true;
        return CVTMI_OK;
    }
    void close()
    {
        if (!s) return;
        if (used) {
            if (!s->done) (void)hipEventCreateWithFlags(&s->done, hipEventDisableTiming);
            if (s->done && hipEventRecord(s->done, st) == hipSuccess) { s->last = st; s->pending = true; }
        }
        std::lock_guard<std::mutex> g(h->pool_mu);
        s->busy = false;
        s = nullptr;
    }
    ~OpqLease() { close(); }
    OpqLease() = default;
    OpqLease(const OpqLease &) = delete;
    OpqLease &operator=(const OpqLease &) = delete;
};

// the exclusive side: taken by the outermost of the (Serial-ordered) non-search calls of an OPQ handle; the stream first waits for
// the searches that are still in flight on other streams, and the searches that follow wait for this call
struct OpqExclusive {
    cvtmi_opq_s *h;
    hipStream_t st;
    bool own = false;
    OpqExclusive(cvtmi_opq_s *handle, hipStream_t stream) : h(handle), st(stream)
    {
        if (h->sync.depth != 1) return;   // an inner call of this thread: the outermost one holds the lock
        h->rw.lock();
        own = true;
        std::lock_guard<std::mutex> g(h->pool_mu);
        for (OpqScratch *c : h->pool)
            if (c->pending && c->last != st) (void)hipStreamWaitEvent(st, c->done, 0);
        if (h->mut_pending && h->mut_stream != st) (void)hipStreamWaitEvent(st, h->mutated, 0);
    }
    ~OpqExclusive()
    {
        if (!own) return;
        if (!h->mutated) (void)hipEventCreateWithFlags(&h->mutated, hipEventDisableTiming);
        if (h->mutated && hipEventRecord(h->mutated, st) == hipSuccess) { h->mut_stream = st; h->mut_pending = true; }
        h->rw.unlock();
    }
    OpqExclusive(const OpqExclusive &) = delete;
    OpqExclusive &operator=(const OpqExclusive &) = delete;
};

// true when p points into page-locked host memory the device can reach (cvtmi_host_alloc, or the caller's own hipHostMalloc /
// hipHostRegister): such buffers are handed to the copy engines as they are, without the staging copy
static bool host_pinned(const void *p)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

static std::atomic<int> g_hnsw_slots_cap{0};   // cvtmi_set_tuning("hnsw_slots"): cap on traversals per CU (0 = what LDS allows, at most 32)
static std::atomic<int> g_small_zero_copy{1};   // cvtmi_set_tuning("opq_small_zero_copy"): 1 .. 8-query host-pointer searches read / write the pinned staging area from the kernels
static std::atomic<int64_t> g_scans_max_work{(int64_t)48 << 20};   // cvtmi_set_tuning("scans_max_work"): rows x query groups up to which the OPQ small-batch form answers (scans_chosen)
static std::atomic<int> g_scan_bigk{1};          // cvtmi_set_tuning("scan_bigk"): 0 = k > 128 on the exact kernels only (one query per workgroup: rounds 4-5), 1 = the filter pipeline
static std::atomic<int> g_scan_packed{1};        // cvtmi_set_tuning("scan_packed_m"): 0 = M = 8 / 4 through the padded rows like every other M < 16 (round 5), 1 = adc_scan16p
static std::atomic<int> g_scan_pad{1};           // cvtmi_set_tuning("scan_pad_m"): 0 = an OPQ index with M < 16 stays on the row-per-lane scan kernels (opq_pads)
static std::atomic<int> g_sq8_host_small{1};     // cvtmi_set_tuning("sq8_host_small"): small SQ8 host-pointer calls run out of a page-locked scratch area (Sq8HostScratch)
static std::atomic<int> g_flat_f32_rows_copy{4};   // cvtmi_set_tuning("flat_f32_rows_copy"): narrowest fp32 row that gets a row-major copy beside the blocked rows once the threshold filter
                                                  // answers on the handle (0 = never): + 4 D bytes per row, the exact finish reads whole cache lines
static std::atomic<int> g_flat_u8_filter_min_nq{129};            // cvtmi_set_tuning("flat_u8_filter_min_nq" / "_min_rows" / "_min_work"): smallest batch, table and
static std::atomic<int64_t> g_flat_u8_filter_min_rows{524288};   // rows x width x queries (in 1e9) the dispatch hands to the uint8 sample + filter pipeline
static std::atomic<int64_t> g_flat_u8_filter_min_work{130};
static std::atomic<int> g_flat_u8_sample_passes{10};  // cvtmi_set_tuning("flat_u8_sample_passes"): the uint8 filter pipeline's sample goes through the streaming kernel up to this many 128-query passes
static std::atomic<int> g_flat_small_zero_copy{1};   // cvtmi_set_tuning("flat_small_zero_copy"): small host-pointer flat searches write their lists into pinned memory from the kernels
static std::atomic<int> g_host_zero_copy{1};   // cvtmi_set_tuning("opq_host_zero_copy"): page-locked result arrays are written by the kernels themselves, the batch is not cut
static std::atomic<int> g_host_chunks{4096};  // cvtmi_set_tuning("opq_host_chunk"): queries per piece of a pipelined host-pointer OPQ batch (0 = one piece)
static std::atomic<int> g_scanh_key{0};  // bumped when a planner setting of adc_scan16h changes: cached item tables are rebuilt
static std::atomic<int> g_inject_failure{-1};  // cvtmi_set_tuning("comm_inject_failure", r): the local search of rank r of a sharded search fails (tests)
static std::atomic<int> g_flat_variant{0};  // cvtmi_set_tuning("flat_variant"): 0 = choose, 1 = exact kernels only, 2 = matrix-core filter wherever it applies
static std::atomic<int> g_flat_f32_stream{1};  // cvtmi_set_tuning("flat_f32_stream"): 0 = off, 1 = choose, 2 = wherever it applies

static int sharded_local_failure(cvtmi_comm_t c)
{
    if (const int inj = g_inject_failure.load(); inj >= 0 && inj == comm_rank(c)) return fail(CVTMI_ESTATE, "injected failure of rank %d (comm_inject_failure)", comm_rank(c));
    return CVTMI_OK;
}

// One process, every GPU: handles[d] holds the row block of device d (its id base set), comms = cvtmi_comm_create_all.  The
// queries go up to every device, the local searches are enqueued device after device (they run side by side), the all-gathers
// leave as one group, the merge runs on the first device.
using ShardLocalSearch = std::function<int(int, const void *, float *, int64_t *)>;
static int sharded_all(cvtmi_comm_t *comms, int ndev, const void *q, size_t q_bytes, int64_t nq, int k, void *dist, int64_t *ids,
                       const int *devices, const ShardLocalSearch &local_search)
{
    std::vector<Tmp> dq(ndev);
    std::vector<int> status(ndev, CVTMI_OK);
    for (int d = 0; d < ndev; ++d) {
        CVTMI_HIP(hipSetDevice(devices[d]));
        float *sd = nullptr;
        int64_t *si = nullptr;
        int rc = sharded_local_failure(comms[d]);
        if (rc == CVTMI_OK) rc = dq[d].alloc(q_bytes);
        if (rc == CVTMI_OK && hipMemcpyAsync(dq[d].p, q, q_bytes, hipMemcpyHostToDevice, nullptr) != hipSuccess) rc = fail(CVTMI_EHIP, "query upload to device %d failed", devices[d]);
        if (rc == CVTMI_OK) rc = comm_local_slot(comms[d], nq, k, &sd, &si);
        if (rc == CVTMI_OK) rc = local_search(d, dq[d].p, sd, si);
        status[d] = rc;
    }
    CVTMI_HIP(hipSetDevice(devices[0]));
    Tmp dd, di;
    CVTMI_TRY(dd.alloc((size_t)nq * k * 4));
    CVTMI_TRY(di.alloc((size_t)nq * k * 8));
    const int rc = comm_exchange_merge_all(comms, ndev, nq, k, status.data(), dd.as<float>(), di.as<int64_t>());
    for (int d = 0; d < ndev; ++d) {   // the temporaries die with this frame: drain every device first
        (void)hipSetDevice(devices[d]);
        (void)hipDeviceSynchronize();
    }
    CVTMI_HIP(hipSetDevice(devices[0]));
    if (rc != CVTMI_OK) return rc;
    CVTMI_HIP(hipMemcpy(dist, dd.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost));
    CVTMI_HIP(hipMemcpy(ids, di.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost));
    return CVTMI_OK;
}

extern "C" {

// ================================================================ library =====================
int cvtmi_version(void) { return CVTMI_VERSION; }
const char *cvtmi_last_error(void) { return g_err.c_str(); }

int cvtmi_device_count(int *count)
{
    if (!count) return fail(CVTMI_EINVAL, "cvtmi_device_count: null");
    *count = 0;
    CVTMI_HIP(hipGetDeviceCount(count));
    return CVTMI_OK;
}

int cvtmi_set_tuning(const char *name, int64_t value)
{
    if (!name) return fail(CVTMI_EINVAL, "cvtmi_set_tuning: null name");
    if (!strcmp(name, "assign_variant")) {
        if (value < 0 || value > 2) return fail(CVTMI_EINVAL, "cvtmi_set_tuning: assign_variant must be 0, 1 or 2");
        set_assign_variant((int)value);
        return CVTMI_OK;
    }
    if (!strcmp(name, "flat_variant")) {
        if (value < 0 || value > 2) return fail(CVTMI_EINVAL, "cvtmi_set_tuning: flat_variant must be 0, 1 or 2");
        g_flat_variant = (int)value;
        return CVTMI_OK;
    }
    if (!strcmp(name, "probe_variant")) {
        if (value < 0 || value > 2) return fail(CVTMI_EINVAL, "cvtmi_set_tuning: probe_variant must be 0, 1 or 2");
        set_probe_variant((int)value);
        return CVTMI_OK;
    }
    if (!strcmp(name, "flat_f32_nt")) {
        if (value < 0 || value > 2) return fail(CVTMI_EINVAL, "cvtmi_set_tuning: flat_f32_nt must be 0, 1 or 2");
        set_flat_f32_nt((int)value);
        return CVTMI_OK;
    }
    if (!strcmp(name, "scanh_balance")) {
        if (value < 0 || value > 2) return fail(CVTMI_EINVAL, "cvtmi_set_tuning: scanh_balance must be 0, 1 or 2");
        set_scanh_balance((int)value);
        ++g_scanh_key;
        return CVTMI_OK;
    }
    if (!strcmp(name, "scans_dbg")) { set_scans_dbg((int)value); return CVTMI_OK; }
    if (!strcmp(name, "opq_host_chunk")) {
        if (value < 0 || value > (1 << 24)) return fail(CVTMI_EINVAL, "cvtmi_set_tuning: opq_host_chunk must be 0..2^24");
        g_host_chunks = (int)value;
        return CVTMI_OK;
    }
    if (!strcmp(name, "scanh_fix")) { set_scanh_fix(value); ++g_scanh_key; return CVTMI_OK; }
    if (!strcmp(name, "scanh_share_hist")) { set_scanh_share_hist((int)value); return CVTMI_OK; }
    if (!strcmp(name, "scanh_tail")) {
        set_scanh_tail((int)value);
        ++g_scanh_key;
        return CVTMI_OK;
    }
    if (!strcmp(name, "scanh_min_rows")) {
        if (value < 1) return fail(CVTMI_EINVAL, "cvtmi_set_tuning: scanh_min_rows must be positive");
        set_scanh_min_rows(value);
        ++g_scanh_key;
        return CVTMI_OK;
    }
    if (!strcmp(name, "scan_seed")) {
        if (value < 0 || value > 1) return fail(CVTMI_EINVAL, "cvtmi_set_tuning: scan_seed must be 0 or 1");
        set_scan_seed((int)value);
        return CVTMI_OK;
    }
    if (!strcmp(name, "flat_f32_stream")) {
        if (value < 0 || value > 2) return fail(CVTMI_EINVAL, "cvtmi_set_tuning: flat_f32_stream must be 0, 1 or 2");
        g_flat_f32_stream = (int)value;
        return CVTMI_OK;
    }
    if (!strcmp(name, "flat_f32_dbg")) { set_flat_f32_dbg((int)value); return CVTMI_OK; }
    if (!strcmp(name, "flat_f32_tfilter")) { set_flat_f32_tfilter((int)value); return CVTMI_OK; }
    if (!strcmp(name, "flat_f32_tfilter_min")) { set_flat_f32_tfilter_min((int)value); return CVTMI_OK; }
    if (!strcmp(name, "flat_f32_tfilter_one")) { set_flat_f32_tfilter_one((int)value); return CVTMI_OK; }
    if (!strcmp(name, "flat_f32_tfilter_bigk")) { set_flat_f32_tfilter_bigk((int)value); return CVTMI_OK; }
    if (!strcmp(name, "flat_f32_tfilter_retry")) { set_flat_f32_tfilter_retry((int)value); return CVTMI_OK; }
    if (!strcmp(name, "flat_f32_tfilter_wide_band")) { set_flat_f32_tfilter_wide_band((int)value); return CVTMI_OK; }
    if (!strcmp(name, "flat_f32_packed")) { set_flat_f32_packed((int)value); return CVTMI_OK; }
    if (!strcmp(name, "flat_f32_tfilter_min_rows")) { set_flat_f32_tfilter_min_rows((int)value); return CVTMI_OK; }
    if (!strcmp(name, "flat_f32_tfilter_sample")) { set_flat_f32_tfilter_sample((int)value); return CVTMI_OK; }
    if (!strcmp(name, "flat_f32_share")) {
        if (value < 0 || value > 3) return fail(CVTMI_EINVAL, "cvtmi_set_tuning: flat_f32_share must be 0 .. 3");
        set_flat_f32_share((int)value);
        return CVTMI_OK;
    }
    if (!strcmp(name, "opq_small_zero_copy")) { g_small_zero_copy = value != 0; return CVTMI_OK; }
    if (!strcmp(name, "host_spin_us")) { g_host_spin_us = value < 0 ? 0 : (int)value; return CVTMI_OK; }
    if (!strcmp(name, "hnsw_top_lds")) { set_hnsw_top_lds((int)value); return CVTMI_OK; }
    if (!strcmp(name, "hnsw_adc_tables")) { set_hnsw_adc_tables((int)value); return CVTMI_OK; }
    if (!strcmp(name, "hnsw_slots")) { g_hnsw_slots_cap = (int)value; return CVTMI_OK; }
    if (!strcmp(name, "flat_u8_tfilter")) { set_flat_u8_tfilter((int)value); return CVTMI_OK; }
    if (!strcmp(name, "flat_u8_tfilter_min_rows")) { set_flat_u8_tfilter_min_rows(value); return CVTMI_OK; }
    if (!strcmp(name, "flat_u8_tfilter_small_min_nq")) { set_flat_u8_tfilter_small_min_nq((int)std::min<int64_t>(value, 1 << 30)); return CVTMI_OK; }
    if (!strcmp(name, "flat_u8_tfilter_min_k")) { set_flat_u8_tfilter_min_k((int)std::min<int64_t>(value, 1 << 20)); return CVTMI_OK; }
    if (!strcmp(name, "flat_u8_tfilter_min_nq_k65")) { set_flat_u8_tfilter_min_nq_k65((int)std::min<int64_t>(value, 1 << 30)); return CVTMI_OK; }
    if (!strcmp(name, "flat_u8_tfilter_min_nq")) { set_flat_u8_tfilter_min_nq((int)std::min<int64_t>(value, 1 << 30)); return CVTMI_OK; }
    if (!strcmp(name, "flat_u8_tfilter_chunks")) { set_flat_u8_tfilter_chunks((int)std::min<int64_t>(value, 4)); return CVTMI_OK; }
    if (!strcmp(name, "flat_u8_tfilter_sample")) { set_flat_u8_tfilter_sample((int)std::min<int64_t>(value, 64)); return CVTMI_OK; }
    if (!strcmp(name, "flat_f32_rows_copy")) { g_flat_f32_rows_copy = value < 0 ? 0 : (value > (1 << 20) ? (1 << 20) : (int)value); return CVTMI_OK; }
    if (!strcmp(name, "flat_u8_gfilter")) { set_flat_u8_gfilter((int)value); return CVTMI_OK; }
    if (!strcmp(name, "sq8_encode_wave")) { set_sq8_encode_wave(value != 0); return CVTMI_OK; }
    if (!strcmp(name, "sq8_filter")) { set_sq8_filter(value != 0); return CVTMI_OK; }
    if (!strcmp(name, "scan_pad_m")) { g_scan_pad = value != 0; return CVTMI_OK; }
    if (!strcmp(name, "scan_packed_m")) { g_scan_packed = value != 0; return CVTMI_OK; }
    if (!strcmp(name, "scan_bigk")) { g_scan_bigk = value != 0; return CVTMI_OK; }
    if (!strcmp(name, "sq8_host_small")) { g_sq8_host_small = value != 0; return CVTMI_OK; }
    if (!strcmp(name, "scans_max_work")) { g_scans_max_work = value < 0 ? 0 : value; return CVTMI_OK; }
    if (!strcmp(name, "flat_u8_filter_min_nq")) { g_flat_u8_filter_min_nq = value < 1 ? 1 : value > (1 << 30) ? (1 << 30) : (int)value; return CVTMI_OK; }
    if (!strcmp(name, "flat_u8_filter_min_rows")) { g_flat_u8_filter_min_rows = value < 0 ? 0 : value; return CVTMI_OK; }
    if (!strcmp(name, "flat_u8_filter_min_work")) { g_flat_u8_filter_min_work = value < 0 ? 0 : value; return CVTMI_OK; }
    if (!strcmp(name, "flat_u8_sample_passes")) { g_flat_u8_sample_passes = value < 0 ? 0 : value > 64 ? 64 : (int)value; return CVTMI_OK; }
    if (!strcmp(name, "flat_small_zero_copy")) { g_flat_small_zero_copy = value != 0; return CVTMI_OK; }
    if (!strcmp(name, "opq_host_zero_copy")) { g_host_zero_copy = value != 0; return CVTMI_OK; }
    if (!strcmp(name, "scan_tail_splits")) { set_scan_tail_splits((int)value); return CVTMI_OK; }
    if (!strcmp(name, "sq8_flags")) { set_sq8_flags((int)value); return CVTMI_OK; }
    if (!strcmp(name, "sq8_wave_blocks")) {
        if (value < 1 || value > 64) return fail(CVTMI_EINVAL, "cvtmi_set_tuning: sq8_wave_blocks must be 1..64");
        set_sq8_wave_blocks((int)value);
        return CVTMI_OK;
    }
    if (!strcmp(name, "flat_u8_mstream_min_rows")) { set_flat_u8_mstream_min_rows(value); return CVTMI_OK; }
    if (!strcmp(name, "flat_u8_mstream_min")) {
        if (value < 1 || value > 129) return fail(CVTMI_EINVAL, "cvtmi_set_tuning: flat_u8_mstream_min must be 1..129");
        set_flat_u8_mstream_min((int)value);
        return CVTMI_OK;
    }
    if (!strcmp(name, "flat_u8_dbg")) {
        if (set_flat_u8_dbg((int)value) != CVTMI_OK) return fail(CVTMI_EUNSUPPORTED, "cvtmi_set_tuning: flat_u8_dbg needs a -DCVTMI_GF_DBG build (timing experiments, results wrong)");
        return CVTMI_OK;
    }
    if (!strcmp(name, "flat_u8_opt")) {
        if (value < 0 || value > 3) return fail(CVTMI_EINVAL, "cvtmi_set_tuning: flat_u8_opt must be 0..3");
        set_flat_u8_opt((int)value);
        return CVTMI_OK;
    }
    if (!strcmp(name, "comm_force_rccl")) { comm_set_force_rccl(value != 0); return CVTMI_OK; }
    if (!strcmp(name, "comm_check_status")) { comm_set_check_status(value != 0); return CVTMI_OK; }
    if (!strcmp(name, "comm_inject_failure")) { g_inject_failure = (int)value; return CVTMI_OK; }
    return fail(CVTMI_EINVAL, "cvtmi_set_tuning: unknown parameter '%s'", name);
}

int cvtmi_set_device(int device)
{
    CVTMI_HIP(hipSetDevice(device));
    return CVTMI_OK;
}

// ================================================================ OPQ =========================
int cvtmi_opq_create(int D, int coarseK, int M, int K, const float *coarse, const float *books, const float *R,
                     const int32_t *perm, cvtmi_opq_t *out)
{
    if (!out) return fail(CVTMI_EINVAL, "cvtmi_opq_create: null out");
    *out = nullptr;
    if (D < 1 || coarseK < 1 || M < 1 || K < 1 || !coarse || !books)
        return fail(CVTMI_EINVAL, "cvtmi_opq_create: bad model shape D=%d coarseK=%d M=%d K=%d", D, coarseK, M, K);
    if (D % M != 0) return fail(CVTMI_EINVAL, "cvtmi_opq_create: D=%d not divisible by M=%d", D, M);
    if (M > 16) return fail(CVTMI_EUNSUPPORTED, "cvtmi_opq_create: M=%d > 16 (IVFelem::PQindex[16])", M);
    if (K > 256) return fail(CVTMI_EUNSUPPORTED, "cvtmi_opq_create: K=%d > 256 (codes are uint8)", K);
    if (R && perm) return fail(CVTMI_EINVAL, "cvtmi_opq_create: give R or perm, not both");
    if (perm)
        for (int i = 0; i < D; ++i)
            if (perm[i] < 0 || perm[i] >= D) return fail(CVTMI_EINVAL, "cvtmi_opq_create: perm[%d]=%d out of range", i, perm[i]);
    int dev = 0;
    CVTMI_HIP(hipGetDevice(&dev));
    cvtmi_opq_s *h = new (std::nothrow) cvtmi_opq_s();
    if (!h) return fail(CVTMI_ENOMEM, "cvtmi_opq_create: out of host memory");
    h->device = dev;
    int rc = dev_alloc_copy(&h->d_coarse, coarse, (size_t)coarseK * D);
    if (rc == CVTMI_OK) rc = dev_alloc_copy(&h->d_books, books, (size_t)D * K);
    if (rc == CVTMI_OK && R) rc = dev_alloc_copy(&h->d_R, R, (size_t)D * D);
    if (rc == CVTMI_OK && perm) rc = dev_alloc_copy(&h->d_perm, perm, (size_t)D);
    if (rc != CVTMI_OK) { cvtmi_opq_destroy(h); return rc; }
    h->m.D = D; h->m.coarseK = coarseK; h->m.M = M; h->m.K = K; h->m.step = D / M;
    h->m.coarse = h->d_coarse; h->m.books = h->d_books; h->m.R = h->d_R; h->m.perm = h->d_perm;
    *out = h;
    return CVTMI_OK;
}

int cvtmi_opq_destroy(cvtmi_opq_t h)
{
    if (!h) return CVTMI_OK;
    (void)hipSetDevice(h->device);
    if (h->d_coarse) (void)hipFree(h->d_coarse);
    if (h->d_books) (void)hipFree(h->d_books);
    if (h->d_R) (void)hipFree(h->d_R);
    if (h->d_perm) (void)hipFree(h->d_perm);
    h->codes.release(); h->lists.release(); h->videos.release(); h->codes_rot.release(); h->codes16.release();
    h->csr_codes.release(); h->csr_videos.release(); h->csr_off.release(); h->csr_scratch.release(); h->csr_stats.release();
    h->s_qrot.release(); h->s_probe.release(); h->s_rot.release();
    for (OpqScratch *c : h->pool) { c->release_all(); delete c; }
    h->pool.clear();
    if (h->mutated) (void)hipEventDestroy(h->mutated);
    for (int e = 0; e < cvtmi_opq_s::kEvRing; ++e) {
        if (h->ev0[e]) (void)hipEventDestroy(h->ev0[e]);
        if (h->ev1[e]) (void)hipEventDestroy(h->ev1[e]);
    }
    h->sync.destroy();
    delete h;
    return CVTMI_OK;
}

static int opq_rotate_impl(cvtmi_opq_t h, const float *x, int64_t n, float *y, hipStream_t st);

int cvtmi_opq_rotate_dev(cvtmi_opq_t h, const float *x, int64_t n, float *y, void *stream)
{
    CHECK_H_SHARED(h);
    if (n < 0 || (n > 0 && (!x || !y))) return fail(CVTMI_EINVAL, "cvtmi_opq_rotate: bad arguments");
    return opq_rotate_impl(h, x, n, y, (hipStream_t)stream);
}

int cvtmi_opq_rotate(cvtmi_opq_t h, const float *x, int64_t n, float *y)
{
    CHECK_H_SHARED(h);
    if (n < 0 || (n > 0 && (!x || !y))) return fail(CVTMI_EINVAL, "cvtmi_opq_rotate: bad arguments");
    if (n == 0) return CVTMI_OK;
    const size_t bytes = (size_t)n * h->m.D * sizeof(float);
    Tmp dx, dy;
    CVTMI_TRY(dx.upload(x, bytes));
    CVTMI_TRY(dy.alloc(bytes));
    CVTMI_TRY(opq_rotate_impl(h, dx.as<float>(), n, dy.as<float>(), nullptr));
    CVTMI_HIP(hipMemcpy(y, dy.p, bytes, hipMemcpyDeviceToHost));
    return CVTMI_OK;
}

int cvtmi_opq_encode_dev(cvtmi_opq_t h, const float *x_rot, int64_t n, int32_t *list_id, uint8_t *codes, void *stream)
{
    CHECK_H_SERIAL(h, stream);
    if (n < 0 || (n > 0 && (!x_rot || !codes))) return fail(CVTMI_EINVAL, "cvtmi_opq_encode: bad arguments");
    if (n == 0) return CVTMI_OK;
    hipStream_t st = (hipStream_t)stream;
    int32_t *lists = list_id;
    if (h->m.coarseK > 1 && !lists) {
        CVTMI_TRY(h->s_probe.reserve((size_t)n * sizeof(int32_t)));
        lists = h->s_probe.as<int32_t>();
    }
    // coarseK == 1: every valid row lands in list 0; the residual is taken against centroid 0 either way, and the
    // matrix-core encode kernel writes the assignment itself
    if (lists && h->m.coarseK == 1 && pq_encode_fuses_lists(h->m, x_rot, n, h->p_encode))
        return launch_pq_encode(h->m, x_rot, n, nullptr, codes, st, h->p_encode, lists);
    if (lists) CVTMI_TRY(launch_coarse_assign(h->m, x_rot, n, lists, st));
    return launch_pq_encode(h->m, x_rot, n, h->m.coarseK > 1 ? lists : nullptr, codes, st, h->p_encode);
}

int cvtmi_opq_encode(cvtmi_opq_t h, const float *x_rot, int64_t n, int32_t *list_id, uint8_t *codes)
{
    CHECK_H_SERIAL(h, nullptr);
    if (n < 0 || (n > 0 && (!x_rot || !codes))) return fail(CVTMI_EINVAL, "cvtmi_opq_encode: bad arguments");
    if (n == 0) return CVTMI_OK;
    Tmp dx, dl, dc;
    CVTMI_TRY(dx.upload(x_rot, (size_t)n * h->m.D * sizeof(float)));
    CVTMI_TRY(dl.alloc((size_t)n * sizeof(int32_t)));
    CVTMI_TRY(dc.alloc((size_t)n * h->m.M));
    CVTMI_TRY(cvtmi_opq_encode_dev(h, dx.as<float>(), n, dl.as<int32_t>(), dc.as<uint8_t>(), nullptr));
    CVTMI_HIP(hipMemcpy(codes, dc.p, (size_t)n * h->m.M, hipMemcpyDeviceToHost));
    if (list_id) CVTMI_HIP(hipMemcpy(list_id, dl.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return CVTMI_OK;
}

// Rotation + encode of raw rows without a caller-side buffer of rotated rows: the rows go through a handle-owned scratch in
// chunks of 128 K rows (64 MB at D = 128) that is reused for every chunk, so the rotated rows live in the 256 MB Infinity
// Cache between the two kernels and their 2 x 4 D bytes per row need not reach HBM.  (The two kernels are bound by different
// pipes -- fp32 matrix cores vs VALU -- but the encode kernel owns the whole register file of its CU, so they cannot share a CU;
// a single fused kernel would have to give up the encode's software pipeline for the rotation's accumulators.)
int cvtmi_opq_rotate_encode_dev(cvtmi_opq_t h, const float *x, int64_t n, int32_t *list_id, uint8_t *codes, void *stream)
{
    CHECK_H_SERIAL(h, stream);
    if (n < 0 || (n > 0 && (!x || !codes))) return fail(CVTMI_EINVAL, "cvtmi_opq_rotate_encode: bad arguments");
    if (n == 0) return CVTMI_OK;
    if (!h->m.perm && !h->m.R) return cvtmi_opq_encode_dev(h, x, n, list_id, codes, stream);
    // the reference's own rotation is a permutation (reorder_, IVFOPQ.cpp:424-439): the exhaustive-model encode gathers through it,
    // no permuted copy of the rows is made
    if (h->m.perm && h->m.coarseK == 1 && pq_encode_takes_perm(h->m, x, n, h->p_encode))
        return launch_pq_encode(h->m, x, n, nullptr, codes, (hipStream_t)stream, h->p_encode, list_id, h->m.perm);
    const int64_t chunk = 131072;
    CVTMI_TRY(h->s_rot.reserve((size_t)std::min(n, chunk) * h->m.D * sizeof(float)));
    for (int64_t a = 0; a < n; a += chunk) {
        const int64_t m = std::min(chunk, n - a);
        CVTMI_TRY(opq_rotate_impl(h, x + a * h->m.D, m, h->s_rot.as<float>(), (hipStream_t)stream));   // (this call holds the handle exclusively)
        CVTMI_TRY(cvtmi_opq_encode_dev(h, h->s_rot.as<float>(), m, list_id ? list_id + a : nullptr, codes + a * h->m.M, stream));
    }
    return CVTMI_OK;
}

int cvtmi_opq_rotate_encode(cvtmi_opq_t h, const float *x, int64_t n, int32_t *list_id, uint8_t *codes)
{
    CHECK_H_SERIAL(h, nullptr);
    if (n < 0 || (n > 0 && (!x || !codes))) return fail(CVTMI_EINVAL, "cvtmi_opq_rotate_encode: bad arguments");
    if (n == 0) return CVTMI_OK;
    Tmp dx, dl, dc;
    CVTMI_TRY(dx.upload(x, (size_t)n * h->m.D * sizeof(float)));
    CVTMI_TRY(dl.alloc((size_t)n * sizeof(int32_t)));
    CVTMI_TRY(dc.alloc((size_t)n * h->m.M));
    CVTMI_TRY(cvtmi_opq_rotate_encode_dev(h, dx.as<float>(), n, dl.as<int32_t>(), dc.as<uint8_t>(), nullptr));
    CVTMI_HIP(hipMemcpy(codes, dc.p, (size_t)n * h->m.M, hipMemcpyDeviceToHost));
    if (list_id) CVTMI_HIP(hipMemcpy(list_id, dl.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return CVTMI_OK;
}

static int opq_ensure_rows(cvtmi_opq_t h, int64_t total, hipStream_t st)
{
    const size_t M = (size_t)h->m.M;
    if ((size_t)total * M > h->codes.cap) {
        size_t want = std::max((size_t)total, (size_t)(h->codes.cap / M) * 2);
        want = std::max(want, (size_t)4096);
        CVTMI_TRY(h->codes.grow(want * M, (size_t)h->n * M, st));
    }
    if (h->has_lists && (size_t)total * 4 > h->lists.cap)
        CVTMI_TRY(h->lists.grow(std::max((size_t)total, h->lists.cap / 2) * 4, (size_t)h->n * 4, st));
    if (h->has_videos && (size_t)total * 4 > h->videos.cap)
        CVTMI_TRY(h->videos.grow(std::max((size_t)total, h->videos.cap / 2) * 4, (size_t)h->n * 4, st));
    return CVTMI_OK;
}

__global__ void iota_i32_kernel(int32_t *p, int64_t begin, int64_t end, int32_t value, int is_iota)
{
    for (int64_t i = begin + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < end; i += (int64_t)gridDim.x * kBlock)
        p[i] = is_iota ? (int32_t)i : value;
}

static int fill_i32(int32_t *p, int64_t begin, int64_t end, int32_t value, int is_iota, hipStream_t st)
{
    if (end <= begin) return CVTMI_OK;
    int64_t blocks = (end - begin + kBlock - 1) / kBlock;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(iota_i32_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, p, begin, end, value, is_iota);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

static int opq_add_common(cvtmi_opq_t h, const uint8_t *codes, const int32_t *list_id, const int32_t *video_id, int64_t n,
                          hipMemcpyKind kind, hipStream_t st)
{
    if (n < 0 || (n > 0 && !codes)) return fail(CVTMI_EINVAL, "cvtmi_opq_add_codes: bad arguments");
    if (n == 0) return CVTMI_OK;
    if (!list_id && h->m.coarseK > 1) return fail(CVTMI_EINVAL, "cvtmi_opq_add_codes: list_id required when coarseK > 1");
    const int64_t total = h->n + n;
    if (total > 0xfffffffeLL) return fail(CVTMI_EUNSUPPORTED, "cvtmi_opq_add_codes: more than 2^32-2 entries per handle");
    // first explicit list / video ids: materialise the implicit prefix
    if (list_id && h->m.coarseK > 1 && !h->has_lists) {
        h->has_lists = true;
        CVTMI_TRY(h->lists.grow((size_t)std::max<int64_t>(total, 4096) * 4, 0, st));
        CVTMI_TRY(fill_i32(h->lists.as<int32_t>(), 0, h->n, 0, 0, st));
    }
    if (video_id && !h->has_videos) {
        h->has_videos = true;
        CVTMI_TRY(h->videos.grow((size_t)std::max<int64_t>(total, 4096) * 4, 0, st));
        CVTMI_TRY(fill_i32(h->videos.as<int32_t>(), 0, h->n, 0, 1, st));
    }
    CVTMI_TRY(opq_ensure_rows(h, total, st));
    const size_t M = (size_t)h->m.M;
    CVTMI_HIP(hipMemcpyAsync(h->codes.as<uint8_t>() + (size_t)h->n * M, codes, (size_t)n * M, kind, st));
    if (h->has_lists) {
        if (list_id) CVTMI_HIP(hipMemcpyAsync(h->lists.as<int32_t>() + h->n, list_id, (size_t)n * 4, kind, st));
        else CVTMI_TRY(fill_i32(h->lists.as<int32_t>(), h->n, total, 0, 0, st));
    }
    if (h->has_videos) {
        if (video_id) CVTMI_HIP(hipMemcpyAsync(h->videos.as<int32_t>() + h->n, video_id, (size_t)n * 4, kind, st));
        else CVTMI_TRY(fill_i32(h->videos.as<int32_t>(), h->n, total, 0, 1, st));
    }
    if (kind == hipMemcpyHostToDevice) CVTMI_HIP(stream_wait(st));
    h->n = total;
    h->csr_valid = false;
    return CVTMI_OK;
}

int cvtmi_opq_add_codes(cvtmi_opq_t h, const uint8_t *codes, const int32_t *list_id, const int32_t *video_id, int64_t n)
{
    CHECK_H_SERIAL(h, nullptr);
    return opq_add_common(h, codes, list_id, video_id, n, hipMemcpyHostToDevice, nullptr);
}

int cvtmi_opq_add_codes_dev(cvtmi_opq_t h, const uint8_t *codes, const int32_t *list_id, const int32_t *video_id,
                            int64_t n, void *stream)
{
    CHECK_H_SERIAL(h, stream);
    return opq_add_common(h, codes, list_id, video_id, n, hipMemcpyDeviceToDevice, (hipStream_t)stream);
}

int cvtmi_opq_reserve(cvtmi_opq_t h, int64_t n_total)
{
    CHECK_H_SERIAL(h, nullptr);
    if (n_total < 0 || n_total > 0xfffffffeLL) return fail(CVTMI_EINVAL, "cvtmi_opq_reserve: bad size");
    return h->codes.grow((size_t)n_total * h->m.M, (size_t)h->n * h->m.M, nullptr);
}

int cvtmi_opq_ntotal(cvtmi_opq_t h, int64_t *n)
{
    if (!h || !n) return fail(CVTMI_EINVAL, "cvtmi_opq_ntotal: null");
    *n = h->n;
    return CVTMI_OK;
}

int cvtmi_opq_reset(cvtmi_opq_t h)
{
    CHECK_H_SERIAL(h, nullptr);
    h->n = 0; h->has_lists = false; h->has_videos = false; h->csr_valid = false; h->rot_n = 0; h->pad_n = 0;
    return CVTMI_OK;
}

int cvtmi_opq_set_id_base(cvtmi_opq_t h, int64_t base)
{
    if (!h) return fail(CVTMI_EINVAL, "cvtmi_opq_set_id_base: null");
    h->id_base = base;
    return CVTMI_OK;
}

// list-ordered copy of the entries (stable: insertion order inside a list, as m_ivfList holds them), built on the device
// by a counting sort (query_video.hip); only 16 bytes of statistics come back to the host
static int opq_build_csr(cvtmi_opq_t h, hipStream_t st)
{
    if (h->csr_valid) return CVTMI_OK;
    const int64_t n = h->n;
    const int M = h->m.M, L = h->m.coarseK;
    int nb = 1;
    CVTMI_TRY(h->csr_scratch.reserve(csr_scratch_bytes(n, L, &nb)));
    CVTMI_TRY(h->csr_codes.reserve(std::max<size_t>((size_t)n * M, 16)));
    CVTMI_TRY(h->csr_videos.reserve(std::max<size_t>((size_t)n * 4, 16)));
    CVTMI_TRY(h->csr_off.reserve(((size_t)L + 1) * 8));
    CVTMI_TRY(h->csr_stats.reserve(16));
    CVTMI_TRY(launch_csr_build(h->has_lists ? h->lists.as<int32_t>() : nullptr, h->has_videos ? h->videos.as<int32_t>() : nullptr,
                               h->codes.as<uint8_t>(), n, L, M, h->csr_scratch.p, h->csr_off.as<int64_t>(), h->csr_codes.as<uint8_t>(),
                               h->csr_videos.as<int32_t>(), h->csr_stats.p, st));
    struct { int64_t longest; int32_t vmin, vmax; } stats;
    int64_t kept = 0;
    CVTMI_HIP(hipMemcpyAsync(&stats, h->csr_stats.p, sizeof stats, hipMemcpyDeviceToHost, st));
    CVTMI_HIP(hipMemcpyAsync(&kept, h->csr_off.as<int64_t>() + L, sizeof kept, hipMemcpyDeviceToHost, st));
    CVTMI_HIP(stream_wait(st));
    h->csr_longest = stats.longest; h->csr_vmin = stats.vmin; h->csr_vmax = stats.vmax; h->csr_kept = kept;
    h->csr_valid = true;
    return CVTMI_OK;
}

int cvtmi_opq_get_entries(cvtmi_opq_t h, int64_t *list_off, int32_t *video_id, uint8_t *codes)
{
    CHECK_H_SERIAL(h, nullptr);
    CVTMI_TRY(opq_build_csr(h, nullptr));
    if (list_off) CVTMI_HIP(hipMemcpy(list_off, h->csr_off.p, ((size_t)h->m.coarseK + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (video_id && h->csr_kept) CVTMI_HIP(hipMemcpy(video_id, h->csr_videos.p, (size_t)h->csr_kept * 4, hipMemcpyDeviceToHost));
    if (codes && h->csr_kept) CVTMI_HIP(hipMemcpy(codes, h->csr_codes.p, (size_t)h->csr_kept * h->m.M, hipMemcpyDeviceToHost));
    return CVTMI_OK;
}

int cvtmi_opq_lut_dev(cvtmi_opq_t h, const float *q_rot, int64_t nq, const int32_t *list_id, float *lut, void *stream)
{
    CHECK_H_SHARED(h);
    if (nq < 0 || (nq > 0 && (!q_rot || !lut))) return fail(CVTMI_EINVAL, "cvtmi_opq_lut: bad arguments");
    return launch_lut(h->m, q_rot, nq, list_id, lut, (hipStream_t)stream);
}

int cvtmi_opq_lut(cvtmi_opq_t h, const float *q_rot, int64_t nq, const int32_t *list_id, float *lut)
{
    CHECK_H_SHARED(h);
    if (nq < 0 || (nq > 0 && (!q_rot || !lut))) return fail(CVTMI_EINVAL, "cvtmi_opq_lut: bad arguments");
    if (nq == 0) return CVTMI_OK;
    if (list_id)
        for (int64_t i = 0; i < nq; ++i)
            if (list_id[i] >= h->m.coarseK) return fail(CVTMI_EINVAL, "cvtmi_opq_lut: list_id[%lld] out of range", (long long)i);
    Tmp dq, dl, dt;
    CVTMI_TRY(dq.upload(q_rot, (size_t)nq * h->m.D * sizeof(float)));
    if (list_id) CVTMI_TRY(dl.upload(list_id, (size_t)nq * sizeof(int32_t)));
    const size_t lb = (size_t)nq * h->m.M * h->m.K * sizeof(float);
    CVTMI_TRY(dt.alloc(lb));
    CVTMI_TRY(launch_lut(h->m, dq.as<float>(), nq, list_id ? dl.as<int32_t>() : nullptr, dt.as<float>(), nullptr));
    CVTMI_HIP(hipMemcpy(lut, dt.p, lb, hipMemcpyDeviceToHost));
    return CVTMI_OK;
}

static int opq_rotate_impl(cvtmi_opq_t h, const float *x, int64_t n, float *y, hipStream_t st)
{
    if (h->m.perm) return launch_permute(h->m.perm, h->m.D, x, n, y, st);
    if (h->m.R) return launch_rotate_gemm(h->m.R, h->m.D, x, n, y, st);
    if (n > 0) CVTMI_HIP(hipMemcpyAsync(y, x, (size_t)n * h->m.D * sizeof(float), hipMemcpyDeviceToDevice, st));
    return CVTMI_OK;
}

// profile mode: shape of the last scan launch, one consistent triple however many searches run side by side
static void opq_note_scan(cvtmi_opq_t h, int64_t bytes, int qt, int splits)
{
    std::lock_guard<std::mutex> g(h->pool_mu);
    h->last_bytes = bytes; h->last_qt = qt; h->last_splits = splits;
}

// profile mode: a slot of the handle's event ring for one scan launch (searches may run side by side)
static int opq_profile_slot(cvtmi_opq_t h, int *slot)
{
    std::lock_guard<std::mutex> g(h->pool_mu);
    *slot = h->ev_count++ % cvtmi_opq_s::kEvRing;
    if (!h->ev0[*slot]) { CVTMI_HIP(hipEventCreate(&h->ev0[*slot])); CVTMI_HIP(hipEventCreate(&h->ev1[*slot])); }
    return CVTMI_OK;
}

// scan_variant 6 (adc_scan16h, adc_scan_h.hip): tables once per query group, a persistent grid over a host-built item table,
// candidates in per-workgroup spill areas, one selection per (segment, query), merge of the groups' partial lists
static int opq_search_h(cvtmi_opq_t h, OpqScratch &S, const float *q_rot, int64_t nq, int k, float *dist, int64_t *ids, const uint8_t *codes_rot,
                        hipStream_t st)
{
    if (S.hplan_n != h->n || S.hplan_nq != nq || S.hplan_splits != h->p_splits || S.hplan_key != g_scanh_key) {
        scanh_plan(h->n, nq, h->p_splits, S.hplan);
        const size_t bytes = S.hplan.items.size() * sizeof(ScanItem), mbytes = S.hplan.multi.size() * sizeof(uint32_t);
        CVTMI_TRY(S.s_items.reserve(std::max<size_t>(bytes + mbytes, 16)));
        // (pageable sources: the runtime stages them before the call returns, so the vectors may change afterwards)
        if (bytes) CVTMI_HIP(hipMemcpyAsync(S.s_items.p, S.hplan.items.data(), bytes, hipMemcpyHostToDevice, st));
        if (mbytes) CVTMI_HIP(hipMemcpyAsync(S.s_items.as<char>() + bytes, S.hplan.multi.data(), mbytes, hipMemcpyHostToDevice, st));
        S.hplan_n = h->n; S.hplan_nq = nq; S.hplan_splits = h->p_splits; S.hplan_key = g_scanh_key;
    }
    const ScanHPlan &hp = S.hplan;
    float *pd = dist;
    int64_t *pi = ids;
    if (hp.stride > 1) {
        const size_t cnt = (size_t)nq * hp.stride * k;
        CVTMI_TRY(S.s_part_d.reserve(cnt * sizeof(float)));
        CVTMI_TRY(S.s_part_id.reserve(cnt * sizeof(int64_t)));
        pd = S.s_part_d.as<float>();
        pi = S.s_part_id.as<int64_t>();
    }
    CVTMI_TRY(S.s_lut.reserve((size_t)nq * 16 * 256 * sizeof(float)));
    CVTMI_TRY(S.s_qlut.reserve(scanh_qlut_bytes(nq)));
    CVTMI_TRY(S.s_qp.reserve(scanh_qp_bytes(nq)));
    CVTMI_TRY(S.s_spill.reserve(scanh_spill_bytes(hp.grid)));
    uint32_t *gthr = nullptr;
    if (hp.stride > 1 && h->p_share) {
        CVTMI_TRY(S.s_gthr.reserve(scanh_gthr_bytes(nq)));   // the bounds and, behind them, the histograms the segments of a query share
        gthr = S.s_gthr.as<uint32_t>();
    }
    int slot = 0;
    if (h->p_profile) {
        CVTMI_TRY(opq_profile_slot(h, &slot));
        CVTMI_HIP(hipEventRecord(h->ev0[slot], st));
    }
    CVTMI_TRY(launch_adc_scan_h(h->m, h->codes.as<uint8_t>(), codes_rot, h->n, h->id_base, q_rot, nq, k, hp, S.s_items.as<ScanItem>(), pd, pi, dist, ids,
                                S.s_lut.as<float>(), S.s_qlut.p, S.s_qp.p, S.s_spill.p, gthr, h->p_lazy, scan_seed_enabled(), st));
    if (h->p_profile) {
        CVTMI_HIP(hipEventRecord(h->ev1[slot], st));
        opq_note_scan(h, ((nq + 7) / 8) * h->n * h->m.M, 8, hp.stride);  // passes x rows x M code bytes
    }
    if (hp.stride > 1)  // (queries of groups scanned in one piece are already in place: the merge skips them)
        CVTMI_TRY(launch_topk_merge(pd, pi, nq, hp.stride, k, dist, ids, st,
                                    reinterpret_cast<const uint32_t *>(S.s_items.as<char>() + hp.items.size() * sizeof(ScanItem))));
    return CVTMI_OK;
}

// M < 16 (the reference's own test model has M = 8: opq/src/multi_frame_index_test.cpp) only had the round-1 row-per-lane kernels: 1.0 M
// queries/s at C2's shape against 3.1 M for M = 16, which does twice the work (round 5, tools/opq_m_sweep.py).  The M = 16 kernels take
// such an index as it is once the rows are padded to 16 code bytes with zeros and every query gets 16 - M all-zero tables behind its own:
// a zero byte looks up a zero, integer bounds and fp32 sums are unchanged (x + 0.0f == x for the non-negative sums here), the exact
// re-sum still adds the model's M entries in the reference's order first.  Costs 16 + 16 bytes per row of derived copies.
static bool opq_pads(const cvtmi_opq_s *h)
{
    return g_scan_pad.load() && h->m.M >= 1 && h->m.M < 16 && h->m.K >= 1 && h->m.K <= 256 && h->m.D <= 256;
}
static ScanPlan opq_plan(cvtmi_opq_t h, int64_t nq, int k)
{
    if (opq_pads(h) && (h->p_variant == 7 || h->p_variant >= 3) && h->p_qtile == 0) {
        OpqModelDev m16 = h->m;
        m16.M = 16;
        // (the persistent grid and the small-batch form build their tables from the codebooks themselves: adc_scan16q / 16a only)
        // (planned as at least four queries: below that plan_scan prefers the fp32-table kernels, which build their tables from the codebooks)
        // M = 8 / 4 natively (adc_scan16p: 16 / M rows per 16-byte load; needs the pre-rotated copy): planned in units of loads -- a load
        // costs what an M = 16 row costs, so the split decision sees n / RPL "rows"
        const bool packed = g_scan_packed.load() && (h->m.M == 8 || h->m.M == 4) && h->p_prerot && h->p_variant != 4 && h->p_variant != 5;
        const int64_t n_plan = packed ? (h->n * h->m.M + 15) / 16 : h->n;
        ScanPlan pp = plan_scan(m16, n_plan, std::max<int64_t>(nq, 4), k, 0, h->p_splits, h->p_variant == 4 || h->p_variant == 5 ? h->p_variant : 3);
        if (pp.variant >= 3 && pp.variant <= 5) {
            pp.real_M = h->m.M;
            pp.packed = packed && pp.variant == 3;
            if (!h->p_tail) { pp.groups_a = 0; pp.splits_b = 0; }
            return pp;
        }
    }
    ScanPlan plan = plan_scan(h->m, h->n, nq, k, h->p_qtile, h->p_splits, h->p_variant);
    if (plan.variant == 6) return plan;
    if (!h->p_tail) { plan.groups_a = 0; plan.splits_b = 0; }
    if (h->p_groups_a > 0 && h->p_splits_b > plan.splits && plan.variant >= 3 &&
        h->p_groups_a < (nq + plan.qtile - 1) / plan.qtile) {
        plan.groups_a = h->p_groups_a; plan.splits_b = h->p_splits_b;
    }
    return plan;
}

// what a search needs of the index beyond the rows: the pre-rotated copy the M = 16 scans stream, extended under the EXCLUSIVE lock
// (once per index state; the searches that follow on other streams wait for the event the exclusive call leaves)
static int opq_prepare(cvtmi_opq_t h, int64_t nq, int k, hipStream_t st)
{
    bool padded = false, packed = false;
    {
        std::shared_lock<std::shared_timed_mutex> rd(h->rw);
        if (h->n == 0) return CVTMI_OK;
        const ScanPlan plan = opq_plan(h, nq, k);
        packed = plan.packed;
        padded = plan.real_M > 0 && !packed;
        const bool want_rot = h->p_prerot && ((h->m.M == 16 && (plan.variant >= 3 || scans_applies(h->m, h->n, nq, k) ||
                                                                (g_scan_bigk.load() && scank_applies(h->m, h->n, nq, k)))) || padded || packed);
        if (!padded && !want_rot) return CVTMI_OK;
        const bool pad_ok = !padded || (h->pad_n == h->n && h->codes16.cap >= (size_t)h->n * 16);
        const bool rot_ok = !want_rot || (h->rot_n == h->n && h->rot_kind == (packed ? 1 : 0) &&
                                          h->codes_rot.cap >= (packed ? ((size_t)h->n * h->m.M + 15) / 16 * 16 : (size_t)h->n * 16));
        if (pad_ok && rot_ok) return CVTMI_OK;
    }
    Serial serial(h->sync, st);
    OpqExclusive excl(h, st);
    if (packed) {   // the rows themselves, rotated inside their 16-byte groups: M bytes per row, no padded copy
        if (h->rot_kind != 1) { h->rot_kind = 1; h->rot_n = 0; }
        if (h->rot_n > h->n) h->rot_n = 0;
        const size_t need = ((size_t)h->n * h->m.M + 15) / 16 * 16;
        if (h->codes_rot.cap < need) {
            int rc = h->codes_rot.reserve(std::max<size_t>((h->codes.cap + 15) / 16 * 16, need));
            if (rc == CVTMI_ENOMEM) rc = h->codes_rot.reserve(need);
            h->rot_n = 0;
            if (rc == CVTMI_ENOMEM) { (void)hipGetLastError(); g_err.clear(); return CVTMI_OK; }   // (the search falls back to the padded / row-per-lane forms)
            CVTMI_TRY(rc);
        }
        CVTMI_TRY(launch_rotate_codes_packed(h->codes.as<uint8_t>(), h->m.M, h->codes_rot.as<uint8_t>(), h->rot_n, h->n, st));
        h->rot_n = h->n;
        return CVTMI_OK;
    }
    if (h->rot_kind != 0) { h->rot_kind = 0; h->rot_n = 0; }
    if (padded) {   // the 16-byte rows first: the rotated copy is made from them
        if (h->pad_n > h->n) h->pad_n = 0;
        if (h->codes16.cap < (size_t)h->n * 16) {
            // The derived copies are an optimisation (16 + 16 bytes per row beside M): when HBM does not hold them the search must still
            // answer -- the row-per-lane kernels take the index as it is (opq_search_leased checks pad_n == n).  ADVICE r5.
            int rc = h->codes16.reserve(std::max<size_t>(h->codes.cap / (size_t)h->m.M * 16, (size_t)h->n * 16));
            if (rc == CVTMI_ENOMEM) rc = h->codes16.reserve((size_t)h->n * 16);   // without the growth margin of the code buffer
            h->pad_n = 0;  // reserve() does not keep the old contents
            if (rc == CVTMI_ENOMEM) { (void)hipGetLastError(); g_err.clear(); return CVTMI_OK; }
            CVTMI_TRY(rc);
        }
        CVTMI_TRY(launch_pad_codes(h->codes.as<uint8_t>(), h->m.M, h->codes16.as<uint8_t>(), h->pad_n, h->n, st));
        h->pad_n = h->n;
        if (!h->p_prerot) return CVTMI_OK;
    }
    if (h->rot_n > h->n) h->rot_n = 0;
    if (h->codes_rot.cap < (size_t)h->n * 16) {
        int rc = h->codes_rot.reserve(std::max<size_t>(padded ? h->codes16.cap : h->codes.cap, (size_t)h->n * 16));
        if (rc == CVTMI_ENOMEM) rc = h->codes_rot.reserve((size_t)h->n * 16);
        h->rot_n = 0;  // reserve() does not keep the old contents
        if (rc == CVTMI_ENOMEM) { (void)hipGetLastError(); g_err.clear(); return CVTMI_OK; }   // (the scans rotate in registers without it)
        CVTMI_TRY(rc);
    }
    CVTMI_TRY(launch_rotate_codes(padded ? h->codes16.as<uint8_t>() : h->codes.as<uint8_t>(), h->codes_rot.as<uint8_t>(), h->rot_n, h->n, st));
    h->rot_n = h->n;
    return CVTMI_OK;
}

// Does the dispatch take the small-batch form (adc_scan_h.hip: scans, up to 128 queries)?  Its per-group passes over the whole table
// were fitted at 1 M rows; round 5 swept the table size (tools/sweep_scan_dispatch.py, profiles/r05_scan_dispatch_sweep.txt): it is
// ahead of the persistent grid while rows x query groups stays under ~48 M (2 M rows: up to 128 queries; 10 M: up to 32; 30 M: 8)
// and behind by up to 2x beyond (100 M rows, 128 queries: 8.3 against 3.7 ms).
static bool scans_chosen(const cvtmi_opq_s *h, int64_t nq, int k)
{
    return h->p_variant == 7 && h->p_splits == 0 && h->p_qtile == 0 && h->p_small && scans_applies(h->m, h->n, nq, k) &&
           h->n * ((nq + 7) / 8) <= g_scans_max_work.load();
}

// the dispatch a search would take, for inspection and for the CPU tests that pin the rules (include/cvtmi.h)
extern "C" int cvtmi_opq_describe_dispatch(int D, int M, int K, int64_t n_rows, int64_t nq, int k, int out[7])
{
    if (!out || D < 1 || M < 1 || M > 16 || D % M != 0 || K < 1 || K > 256 || n_rows < 0 || nq < 1 || k < 1)
        return fail(CVTMI_EINVAL, "cvtmi_opq_describe_dispatch: bad arguments");
    cvtmi_opq_s h;   // default settings; nothing of it touches a device
    h.m.D = D; h.m.M = M; h.m.K = K; h.m.step = D / M; h.m.coarseK = 1;
    h.n = n_rows;
    const bool small = scans_chosen(&h, nq, k);
    const ScanPlan p = opq_plan(&h, nq, k);
    out[0] = small ? 1 : 0; out[1] = p.variant; out[2] = p.qtile; out[3] = p.splits; out[4] = p.groups_a; out[5] = p.splits_b; out[6] = p.real_M;
    return CVTMI_OK;
}

// one search on stream st with the scratch set S; the caller holds h->rw shared
static int opq_search_leased(cvtmi_opq_t h, OpqScratch &S, const float *q, int64_t nq, int rotate, int k, float *dist, int64_t *ids, hipStream_t st)
{
    if (h->n == 0)  // an empty index (e.g. a rank whose row block is empty): all padding, (+inf, -1)
        return launch_topk_select(nullptr, nullptr, nq, 0, k, dist, ids, st);
    if (scans_chosen(h, nq, k)) {
        // 1 .. 128 queries (up to sixteen query groups): global bounds first, candidate lists, one selection workgroup per query
        // (adc_scan_h.hip) -- four launches, the rotation folded into the first
        const uint8_t *crot = (h->m.M == 16 && h->p_prerot && h->rot_n == h->n && h->codes_rot.p) ? h->codes_rot.as<uint8_t>() : nullptr;
        CVTMI_TRY(S.s_lut.reserve((size_t)((nq + 7) / 8 * 8) * 16 * 256 * sizeof(float)));
        CVTMI_TRY(S.s_qlut.reserve(scanh_qlut_bytes(nq)));
        CVTMI_TRY(S.s_qp.reserve(scanh_qp_bytes(nq)));
        CVTMI_TRY(S.s_spill.reserve(scans_scratch_bytes()));
        int slot = 0;
        if (h->p_profile) {
            CVTMI_TRY(opq_profile_slot(h, &slot));
            CVTMI_HIP(hipEventRecord(h->ev0[slot], st));
        }
        const float *qs = q;
        int rot = rotate && (h->m.perm || h->m.R);
        if (rot && !scans_fuses_rotation(h->m)) {   // a dense rotation wider than 128: the rotation kernel first
            CVTMI_TRY(S.s_qrot.reserve((size_t)nq * h->m.D * sizeof(float)));
            CVTMI_TRY(opq_rotate_impl(h, q, nq, S.s_qrot.as<float>(), st));
            qs = S.s_qrot.as<float>(); rot = 0;
        }
        CVTMI_TRY(launch_adc_scan_small(h->m, h->codes.as<uint8_t>(), crot, h->n, h->id_base, qs, rot, nq, k, dist, ids, S.s_lut.as<float>(), S.s_qlut.p,
                                        S.s_qp.p, S.s_spill.p, h->p_lazy, st));
        if (h->p_profile) {
            CVTMI_HIP(hipEventRecord(h->ev1[slot], st));
            opq_note_scan(h, h->n * h->m.M, 8, 1);
        }
        return CVTMI_OK;
    }
    const float *q_rot = q;
    if (rotate && (h->m.perm || h->m.R)) {
        CVTMI_TRY(S.s_qrot.reserve((size_t)nq * h->m.D * sizeof(float)));
        CVTMI_TRY(opq_rotate_impl(h, q, nq, S.s_qrot.as<float>(), st));
        q_rot = S.s_qrot.as<float>();
    }
    if (g_scan_bigk.load() && h->p_variant == 7 && h->p_qtile == 0 && h->p_splits == 0 && scank_applies(h->m, h->n, nq, k)) {
        // k = 129 .. 2048 (round 6): sampled histogram bound, candidate lists, one selection workgroup per query (adc_scan_h.hip); the
        // queries it could not answer (a list that overflowed, a crowded band, tables that bound nothing) are flagged and go through the
        // exact kernel behind it.  Batches whose candidate lists would pass 1 GB go in pieces.
        const uint8_t *crot = (h->p_prerot && h->rot_kind == 0 && h->rot_n == h->n && h->codes_rot.p) ? h->codes_rot.as<uint8_t>() : nullptr;
        int64_t per = nq;
        while (per > 8 && scank_scratch_bytes(h->n, per, k) > ((size_t)1 << 30)) per = ((per / 2) + 7) / 8 * 8;
        CVTMI_TRY(S.s_lut.reserve((size_t)((per + 7) / 8 * 8) * 16 * 256 * sizeof(float)));
        CVTMI_TRY(S.s_qlut.reserve(scanh_qlut_bytes(per)));
        CVTMI_TRY(S.s_qp.reserve(scanh_qp_bytes(per)));
        CVTMI_TRY(S.s_spill.reserve(scank_scratch_bytes(h->n, per, k)));
        int slot = 0;
        if (h->p_profile) {
            CVTMI_TRY(opq_profile_slot(h, &slot));
            CVTMI_HIP(hipEventRecord(h->ev0[slot], st));
        }
        ScanPlan exact;
        exact.qtile = 1; exact.splits = 1; exact.variant = 0;
        for (int64_t q0 = 0; q0 < nq; q0 += per) {
            const int64_t n1 = std::min(per, nq - q0);
            uint32_t *flags = nullptr;
            CVTMI_TRY(launch_adc_scan_bigk(h->m, h->codes.as<uint8_t>(), crot, h->n, h->id_base, q_rot + q0 * h->m.D, n1, k, dist + q0 * k, ids + q0 * k,
                                           S.s_lut.as<float>(), S.s_qlut.p, S.s_qp.p, S.s_spill.p, h->p_lazy, &flags, st));
            CVTMI_TRY(launch_adc_scan(h->m, h->codes.as<uint8_t>(), h->n, h->id_base, q_rot + q0 * h->m.D, n1, k, exact, dist + q0 * k, ids + q0 * k,
                                      nullptr, nullptr, st, nullptr, h->p_lazy, nullptr, nullptr, flags));
        }
        if (h->p_profile) {
            CVTMI_HIP(hipEventRecord(h->ev1[slot], st));
            opq_note_scan(h, ((nq + 7) / 8) * h->n * h->m.M * 5 / 4, 8, 1);   // passes x rows x M code bytes, the sampled pass included
        }
        return CVTMI_OK;
    }
    ScanPlan plan = opq_plan(h, nq, k);
    if (plan.packed && !(h->rot_kind == 1 && h->rot_n == h->n && h->codes_rot.p))   // (the packed rotation is not there: as before)
        plan = plan_scan(h->m, h->n, nq, k, h->p_qtile, h->p_splits, h->p_variant);
    if (plan.real_M > 0 && !plan.packed && !(h->pad_n == h->n && h->codes16.p)) plan = plan_scan(h->m, h->n, nq, k, h->p_qtile, h->p_splits, h->p_variant);   // (the padded rows are not there: as before)
    const bool packed = plan.packed, padded = plan.real_M > 0 && !packed;
    OpqModelDev m_scan = h->m;
    if (padded || packed) m_scan.M = 16;
    const uint8_t *scan_rows = padded ? h->codes16.as<uint8_t>() : h->codes.as<uint8_t>();
    // the scan streams the pre-rotated copy of the rows when it is up to date (opq_prepare); otherwise it rotates in registers
    const uint8_t *codes_rot = packed ? h->codes_rot.as<uint8_t>()
                             : (plan.variant >= 3 && (h->m.M == 16 || padded) && h->p_prerot && h->rot_kind == 0 && h->rot_n == h->n && h->codes_rot.p) ? h->codes_rot.as<uint8_t>() : nullptr;
    if (plan.variant == 6) return opq_search_h(h, S, q_rot, nq, k, dist, ids, codes_rot, st);
    float *pd = dist;
    int64_t *pi = ids;
    if (plan.stride() > 1) {
        const size_t cnt = (size_t)nq * plan.stride() * k;
        CVTMI_TRY(S.s_part_d.reserve(cnt * sizeof(float)));
        CVTMI_TRY(S.s_part_id.reserve(cnt * sizeof(int64_t)));
        pd = S.s_part_d.as<float>();
        pi = S.s_part_id.as<int64_t>();
    }
    int slot = 0;
    if (h->p_profile) {
        CVTMI_TRY(opq_profile_slot(h, &slot));
        CVTMI_HIP(hipEventRecord(h->ev0[slot], st));
    }
    float *lut_scratch = nullptr;
    if (plan.variant >= 3) {  // per-query fp32 tables in HBM (16 KB per query at M=16, K=256)
        CVTMI_TRY(S.s_lut.reserve((size_t)nq * m_scan.M * 256 * sizeof(float)));
        lut_scratch = S.s_lut.as<float>();
    }
    uint32_t *gthr = nullptr;
    if (plan.variant >= 3 && plan.stride() > 1 && h->p_share) {
        CVTMI_TRY(S.s_gthr.reserve((size_t)nq * sizeof(uint32_t)));
        gthr = S.s_gthr.as<uint32_t>();
    }
    // (two-region plan, first region in one piece: those queries' lists are written in place by the scan, the merge starts behind them)
    const int64_t placed = plan.stride() > 1 ? scan_in_place_queries(plan, m_scan.M, nq) : 0;
    CVTMI_TRY(launch_adc_scan(m_scan, scan_rows, h->n, h->id_base, q_rot, nq, k, plan, pd, pi, lut_scratch,
                              codes_rot, st, gthr, h->p_lazy, placed ? dist : nullptr, placed ? ids : nullptr));
    if (h->p_profile) {
        CVTMI_HIP(hipEventRecord(h->ev1[slot], st));
        const int64_t groups = (nq + plan.qtile - 1) / plan.qtile;
        opq_note_scan(h, groups * h->n * h->m.M, plan.qtile, plan.splits);  // passes x rows x M code bytes
    }
    if (plan.stride() > 1)
        CVTMI_TRY(launch_topk_merge(pd + placed * plan.stride() * k, pi + placed * plan.stride() * k, nq - placed, plan.stride(), k, dist + placed * k,
                                    ids + placed * k, st));
    return CVTMI_OK;
}

int cvtmi_opq_search_dev(cvtmi_opq_t h, const float *q, int64_t nq, int rotate, int k, float *dist, int64_t *ids,
                         void *stream)
{
    CHECK_H(h);
    if (nq < 0 || (nq > 0 && (!q || !dist || !ids))) return fail(CVTMI_EINVAL, "cvtmi_opq_search: bad arguments");
    if (h->m.coarseK != 1)
        return fail(CVTMI_EUNSUPPORTED, "cvtmi_opq_search: exhaustive search needs coarseK == 1 (use cvtmi_opq_query_video)");
    if (k < 1 || k > CVTMI_K_MAX) return fail(CVTMI_EUNSUPPORTED, "cvtmi_opq_search: k=%d outside 1..%d", k, CVTMI_K_MAX);
    if (nq == 0) return CVTMI_OK;
    hipStream_t st = (hipStream_t)stream;
    CVTMI_TRY(opq_prepare(h, nq, k, st));
    std::shared_lock<std::shared_timed_mutex> rd(h->rw);
    OpqLease lease;
    CVTMI_TRY(lease.open(h, st, false));
    return opq_search_leased(h, *lease.s, q, nq, rotate, k, dist, ids, st);
}


// Host-pointer search, the reference's own call shape (opq/src/multi_frame_index_test.cpp:45-54 hands over host buffers).  A large
// batch is cut into chunks that alternate between TWO scratch sets with their own streams: while chunk i is scanned, the queries of
// chunk i + 1 go up and the results of chunk i - 1 come down (pinned staging areas the sets keep), and the scan kernels of
// neighbouring chunks fill each other's last, partly occupied round of workgroups.
int cvtmi_opq_search(cvtmi_opq_t h, const float *q, int64_t nq, int rotate, int k, float *dist, int64_t *ids)
{
    CHECK_H(h);
    if (nq < 0 || (nq > 0 && (!q || !dist || !ids))) return fail(CVTMI_EINVAL, "cvtmi_opq_search: bad arguments");
    if (nq == 0) return CVTMI_OK;
    if (h->m.coarseK != 1)
        return fail(CVTMI_EUNSUPPORTED, "cvtmi_opq_search: exhaustive search needs coarseK == 1 (use cvtmi_opq_query_video)");
    if (k < 1 || k > CVTMI_K_MAX) return fail(CVTMI_EUNSUPPORTED, "cvtmi_opq_search: k=%d outside 1..%d", k, CVTMI_K_MAX);
    const int D = h->m.D;
    // pieces of `per` queries: 4096 by default = 512 query groups = ONE full round of the scan's workgroups on 256 CUs, so cutting
    // the batch there costs the scan nothing (10 000 queries: 1 + 1 + 0.44 rounds either way)
    // Page-locked result arrays (cvtmi_host_alloc): the kernels write the lists straight into them (device-visible host memory, posted
    // PCIe writes: 12 MB over the 3 ms of a 10 000-query scan) -- no device copy of the results, no copy engine, and therefore no
    // reason to cut the batch: ONE launch chain, the same the device-pointer entry issues; the query groups that are scanned in one
    // piece deliver their lists as their workgroups end.  Round 5: 3.45-3.55 -> 3.15-3.25 ms per 10 000 queries (device pointers: 3.0-3.1).
    const bool q_pinned = host_pinned(q), out_pinned = host_pinned(dist) && host_pinned(ids);
    float *zd = nullptr;
    int64_t *zi = nullptr;
    if (out_pinned && g_host_zero_copy.load()) {
        void *pd = nullptr, *pi = nullptr;
        if (hipHostGetDevicePointer(&pd, dist, 0) == hipSuccess && hipHostGetDevicePointer(&pi, ids, 0) == hipSuccess && pd && pi) {
            zd = static_cast<float *>(pd); zi = static_cast<int64_t *>(pi);
        } else {
            (void)hipGetLastError();
        }
    }
    int64_t per = g_host_chunks.load();
    const int64_t per_default = per > 0 ? per : 4096;
    if (per <= 0 || nq < per + per / 4 || zd) per = nq;
    // (zero-copy results keep the batch whole only while its QUERIES fit the staging guard below: past that -- more than 131 072
    //  queries at D = 128 -- the batch goes out in the default pieces, still straight into the caller's arrays, rather than through a
    //  freshly allocated device copy of everything: ADVICE r5)
    if (zd && (per > (64 << 20) / (int64_t)(D * 4) || (size_t)per * k * 12 > ((size_t)256 << 20))) per = per_default;
    per = (per + 7) / 8 * 8;   // whole query groups
    const int chunks = (int)((nq + per - 1) / per);
    if (per > (64 << 20) / (int64_t)(D * 4) || (size_t)per * k * 12 > ((size_t)256 << 20)) {
        // very large pieces: no staging area of that size is kept around
        Tmp dq, dd, di;
        const size_t qb = (size_t)nq * D * sizeof(float), db = (size_t)nq * k * sizeof(float), ib = (size_t)nq * k * sizeof(int64_t);
        CVTMI_TRY(dq.upload(q, qb));
        CVTMI_TRY(dd.alloc(db));
        CVTMI_TRY(di.alloc(ib));
        CVTMI_TRY(cvtmi_opq_search_dev(h, dq.as<float>(), nq, rotate, k, dd.as<float>(), di.as<int64_t>(), nullptr));
        CVTMI_HIP(hipMemcpy(dist, dd.p, db, hipMemcpyDeviceToHost));
        CVTMI_HIP(hipMemcpy(ids, di.p, ib, hipMemcpyDeviceToHost));
        return CVTMI_OK;
    }
    CVTMI_TRY(opq_prepare(h, per, k, nullptr));
    std::shared_lock<std::shared_timed_mutex> rd(h->rw);
    OpqLease lease[2];
    const int nsets = chunks > 1 ? 2 : 1;
    for (int i = 0; i < nsets; ++i) CVTMI_TRY(lease[i].open(h, nullptr, true));
    const size_t qb = (size_t)per * D * sizeof(float), db = (size_t)per * k * sizeof(float), ib = (size_t)per * k * sizeof(int64_t);
    for (int i = 0; i < nsets; ++i) {
        OpqScratch &S = *lease[i].s;
        CVTMI_TRY(S.io_q.reserve(qb));
        if (zd && !scans_applies(h->m, h->n, nq, k)) {   // (what may take the small-batch form below stages everything)
            if (!q_pinned) CVTMI_TRY(S.io_pin.reserve(qb));
            continue;
        }
        CVTMI_TRY(S.io_d.reserve(db));
        CVTMI_TRY(S.io_i.reserve(ib));
        CVTMI_TRY(S.io_pin.reserve(qb + db + ib));   // [queries | distances | ids]
    }
    struct InFlight { int64_t q0 = 0, n = 0; } fl[2];
    // pinned staging -> the caller's (pageable) arrays: a large copy is shared with a helper thread (one core moves ~10 GB/s: the
    // 12 MB of a 10 000 x 100 result would otherwise cost a third of the scan's time)
    const auto copy_out = [](void *dst, const void *src, size_t bytes) {
        if (bytes < ((size_t)1 << 20)) { memcpy(dst, src, bytes); return; }
        const size_t half = (bytes / 2) & ~(size_t)63;
        try {   // (thread creation can throw std::system_error: nothing may cross the C ABI)
            std::thread helper([=]() { memcpy(static_cast<char *>(dst) + half, static_cast<const char *>(src) + half, bytes - half); });
            memcpy(dst, src, half);
            helper.join();
        } catch (...) {
            memcpy(dst, src, bytes);
        }
    };
    // results of the chunk a set holds -> the caller's arrays (after its stream has drained)
    const auto drain = [&](int i) -> int {
        if (!fl[i].n) return CVTMI_OK;
        OpqScratch &S = *lease[i].s;
        CVTMI_HIP(stream_wait(lease[i].st));
        copy_out(dist + fl[i].q0 * k, S.io_pin.as<char>() + qb, (size_t)fl[i].n * k * sizeof(float));
        copy_out(ids + fl[i].q0 * k, S.io_pin.as<char>() + qb + db, (size_t)fl[i].n * k * sizeof(int64_t));
        fl[i].n = 0;
        return CVTMI_OK;
    };
    // Small batches (the reference's call pattern: a handful of frames per Query; here whatever takes the small-batch path, up to 128
    // queries): the copies are a third of such a call.  The table kernel reads the queries and the selection kernel writes the results straight from / to the pinned staging area (page-locked
    // host memory is device-visible: one PCIe read of the queries, posted writes of the lists) -- no copy engine in the chain.
    if (chunks == 1 && g_small_zero_copy.load() && h->n > 0 && scans_chosen(h, nq, k)) {
        OpqScratch &S = *lease[0].s;
        hipStream_t st = lease[0].st;
        void *pin_dev = nullptr;
        if (hipHostGetDevicePointer(&pin_dev, S.io_pin.p, 0) == hipSuccess && pin_dev) {
            char *pin = S.io_pin.as<char>(), *pd = static_cast<char *>(pin_dev);
            const size_t qn = (size_t)nq * D * sizeof(float), dn = (size_t)nq * k * sizeof(float), in = (size_t)nq * k * sizeof(int64_t);
            memcpy(pin, q, qn);
            CVTMI_TRY(opq_search_leased(h, S, reinterpret_cast<const float *>(pd), nq, rotate, k, reinterpret_cast<float *>(pd + qb),
                                        reinterpret_cast<int64_t *>(pd + qb + db), st));
            CVTMI_HIP(stream_wait(st));
            memcpy(dist, pin + qb, dn);
            memcpy(ids, pin + qb + db, in);
            return CVTMI_OK;
        }
        (void)hipGetLastError();
    }
    int c = 0;
    // a piece that fails leaves earlier pieces in flight, writing into the caller's arrays or the staging areas: nothing is handed
    // back (and no lease released) before both streams have drained
    const auto fail_after_drain = [&](int rc) -> int {
        for (int j = 0; j < nsets; ++j) (void)stream_wait(lease[j].st);
        return rc;
    };
    for (int64_t q0 = 0; q0 < nq; q0 += per, ++c) {
        const int i = c % nsets;
        const int64_t n = std::min(per, nq - q0);
        CVTMI_TRY(drain(i));   // the set's previous chunk: its staging area is about to be overwritten
        OpqScratch &S = *lease[i].s;
        hipStream_t st = lease[i].st;
        const void *src = q + q0 * D;
        if (zd && c >= nsets) CVTMI_HIP(stream_wait(st));   // (zero-copy pieces: the set's query upload of two pieces ago must have been consumed)
        if (!q_pinned) { memcpy(S.io_pin.p, src, (size_t)n * D * sizeof(float)); src = S.io_pin.p; }
        CVTMI_HIP(hipMemcpyAsync(S.io_q.p, src, (size_t)n * D * sizeof(float), hipMemcpyHostToDevice, st));
        if (zd) {   // (normally one piece: the final wait below is all that is left)
            const int rc = opq_search_leased(h, S, S.io_q.as<float>(), n, rotate, k, zd + q0 * k, zi + q0 * k, st);
            if (rc != CVTMI_OK) return fail_after_drain(rc);
            continue;
        }
        {
            const int rc = opq_search_leased(h, S, S.io_q.as<float>(), n, rotate, k, S.io_d.as<float>(), S.io_i.as<int64_t>(), st);
            if (rc != CVTMI_OK) return fail_after_drain(rc);
        }
        if (out_pinned) {  // straight into the caller's page-locked arrays; the final drain only waits for the streams
            CVTMI_HIP(hipMemcpyAsync(dist + q0 * k, S.io_d.p, (size_t)n * k * sizeof(float), hipMemcpyDeviceToHost, st));
            CVTMI_HIP(hipMemcpyAsync(ids + q0 * k, S.io_i.p, (size_t)n * k * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        } else {
            CVTMI_HIP(hipMemcpyAsync(S.io_pin.as<char>() + qb, S.io_d.p, (size_t)n * k * sizeof(float), hipMemcpyDeviceToHost, st));
            CVTMI_HIP(hipMemcpyAsync(S.io_pin.as<char>() + qb + db, S.io_i.p, (size_t)n * k * sizeof(int64_t), hipMemcpyDeviceToHost, st));
            fl[i].q0 = q0; fl[i].n = n;
        }
    }
    if (out_pinned)
        for (int i = 0; i < nsets; ++i) CVTMI_HIP(stream_wait(lease[i].st));
    for (int i = 0; i < nsets; ++i) CVTMI_TRY(drain((c + i) % nsets));
    return CVTMI_OK;
}

// row-sharded search: the local scan writes its lists straight into this rank's slot of the communicator's gather buffer,
// then one all-gather + merge (shard.hip).  Lock order: communicator first, then the handle -- collectives on one
// communicator leave in the order its lock was taken, which therefore has to be the same on every rank (drive one
// communicator from one thread, or issue the searches that share it in one order everywhere).
int cvtmi_opq_search_sharded_dev(cvtmi_opq_t h, cvtmi_comm_t c, const float *q, int64_t nq, int rotate, int k, float *dist,
                                 int64_t *ids, void *stream)
{
    CVTMI_TRY(comm_validate(c));
    if (!h) return fail(CVTMI_EINVAL, "cvtmi_opq_search_sharded: null handle");
    if (nq < 0 || (nq > 0 && (!dist || !ids))) return fail(CVTMI_EINVAL, "cvtmi_opq_search_sharded: bad arguments");
    if (k < 1 || k > CVTMI_K_MAX) return fail(CVTMI_EUNSUPPORTED, "cvtmi_opq_search_sharded: k=%d outside 1..%d", k, CVTMI_K_MAX);
    if (nq == 0) return CVTMI_OK;
    Serial serial_c(*comm_sync(c), (hipStream_t)stream);
    CHECK_H(h);
    if (comm_world(c) == 1 && !comm_has_transport(c)) return cvtmi_opq_search_dev(h, q, nq, rotate, k, dist, ids, stream);
    // from here on every rank reaches the collective, whatever its local search did
    int rc = comm_device(c) != h->device ? fail(CVTMI_EINVAL, "cvtmi_opq_search_sharded: handle and communicator live on different devices") : CVTMI_OK;
    float *sd = nullptr;
    int64_t *si = nullptr;
    if (rc == CVTMI_OK) rc = !q ? fail(CVTMI_EINVAL, "cvtmi_opq_search_sharded: null queries") : sharded_local_failure(c);
    if (rc == CVTMI_OK) rc = comm_local_slot(c, nq, k, &sd, &si);
    if (rc == CVTMI_OK) rc = cvtmi_opq_search_dev(h, q, nq, rotate, k, sd, si, stream);
    return comm_exchange_merge(c, nq, k, rc, dist, ids, (hipStream_t)stream);
}

int cvtmi_opq_search_sharded(cvtmi_opq_t h, cvtmi_comm_t c, const float *q, int64_t nq, int rotate, int k, float *dist, int64_t *ids)
{
    CVTMI_TRY(comm_validate(c));
    CHECK_H(h);
    if (nq < 0 || (nq > 0 && (!q || !dist || !ids))) return fail(CVTMI_EINVAL, "cvtmi_opq_search_sharded: bad arguments");
    if (k < 1 || k > CVTMI_K_MAX) return fail(CVTMI_EUNSUPPORTED, "cvtmi_opq_search_sharded: k=%d outside 1..%d", k, CVTMI_K_MAX);
    if (nq == 0) return CVTMI_OK;
    Tmp dq, dd, di;
    CVTMI_TRY(dq.upload(q, (size_t)nq * h->m.D * sizeof(float)));
    CVTMI_TRY(dd.alloc((size_t)nq * k * sizeof(float)));
    CVTMI_TRY(di.alloc((size_t)nq * k * sizeof(int64_t)));
    CVTMI_TRY(cvtmi_opq_search_sharded_dev(h, c, dq.as<float>(), nq, rotate, k, dd.as<float>(), di.as<int64_t>(), nullptr));
    CVTMI_HIP(hipMemcpy(dist, dd.p, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost));
    CVTMI_HIP(hipMemcpy(ids, di.p, (size_t)nq * k * sizeof(int64_t), hipMemcpyDeviceToHost));
    {   // the copies above synchronised: a failure the deferred status check saw in THIS search is known now
        Serial serial_c(*comm_sync(c), nullptr);
        CVTMI_TRY(comm_take_deferred(c));
    }
    return CVTMI_OK;
}

int cvtmi_opq_search_sharded_all(cvtmi_opq_t *handles, cvtmi_comm_t *comms, int ndev, const float *q, int64_t nq, int rotate, int k,
                                 float *dist, int64_t *ids)
{
    if (!handles || !comms || ndev < 1) return fail(CVTMI_EINVAL, "cvtmi_opq_search_sharded_all: bad arguments");
    if (nq < 0 || (nq > 0 && (!q || !dist || !ids))) return fail(CVTMI_EINVAL, "cvtmi_opq_search_sharded_all: bad arguments");
    if (k < 1 || k > CVTMI_K_MAX) return fail(CVTMI_EUNSUPPORTED, "cvtmi_opq_search_sharded_all: k=%d outside 1..%d", k, CVTMI_K_MAX);
    std::vector<int> devices(ndev);
    for (int d = 0; d < ndev; ++d) {
        CVTMI_TRY(comm_validate(comms[d]));
        if (!handles[d]) return fail(CVTMI_EINVAL, "cvtmi_opq_search_sharded_all: null handle %d", d);
        if (comm_world(comms[d]) != ndev || comm_rank(comms[d]) != d || comm_device(comms[d]) != handles[d]->device)
            return fail(CVTMI_EINVAL, "cvtmi_opq_search_sharded_all: communicator %d does not belong to handle %d", d, d);
        devices[d] = handles[d]->device;
    }
    if (nq == 0) return CVTMI_OK;
    return sharded_all(comms, ndev, q, (size_t)nq * handles[0]->m.D * sizeof(float), nq, k, dist, ids, devices.data(),
                       [&](int d, const void *qd, float *sd, int64_t *si) {
                           return cvtmi_opq_search_dev(handles[d], static_cast<const float *>(qd), nq, rotate, k, sd, si, nullptr);
                       });
}

// IVFOPQ::Query / QueryThrehold (IVFOPQ.cpp:213-422) only read the index: like the exhaustive search, a query leases a scratch set
// (rotated frames, probe lists) and holds the handle shared, so callers on several threads proceed side by side.  The list-ordered
// copy of the entries is (re)built under the exclusive lock by the first query after an append.
static int opq_query_prepare(cvtmi_opq_t h, hipStream_t st)
{
    {
        std::shared_lock<std::shared_timed_mutex> rd(h->rw);
        if (h->csr_valid) return CVTMI_OK;
    }
    Serial serial(h->sync, st);
    OpqExclusive excl(h, st);
    return opq_build_csr(h, st);
}
static int opq_query_video_leased(cvtmi_opq_t h, OpqScratch &S, const float *q, int64_t nq, int rotate, int nprobe, int img_num, float *match_score,
                                  hipStream_t st)
{
    if (!h->csr_valid) return fail(CVTMI_EINVAL, "cvtmi_opq_query_video: the index changed while the query was being prepared");
    if (nprobe > h->m.coarseK) nprobe = h->m.coarseK;  // the reference pops an empty heap here (UB)
    if (h->csr_kept > 0 && (h->csr_vmin < 0 || h->csr_vmax >= img_num))
        return fail(CVTMI_EINVAL, "cvtmi_opq_query_video: video id %d outside img_num=%d", h->csr_vmin < 0 ? h->csr_vmin : h->csr_vmax, img_num);
    const float *q_rot = q;
    if (rotate && (h->m.perm || h->m.R)) {
        CVTMI_TRY(S.s_qrot.reserve((size_t)nq * h->m.D * sizeof(float)));
        CVTMI_TRY(opq_rotate_impl(h, q, nq, S.s_qrot.as<float>(), st));
        q_rot = S.s_qrot.as<float>();
    }
    const size_t probe_bytes = ((size_t)nq * nprobe * sizeof(int32_t) + 15) / 16 * 16;
    CVTMI_TRY(S.s_probe.reserve(probe_bytes + coarse_probe_scratch_bytes(nq, nprobe)));
    CVTMI_TRY(launch_coarse_probe(h->m, q_rot, nq, nprobe, S.s_probe.as<int32_t>(), st, S.s_probe.as<char>() + probe_bytes));
    return launch_query_video(h->m, q_rot, nq, nprobe, S.s_probe.as<int32_t>(), h->csr_off.as<int64_t>(), h->csr_codes.as<uint8_t>(),
                              h->csr_videos.as<int32_t>(), img_num, match_score, h->csr_longest, st);
}

int cvtmi_opq_query_video_dev(cvtmi_opq_t h, const float *q, int64_t nq, int rotate, int nprobe, int img_num, float *match_score,
                              void *stream)
{
    CHECK_H(h);
    if (nq < 0 || img_num < 0 || (nq > 0 && img_num > 0 && (!q || !match_score)))
        return fail(CVTMI_EINVAL, "cvtmi_opq_query_video: bad arguments");
    if (nprobe < 1) return fail(CVTMI_EINVAL, "cvtmi_opq_query_video: nprobe=%d", nprobe);
    if (nq == 0 || img_num == 0) return CVTMI_OK;
    hipStream_t st = (hipStream_t)stream;
    for (int attempt = 0;; ++attempt) {   // (an append between the preparation and the shared lock sends the query round again)
        CVTMI_TRY(opq_query_prepare(h, st));
        std::shared_lock<std::shared_timed_mutex> rd(h->rw);
        if (!h->csr_valid && attempt < 8) continue;
        OpqLease lease;
        CVTMI_TRY(lease.open(h, st, false));
        return opq_query_video_leased(h, *lease.s, q, nq, rotate, nprobe, img_num, match_score, st);
    }
}

// host pointers: frames in and the dense [frames][videos] score matrix out through the leased set's staging buffers (kept between
// calls: the reference's own call is a handful of frames, a device allocation per call would cost more than the query)
int cvtmi_opq_query_video(cvtmi_opq_t h, const float *q, int64_t nq, int rotate, int nprobe, int img_num,
                          float *match_score)
{
    CHECK_H(h);
    if (nq < 0 || img_num < 0 || (nq > 0 && img_num > 0 && (!q || !match_score)))
        return fail(CVTMI_EINVAL, "cvtmi_opq_query_video: bad arguments");
    if (nprobe < 1) return fail(CVTMI_EINVAL, "cvtmi_opq_query_video: nprobe=%d", nprobe);
    if (nq == 0 || img_num == 0) return CVTMI_OK;
    for (int attempt = 0;; ++attempt) {
        CVTMI_TRY(opq_query_prepare(h, nullptr));
        std::shared_lock<std::shared_timed_mutex> rd(h->rw);
        if (!h->csr_valid && attempt < 8) continue;
        OpqLease lease;
        CVTMI_TRY(lease.open(h, nullptr, true));
        OpqScratch &S = *lease.s;
        hipStream_t st = lease.st;
        const size_t qb = (size_t)nq * h->m.D * sizeof(float), mb = (size_t)nq * img_num * sizeof(float);
        CVTMI_TRY(S.io_q.reserve(qb));
        CVTMI_TRY(S.io_d.reserve(mb));
        CVTMI_HIP(hipMemcpyAsync(S.io_q.p, q, qb, hipMemcpyHostToDevice, st));
        CVTMI_TRY(opq_query_video_leased(h, S, S.io_q.as<float>(), nq, rotate, nprobe, img_num, S.io_d.as<float>(), st));
        CVTMI_HIP(hipMemcpyAsync(match_score, S.io_d.p, mb, hipMemcpyDeviceToHost, st));
        CVTMI_HIP(stream_wait(st));
        return CVTMI_OK;
    }
}

int cvtmi_opq_set_param(cvtmi_opq_t h, const char *name, int64_t value)
{
    if (!h || !name) return fail(CVTMI_EINVAL, "cvtmi_opq_set_param: null");
    if (!strcmp(name, "splits")) { h->p_splits = (int)value; return CVTMI_OK; }
    if (!strcmp(name, "tail_split")) { h->p_tail = value != 0; return CVTMI_OK; }
    if (!strcmp(name, "prerotate")) { h->p_prerot = value != 0; return CVTMI_OK; }
    if (!strcmp(name, "groups_a")) { h->p_groups_a = (int)value; return CVTMI_OK; }
    if (!strcmp(name, "splits_b")) { h->p_splits_b = (int)value; return CVTMI_OK; }
    if (!strcmp(name, "qtile")) {
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8)
            return fail(CVTMI_EINVAL, "cvtmi_opq_set_param: qtile must be 0, 1, 2, 4 or 8");
        h->p_qtile = (int)value;
        return CVTMI_OK;
    }
    if (!strcmp(name, "profile")) { h->p_profile = value != 0; return CVTMI_OK; }
    if (!strcmp(name, "scan_lazy")) { h->p_lazy = value != 0; return CVTMI_OK; }
    if (!strcmp(name, "scan_small")) { h->p_small = value != 0; return CVTMI_OK; }
    if (!strcmp(name, "scan_share")) { h->p_share = value != 0; return CVTMI_OK; }
    if (!strcmp(name, "encode_variant")) {
        if (value < 0 || value > 2) return fail(CVTMI_EINVAL, "cvtmi_opq_set_param: encode_variant must be 0, 1 or 2");
        h->p_encode = (int)value;
        return CVTMI_OK;
    }
    if (!strcmp(name, "scan_variant")) {
        if (value < 0 || value > 7) return fail(CVTMI_EINVAL, "cvtmi_opq_set_param: scan_variant must be 0..7");
        h->p_variant = (int)value;
        return CVTMI_OK;
    }
    return fail(CVTMI_EINVAL, "cvtmi_opq_set_param: unknown parameter '%s'", name);
}

int cvtmi_opq_last_scan(cvtmi_opq_t h, float *ms, int64_t *code_bytes, int *qtile, int *splits)
{
    CHECK_H_SERIAL(h, nullptr);
    if (h->ev_count == 0) return fail(CVTMI_ESTATE, "cvtmi_opq_last_scan: no profiled search since the last call");
    const int cnt = h->ev_count < cvtmi_opq_s::kEvRing ? h->ev_count : cvtmi_opq_s::kEvRing;
    double sum = 0.0;
    for (int e = 0; e < cnt; ++e) {
        CVTMI_HIP(hipEventSynchronize(h->ev1[e]));
        float t = 0.f;
        CVTMI_HIP(hipEventElapsedTime(&t, h->ev0[e], h->ev1[e]));
        sum += t;
    }
    h->ev_count = 0;
    if (ms) *ms = (float)(sum / cnt);
    std::lock_guard<std::mutex> g(h->pool_mu);
    if (code_bytes) *code_bytes = h->last_bytes;
    if (qtile) *qtile = h->last_qt;
    if (splits) *splits = h->last_splits;
    return CVTMI_OK;
}

// page-locked host memory for the arrays of the host-pointer entries (queries in, results out)
int cvtmi_host_alloc(size_t bytes, void **p)
{
    if (!p) return fail(CVTMI_EINVAL, "cvtmi_host_alloc: null");
    *p = nullptr;
    hipError_t e = hipHostMalloc(p, bytes ? bytes : 16, hipHostMallocDefault);
    if (e != hipSuccess) { *p = nullptr; return fail(CVTMI_ENOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
    return CVTMI_OK;
}

int cvtmi_host_free(void *p)
{
    if (p) CVTMI_HIP(hipHostFree(p));
    return CVTMI_OK;
}

// ================================================================ merge =======================
int cvtmi_topk_merge_dev(const float *in_dist, const int64_t *in_ids, int64_t nq, int L, int k, float *dist, int64_t *ids,
                         void *stream)
{
    if (nq < 0 || L < 1 || (nq > 0 && (!in_dist || !in_ids || !dist || !ids)))
        return fail(CVTMI_EINVAL, "cvtmi_topk_merge: bad arguments");
    return launch_topk_merge(in_dist, in_ids, nq, L, k, dist, ids, (hipStream_t)stream);
}

int cvtmi_topk_merge(const float *in_dist, const int64_t *in_ids, int64_t nq, int L, int k, float *dist, int64_t *ids)
{
    if (nq < 0 || L < 1 || k < 1 || (nq > 0 && (!in_dist || !in_ids || !dist || !ids)))
        return fail(CVTMI_EINVAL, "cvtmi_topk_merge: bad arguments");
    if (nq == 0) return CVTMI_OK;
    const size_t cnt = (size_t)nq * L * k, oc = (size_t)nq * k;
    Tmp a, b, c, d;
    CVTMI_TRY(a.upload(in_dist, cnt * sizeof(float)));
    CVTMI_TRY(b.upload(in_ids, cnt * sizeof(int64_t)));
    CVTMI_TRY(c.alloc(oc * sizeof(float)));
    CVTMI_TRY(d.alloc(oc * sizeof(int64_t)));
    CVTMI_TRY(cvtmi_topk_merge_dev(a.as<float>(), b.as<int64_t>(), nq, L, k, c.as<float>(), d.as<int64_t>(), nullptr));
    CVTMI_HIP(hipMemcpy(dist, c.p, oc * sizeof(float), hipMemcpyDeviceToHost));
    CVTMI_HIP(hipMemcpy(ids, d.p, oc * sizeof(int64_t), hipMemcpyDeviceToHost));
    return CVTMI_OK;
}

int cvtmi_topk_select_dev(const float *scores, int64_t nq, int64_t n, int k, float *dist, int64_t *ids, void *stream)
{
    if (nq < 0 || n < 0 || (nq > 0 && (!dist || !ids || (n > 0 && !scores))))
        return fail(CVTMI_EINVAL, "cvtmi_topk_select: bad arguments");
    return launch_topk_select(scores, nullptr, nq, n, k, dist, ids, (hipStream_t)stream);
}

int cvtmi_topk_select(const float *scores, int64_t nq, int64_t n, int k, float *dist, int64_t *ids)
{
    if (nq < 0 || n < 0 || k < 1 || (nq > 0 && (!dist || !ids || (n > 0 && !scores))))
        return fail(CVTMI_EINVAL, "cvtmi_topk_select: bad arguments");
    if (nq == 0) return CVTMI_OK;
    Tmp a, c, d;
    CVTMI_TRY(a.upload(scores, (size_t)nq * n * sizeof(float)));
    CVTMI_TRY(c.alloc((size_t)nq * k * sizeof(float)));
    CVTMI_TRY(d.alloc((size_t)nq * k * sizeof(int64_t)));
    CVTMI_TRY(cvtmi_topk_select_dev(a.as<float>(), nq, n, k, c.as<float>(), d.as<int64_t>(), nullptr));
    CVTMI_HIP(hipMemcpy(dist, c.p, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost));
    CVTMI_HIP(hipMemcpy(ids, d.p, (size_t)nq * k * sizeof(int64_t), hipMemcpyDeviceToHost));
    return CVTMI_OK;
}

// ================================================================ flat ========================
int cvtmi_flat_create(int metric, int D, cvtmi_flat_t *out)
{
    if (!out) return fail(CVTMI_EINVAL, "cvtmi_flat_create: null out");
    *out = nullptr;
    if (metric != CVTMI_METRIC_IP && metric != CVTMI_METRIC_L2F && metric != CVTMI_METRIC_L2U8)
        return fail(CVTMI_EINVAL, "cvtmi_flat_create: unknown metric %d", metric);
    if (D < 1 || D > 4096) return fail(CVTMI_EUNSUPPORTED, "cvtmi_flat_create: D=%d outside 1..4096", D);
    int dev = 0;
    CVTMI_HIP(hipGetDevice(&dev));
    cvtmi_flat_s *h = new (std::nothrow) cvtmi_flat_s();
    if (!h) return fail(CVTMI_ENOMEM, "cvtmi_flat_create: out of host memory");
    h->device = dev; h->metric = metric; h->D = D;
    h->row_bytes = metric == CVTMI_METRIC_L2U8 ? (size_t)D : (size_t)D * sizeof(float);
    *out = h;
    return CVTMI_OK;
}

int cvtmi_flat_destroy(cvtmi_flat_t h)
{
    if (!h) return CVTMI_OK;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    for (DevBuf *b : { &h->data, &h->labels, &h->norms, &h->add_stage, &h->f_pack, &h->f_bias, &h->f_istats, &h->fs_bias, &h->fs_stats, &h->f_rows })
        b->release();
    for (FlatScratch *c : h->pool) { c->release_all(); delete c; }
    if (h->mutated) (void)hipEventDestroy(h->mutated);
    delete h;
    return CVTMI_OK;
}

__global__ void iota_i64_kernel(int64_t *p, int64_t begin, int64_t end)
{
    for (int64_t i = begin + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < end; i += (int64_t)gridDim.x * kBlock) p[i] = i;
}

static int flat_add_common(cvtmi_flat_t h, const void *x, const int64_t *labels, int64_t n, hipMemcpyKind kind,
                           hipStream_t st)
{
    if (n < 0 || (n > 0 && !x)) return fail(CVTMI_EINVAL, "cvtmi_flat_add: bad arguments");
    if (n == 0) return CVTMI_OK;
    const int64_t total = h->n + n;
    if (total > 0xfffffffeLL) return fail(CVTMI_EUNSUPPORTED, "cvtmi_flat_add: more than 2^32-2 rows per handle");
    // (the operand copies keep covering their rows; flat_prepare packs the appended ones)
    bool explicit_labels = labels != nullptr;
    if (explicit_labels && kind == hipMemcpyHostToDevice && h->identity) {
        bool same = true;
        for (int64_t i = 0; i < n && same; ++i) same = labels[i] == h->n + i;
        if (same) explicit_labels = false;
    }
    const bool blocked = flat_blocked(h->metric, h->D);  // fp32 rows live in 64-row blocks: whole blocks are kept
    const size_t rows_held = blocked ? (size_t)((h->n + 63) / 64 * 64) : (size_t)h->n;
    const size_t rows_need = blocked ? (size_t)((total + 63) / 64 * 64) : (size_t)total;
    if (rows_need * h->row_bytes > h->data.cap) {
        size_t rows = std::max<size_t>(rows_need, (h->data.cap / h->row_bytes) * 2);
        rows = std::max<size_t>(rows, 1024);
        CVTMI_TRY(h->data.grow(rows * h->row_bytes, rows_held * h->row_bytes, st));
    }
    if (explicit_labels && h->identity) {
        h->identity = false;
        CVTMI_TRY(h->labels.grow((size_t)std::max<int64_t>(total, 1024) * 8, 0, st));
        if (h->n) {
            hipLaunchKernelGGL(iota_i64_kernel, dim3(1024), dim3(kBlock), 0, st, h->labels.as<int64_t>(), (int64_t)0, h->n);
            CVTMI_HIP(hipGetLastError());
        }
    }
    if (!h->identity && (size_t)total * 8 > h->labels.cap)
        CVTMI_TRY(h->labels.grow(std::max<size_t>((size_t)total, h->labels.cap / 4) * 8, (size_t)h->n * 8, st));
    if (blocked) {
        const float *src = static_cast<const float *>(x);
        if (kind == hipMemcpyHostToDevice || ((uintptr_t)x & 15) != 0) {  // staged: host rows, or a device pointer off 16 bytes
            CVTMI_TRY(h->add_stage.reserve((size_t)n * h->row_bytes));
            CVTMI_HIP(hipMemcpyAsync(h->add_stage.p, x, (size_t)n * h->row_bytes, kind, st));
            src = h->add_stage.as<float>();
        }
        CVTMI_TRY(launch_flat_block(src, n, h->D, h->n, h->data.as<float>(), st));
        if (flat_f32_stream_qmax(h->D) > 0 || flat_f32_tfilter_width(h->D)) {   // score bias + row statistics of the streaming search / the threshold filter, padding rows zeroed
            if (!h->fs_stats.p) {
                CVTMI_TRY(h->fs_stats.reserve(16));
                CVTMI_HIP(hipMemsetAsync(h->fs_stats.p, 0, 16, st));
            }
            if (rows_need * 4 > h->fs_bias.cap)
                CVTMI_TRY(h->fs_bias.grow(std::max<size_t>(rows_need, h->fs_bias.cap / 2) * 4, rows_held * 4, st));
            CVTMI_TRY(launch_flat_f32_bias(h->data.as<float>(), h->D, h->metric, h->n, total, h->fs_bias.as<float>(), h->fs_stats.as<uint32_t>(), st));
            h->fs_stats_n = -1;
        }
    } else {
        CVTMI_HIP(hipMemcpyAsync(h->data.as<uint8_t>() + (size_t)h->n * h->row_bytes, x, (size_t)n * h->row_bytes, kind, st));
    }
    if (!h->identity) {
        if (labels) CVTMI_HIP(hipMemcpyAsync(h->labels.as<int64_t>() + h->n, labels, (size_t)n * 8, kind, st));
        else {
            hipLaunchKernelGGL(iota_i64_kernel, dim3(1024), dim3(kBlock), 0, st, h->labels.as<int64_t>(), h->n, total);
            CVTMI_HIP(hipGetLastError());
        }
    }
    if (h->metric == CVTMI_METRIC_L2U8 && h->D % 32 == 0 && h->D <= 512) {
        if ((size_t)total * 4 > h->norms.cap)
            CVTMI_TRY(h->norms.grow(std::max<size_t>((size_t)total, h->norms.cap / 2) * 4, (size_t)h->n * 4, st));
        CVTMI_TRY(launch_flat_u8_norms(h->data.as<uint8_t>() + (size_t)h->n * h->row_bytes, n, h->D,
                                       h->norms.as<int32_t>() + h->n, st));
    }
    if (kind == hipMemcpyHostToDevice) CVTMI_HIP(stream_wait(st));
    h->n = total;
    return CVTMI_OK;
}

// a mutation on stream st: exclusive, ordered after every search that is still in flight on another stream, and searches
// that come later on other streams wait for it (FlatLease::open)
struct FlatMutation {
    cvtmi_flat_s *h;
    hipStream_t st;
    std::unique_lock<std::shared_timed_mutex> lk;
    FlatMutation(cvtmi_flat_s *handle, hipStream_t stream) : h(handle), st(stream), lk(handle->rw)
    {
        std::lock_guard<std::mutex> g(h->pool_mu);
        for (FlatScratch *c : h->pool)
            if (c->pending && c->last != st) (void)hipStreamWaitEvent(st, c->done, 0);
        if (h->mut_pending && h->mut_stream != st) (void)hipStreamWaitEvent(st, h->mutated, 0);
    }
    ~FlatMutation()
    {
        if (!h->mutated) (void)hipEventCreateWithFlags(&h->mutated, hipEventDisableTiming);
        if (h->mutated && hipEventRecord(h->mutated, st) == hipSuccess) { h->mut_stream = st; h->mut_pending = true; }
    }
};

int cvtmi_flat_add(cvtmi_flat_t h, const void *x, const int64_t *labels, int64_t n)
{
    CHECK_H(h);
    FlatMutation mut(h, nullptr);
    return flat_add_common(h, x, labels, n, hipMemcpyHostToDevice, nullptr);
}

int cvtmi_flat_add_dev(cvtmi_flat_t h, const void *x, const int64_t *labels, int64_t n, void *stream)
{
    CHECK_H(h);
    FlatMutation mut(h, (hipStream_t)stream);
    return flat_add_common(h, x, labels, n, hipMemcpyDeviceToDevice, (hipStream_t)stream);
}

int cvtmi_flat_ntotal(cvtmi_flat_t h, int64_t *n)
{
    if (!h || !n) return fail(CVTMI_EINVAL, "cvtmi_flat_ntotal: null");
    *n = h->n;
    return CVTMI_OK;
}

int cvtmi_flat_reset(cvtmi_flat_t h)
{
    CHECK_H(h);
    FlatMutation mut(h, nullptr);
    h->n = 0; h->identity = true; h->f_pack_n = -1; h->f_rows_n = -1;
    h->fs_stats_n = -1; h->fs_nonfinite = false;
    (void)hipDeviceSynchronize();
    if (h->fs_stats.p) CVTMI_HIP(hipMemset(h->fs_stats.p, 0, 16));
    h->f_pack.release(); h->f_bias.release(); h->f_rows.release(); h->f_rows_failed = false;  // the filter's copies are as large as the rows: give them back
    return CVTMI_OK;
}

// the exact search over rows [0, n_rows) of the handle: k smallest (distance, row) per query, rows not yet mapped to labels
// max_stream_passes: the uint8 streaming kernel serves 128 queries per pass; callers that search a short row range for many queries (the
// filter pipeline's sample stage) cap the passes and fall through to the row-tile kernels beyond
static int flat_search_rows(cvtmi_flat_t h, FlatScratch &S, int64_t n_rows, const void *q, int64_t nq, int k, float *dist, int64_t *rows, hipStream_t st,
                            int64_t max_stream_passes = INT64_MAX, const uint32_t *only_if = nullptr)
{
    // uint8: anything the filter pipeline did not take goes through the streaming matrix-core kernel, 128 queries per pass (its cost hardly
    // depends on k: 10 M x 512-d, k = 128: nq = 1000 40.6 -> 10 ms, nq = 4096 117 -> 40 ms against the row-tile kernels)
    if (h->metric == CVTMI_METRIC_L2U8 && g_flat_variant != 1 && h->norms.p && nq >= 1 && flat_u8_mstream_applies(h->D, n_rows, std::min<int64_t>(nq, 128), k) &&
        ((uintptr_t)q & 15) == 0 && (nq + 127) / 128 <= max_stream_passes && !only_if) {   // (a predicated run: the row-per-lane kernels, which take one)
        const int64_t passes = (nq + 127) / 128, per = (nq + passes - 1) / passes;   // balanced: 129 queries = 65 + 64
        const int NS = flat_u8_stream_slices();
        int nqp = 0, waves = 0;
        const size_t bytes = flat_u8_mstream_scratch(n_rows, per, &nqp, &waves);
        CVTMI_TRY(S.s_stage.reserve(bytes));
        CVTMI_TRY(S.s_part_d.reserve((size_t)per * NS * k * sizeof(float)));
        CVTMI_TRY(S.s_part_id.reserve((size_t)per * NS * k * sizeof(int64_t)));
        for (int64_t a = 0; a < nq; a += per) {
            const int64_t m = std::min(per, nq - a);
            (void)flat_u8_mstream_scratch(n_rows, m, &nqp, &waves);
            const uint8_t *qa = reinterpret_cast<const uint8_t *>(q) + a * h->D;
            int32_t *tmin = S.s_stage.as<int32_t>(), *wmin = tmin + (size_t)flat_u8_mstream_groups(n_rows) * nqp;
            CVTMI_TRY(launch_flat_u8_mstream(h->D, h->data.as<uint8_t>(), h->norms.as<int32_t>(), n_rows, qa, m, tmin, wmin, st));
            CVTMI_TRY(launch_flat_u8_mstream_finish(h->D, h->data.as<uint8_t>(), n_rows, qa, m, k, wmin, waves, tmin, nqp, flat_u8_mstream_group(), S.s_part_d.as<float>(),
                                                    S.s_part_id.as<int64_t>(), dist + a * k, rows + a * k, st));
        }
        return CVTMI_OK;
    }
    const bool mfma = h->metric == CVTMI_METRIC_L2U8 && !only_if && flat_u8_mfma_qtile(h->D, k, nq) > 0;
    const int qt = mfma ? flat_u8_mfma_qtile(h->D, k, nq) : (k > 128 ? 1 : flat_qtile(nq));   // k > 128: one query per workgroup (kernels.h: kBigK)
    int splits = mfma ? flat_u8_mfma_splits(n_rows, nq, qt) : flat_plan_splits(n_rows, nq, qt);
    if (mfma && splits >= 8) splits = (splits / 8) * 8;  // a row split per XCD: query groups share its L2
    // a predicated re-run normally finds nothing to do, and what it finds is a few queries: its row splits do not follow the plan for the
    // whole batch (1000 queries: one or two splits -- ONE flagged query then waited for a single workgroup to read every row: 5 ms on
    // 0.5 GB of 300-d rows) but are 32 wherever the rows allow it; an empty workgroup costs a dispatch and the read of its flags
    // (never fewer than the plan's own: a small batch has few query groups and the plan cuts the rows finer for it -- 15 queries over 300 000 x
    //  2048-d rows, 11 of them flagged: 146 splits instead of 32, 5.4 -> 1.9 ms)
    if (only_if && !mfma) {
        splits = (int)std::max<int64_t>(splits, std::min<int64_t>(32, n_rows / 8192));
        // (the partial lists of a re-run are sized for every query, flagged or not: at most ~1 GB of them)
        const int64_t room = std::max<int64_t>(1, (int64_t)(1LL << 30) / std::max<int64_t>(1, nq * (int64_t)k * 12));
        if (splits > room) splits = (int)room;
    }
    float *pd = dist;
    int64_t *pi = rows;
    if (splits > 1) {
        const size_t cnt = (size_t)nq * splits * k;
        CVTMI_TRY(S.s_part_d.reserve(cnt * sizeof(float)));
        CVTMI_TRY(S.s_part_id.reserve(cnt * sizeof(int64_t)));
        pd = S.s_part_d.as<float>();
        pi = S.s_part_id.as<int64_t>();
    }
    if (mfma) {
        CVTMI_TRY(S.s_gthr.reserve((size_t)nq * (1 + 16) * sizeof(uint32_t)));
        CVTMI_TRY(launch_flat_u8_mfma(h->D, h->data.as<uint8_t>(), h->norms.as<int32_t>(), n_rows,
                                      reinterpret_cast<const uint8_t *>(q), nq, k, splits, pd, pi, S.s_gthr.as<uint32_t>(), st));
    }
    else
        CVTMI_TRY(launch_flat_search(h->metric, h->D, h->data.p, n_rows, q, nq, k, qt, splits, pd, pi, st, only_if));
    if (splits > 1) CVTMI_TRY(launch_topk_merge(pd, pi, nq, splits, k, dist, rows, st, only_if));
    return CVTMI_OK;
}

// fp32 search as a stream over the rows (flat_f32_stream.hip).  *done = false: not applicable, the other paths answer
static int flat_search_streamed(cvtmi_flat_t h, FlatScratch &S, const float *q, int64_t nq, int k, float *dist, int64_t *rows, hipStream_t st, bool *done,
                                int *how = nullptr)
{
    *done = false;
    const int D = h->D;
    const int64_t n = h->n;
    if (!h->fs_bias.p || !h->fs_stats.p || h->fs_stats_n != n || h->fs_nonfinite) return CVTMI_OK;
    if (flat_f32_tfilter_applies(h->metric, D, n, nq, k) && h->f_pack.p && h->f_pack_n == n && h->f_pack_nch == flat_f32_tfilter_nch(D) && !h->f_nonfinite &&
        S.fs_scratch.reserve(flat_f32_tfilter_scratch(nq, k)) == CVTMI_OK) {
        // large batches (round 6, flat_f32_tfilter.hip): sample maxima -> per-query threshold -> barrier-free threshold filter (queries in
        // LDS, the rows' bf16 operand copy in registers) -> exact distances of the candidates; flagged queries go through the exact
        // kernels below, as for the stream
        CVTMI_TRY(S.fs_redo.reserve((size_t)nq * 2 * sizeof(uint32_t)));
        CVTMI_TRY(launch_flat_f32_tfilter(h->metric, D, h->data.as<float>(), (h->f_rows.p && h->f_rows_n == n) ? h->f_rows.as<float>() : nullptr, h->f_pack.p, h->f_istats.as<uint32_t>(), h->fs_bias.as<float>(), h->fs_stats.as<uint32_t>(), n, q, nq, k,
                                          S.fs_scratch.p, dist, rows, S.fs_redo.as<uint32_t>(), st));
        CVTMI_TRY(flat_search_rows(h, S, n, q, nq, k, dist, rows, st, INT64_MAX, S.fs_redo.as<uint32_t>()));
        *done = true;
        if (how) *how = 3;
        return CVTMI_OK;
    }
    (void)hipGetLastError();
    if (!flat_f32_stream_applies(h->metric, D, n, k)) return CVTMI_OK;   // (a width only the threshold filter takes)
    const int qmax = flat_f32_stream_qmax(D), qpriv = flat_f32_stream_private_max(D);
    int64_t passes = (nq + qmax - 1) / qmax;
    // just past one private-ring pass, two of them beat one pass of the shared ring (1 M x 128-d, 128 queries: 0.28 against 0.32 ms)
    if (nq > qpriv && nq <= 2 * qpriv) passes = 2;
    const int64_t per = (nq + passes - 1) / passes;
    if (S.fs_scratch.reserve(flat_f32_stream_scratch(D, n, per)) != CVTMI_OK) return CVTMI_OK;   // no room: the exact path answers
    CVTMI_TRY(S.fs_redo.reserve((size_t)nq * 2 * sizeof(uint32_t)));   // redo flags, then list counters
    // round 6: the bf16 operand copy of the threshold filter, when the handle keeps one, is what a small batch streams (half the bytes)
    const bool have_pack = h->f_pack.p && h->f_istats.p && h->f_pack_n == n && D % 16 == 0 && h->f_pack_nch == D / 16 && !h->f_nonfinite;
    for (int64_t a = 0; a < nq; a += per) {
        const int64_t m = std::min(per, nq - a);
        CVTMI_TRY(launch_flat_f32_stream(h->metric, D, h->data.as<float>(), h->fs_bias.as<float>(), h->fs_stats.as<uint32_t>(), n, q + a * D, m, k,
                                         S.fs_scratch.p, dist + a * k, rows + a * k, S.fs_redo.as<uint32_t>() + a,
                                         S.fs_redo.as<uint32_t>() + nq + a, st, have_pack ? h->f_pack.p : nullptr,
                                         have_pack ? h->f_istats.as<uint32_t>() : nullptr,
                                         (h->f_rows.p && h->f_rows_n == n) ? h->f_rows.as<float>() : nullptr));
    }
    // queries the bound does not cover / whose lists ran over: the exact kernels, predicated on the flags (they exit at once otherwise)
    CVTMI_TRY(flat_search_rows(h, S, n, q, nq, k, dist, rows, st, INT64_MAX, S.fs_redo.as<uint32_t>()));
    *done = true;
    return CVTMI_OK;
}

// fp32 search through the matrix-core filter (flat_mfma.hip).  *done = false: not applicable / gave up, take the exact path
static int flat_search_filtered(cvtmi_flat_t h, FlatScratch &S, const float *q, int64_t nq, int k, float *dist, int64_t *rows, hipStream_t st, bool *done)
{
    *done = false;
    const int D = h->D;
    const int64_t n = h->n;
    if (h->f_pack_n != n || h->f_pack_nch != D / 16 || h->f_nonfinite) return CVTMI_OK;   // no operand copy (flat_prepare could not build it) / non-finite rows: exact path
    CVTMI_TRY(S.f_stats.reserve(16));
    CVTMI_HIP(hipMemcpyAsync(S.f_stats.p, h->f_istats.p, 8, hipMemcpyDeviceToDevice, st));   // [0] max |x|^2, [1] non-finite rows; [2], [3] are this call's
    // 1. exact search of a leading sample: its k-th best bounds the global k-th best
    // a smaller sample costs less exact work but doubles the survivors: worth it while k is small
    const int frac = k <= 16 ? 32 : 16;
    int64_t ns = std::max<int64_t>(frac == 32 ? 32768 : 65536, (n / frac + 63) / 64 * 64);
    const int cap = ((frac == 32 ? 48 : 24) * k + 1024 + 63) / 64 * 64;
    CVTMI_TRY(S.f_sd.reserve((size_t)nq * k * sizeof(float)));
    CVTMI_TRY(S.f_si.reserve((size_t)nq * k * sizeof(int64_t)));
    CVTMI_TRY(S.f_thr.reserve((size_t)nq * sizeof(float)));
    CVTMI_TRY(S.f_cnt.reserve((size_t)nq * sizeof(uint32_t)));
    const uint64_t pair_cap64 = (uint64_t)nq * cap;
    const uint32_t pair_cap = pair_cap64 > 0x7ffffff0ull ? 0x7ffffff0u : (uint32_t)pair_cap64;
    // the big scratch (16 bytes per survivor slot + 8 per list entry): if it does not fit, the exact path answers
    if (S.f_cand.reserve((size_t)pair_cap * sizeof(uint4)) != CVTMI_OK || S.f_seld.reserve((size_t)nq * cap * sizeof(float)) != CVTMI_OK ||
        S.f_seli.reserve((size_t)nq * cap * sizeof(int32_t)) != CVTMI_OK)
        return CVTMI_OK;
    CVTMI_TRY(S.f_marg.reserve((size_t)nq * sizeof(float)));
    uint32_t *stats = S.f_stats.as<uint32_t>();  // [0] max |x|^2, [1] non-finite rows, [2] overflow / worst list, [3] pair count
    // one filter stage: given the exact top k of rows [0, r0) in (sd, si), the exact top k of rows [0, r1) into (od, oi):
    // thresholds, filter over [r0, r1), second cut on approximate scores, exact distances of what is left, sort
    auto stage = [&](int64_t r0, int64_t r1, const float *sd, const int64_t *si, float *od, int64_t *oi, uint32_t *worst) -> int {
        CVTMI_HIP(hipMemsetAsync(stats + 2, 0, 8, st));
        CVTMI_TRY(launch_flat_thr(q, nq, D, h->metric, sd, k, stats, S.f_thr.as<float>(), S.f_marg.as<float>(), st));
        CVTMI_HIP(hipMemsetAsync(S.f_cnt.p, 0, (size_t)nq * sizeof(uint32_t), st));
        CVTMI_TRY(launch_flat_filter(q, nq, D, h->f_pack.as<uint4>(), h->f_bias.as<uint32_t>(), S.f_thr.as<float>(), r0, r1, pair_cap,
                                     stats + 3, S.f_cand.as<uint4>(), st));
        CVTMI_TRY(launch_flat_finish(h->metric, h->data.as<float>(), r1, D, q, nq, stats + 3, pair_cap, S.f_cand.as<uint4>(), cap, k,
                                     S.f_marg.as<float>(), sd, si, S.f_cnt.as<uint32_t>(), S.f_seld.as<float>(), S.f_seli.as<int32_t>(),
                                     od, oi, stats + 2, st));
        CVTMI_HIP(hipMemcpyAsync(worst, stats + 2, 4, hipMemcpyDeviceToHost, st));
        CVTMI_HIP(stream_wait(st));
        return CVTMI_OK;
    };
    // 1. the exact top k of the leading ns rows.  The exact kernels only see a sample of the sample (ns / 16 rows); a first
    //    filter stage extends it to ns (falling back to the exact kernels on all ns rows if a list runs over)
    uint32_t worst = 0;
    const int64_t ns0 = std::max<int64_t>(8192, (ns / 16 + 63) / 64 * 64);
    bool have_sample = false;
    if (ns0 * 4 <= ns && nq >= 256) {  // (small batches: the extra launches and the sync cost more than the exact work saved)
        CVTMI_TRY(S.f_sd2.reserve((size_t)nq * k * sizeof(float)));
        CVTMI_TRY(S.f_si2.reserve((size_t)nq * k * sizeof(int64_t)));
        CVTMI_TRY(flat_search_rows(h, S, ns0, q, nq, k, S.f_sd2.as<float>(), S.f_si2.as<int64_t>(), st));
        CVTMI_TRY(stage(ns0, ns, S.f_sd2.as<float>(), S.f_si2.as<int64_t>(), S.f_sd.as<float>(), S.f_si.as<int64_t>(), &worst));
        have_sample = worst <= (uint32_t)cap;
    }
    if (!have_sample) CVTMI_TRY(flat_search_rows(h, S, ns, q, nq, k, S.f_sd.as<float>(), S.f_si.as<int64_t>(), st));
    // 2. the remaining rows
    CVTMI_TRY(stage(ns, n, S.f_sd.as<float>(), S.f_si.as<int64_t>(), dist, rows, &worst));
    h->f_last_worst = (long long)worst;
    if (worst > (uint32_t)cap) return CVTMI_OK;  // a list ran over: the exact path answers this call (and overwrites the output)
    *done = true;
    return CVTMI_OK;
}

// uint8 L2 through the filter pipeline (flat_mfma.hip): exact integer distances on the i8 matrix cores, thresholds from an
// exactly searched leading sample.  *done = false: not applicable / a list ran over, the row-tile kernels answer
static int flat_search_filtered_u8(cvtmi_flat_t h, FlatScratch &S, const uint8_t *q, int64_t nq, int k, float *dist, int64_t *rows, hipStream_t st, bool *done)
{
    *done = false;
    const int D = h->D;
    const int64_t n = h->n;
    if (h->f_pack_n != n) return CVTMI_OK;   // no operand copy (flat_prepare could not build it): the row-tile kernels answer
    // (at least 262 144 rows where the table has twice that: the smallest sample the streaming kernel takes -- through the row-tile
    //  kernels a sample costs ~1 ms whatever its size)
    const int64_t ns = std::max<int64_t>(n >= 2 * 262144 ? 262144 : 65536, (n / 32 + 63) / 64 * 64);
    const int cap = std::min(4096 - k, (48 * k + 1024 + 63) / 64 * 64);
    const uint64_t pair_cap64 = (uint64_t)nq * cap;
    const uint32_t pair_cap = pair_cap64 > 0x7ffffff0ull ? 0x7ffffff0u : (uint32_t)pair_cap64;
    CVTMI_TRY(S.f_stats.reserve(16));
    CVTMI_TRY(S.f_sd.reserve((size_t)nq * k * sizeof(float)));
    CVTMI_TRY(S.f_si.reserve((size_t)nq * k * sizeof(int64_t)));
    CVTMI_TRY(S.f_cnt.reserve((size_t)nq * sizeof(uint32_t)));
    if (S.f_cand.reserve((size_t)pair_cap * sizeof(uint4)) != CVTMI_OK || S.f_seld.reserve((size_t)nq * cap * sizeof(float)) != CVTMI_OK ||
        S.f_seli.reserve((size_t)nq * cap * sizeof(int32_t)) != CVTMI_OK)
        return CVTMI_OK;
    uint32_t *stats = S.f_stats.as<uint32_t>();  // [2] worst list / overflow, [3] pair count
    // one filter stage: the exact top k of rows [0, r0) in (sd, si) -> the exact top k of rows [0, r1) in (od, oi)
    uint32_t worst = 0;
    auto stage = [&](int64_t r0, int64_t r1, const float *sd, const int64_t *si, float *od, int64_t *oi) -> int {
        CVTMI_HIP(hipMemsetAsync(stats + 2, 0, 8, st));
        CVTMI_HIP(hipMemsetAsync(S.f_cnt.p, 0, (size_t)nq * sizeof(uint32_t), st));
        CVTMI_TRY(launch_flat_u8_filter(q, nq, D, h->f_pack.as<uint4>(), h->norms.as<int32_t>(), sd, k, r0, r1, pair_cap, stats + 3,
                                        S.f_cand.as<uint4>(), st));
        CVTMI_TRY(launch_flat_u8_finish(nq, stats + 3, pair_cap, S.f_cand.as<uint4>(), cap, k, sd, si, S.f_cnt.as<uint32_t>(),
                                        S.f_seld.as<float>(), S.f_seli.as<int32_t>(), od, oi, stats + 2, st));
        CVTMI_HIP(hipMemcpyAsync(&worst, stats + 2, 4, hipMemcpyDeviceToHost, st));
        CVTMI_HIP(stream_wait(st));
        return CVTMI_OK;
    };
    // (a two-level sample -- exact kernels on ns / 8 rows, a first filter stage up to ns, as the fp32 path does -- was measured and lost:
    //  the second stage's launches and host sync cost more than the 1.2 ms of exact search they save; nq = 1000: 6.4 -> 7.0 ms)
    // The sample goes through the streaming kernel (128 queries per pass, ~0.12 ms per pass over 312 K rows) while that is cheaper than the
    // row-tile kernels' exact search of it (1.0-2.1 ms whatever the batch: every query block warms its thresholds up from scratch): up to
    // ten passes.  10 M x 512-d, k = 10 (tools/sweep_u8_sample.py, round 5): nq = 256 2.6 -> 1.6 ms, 384 / 512 3.9 -> 3.0, 640 / 768
    // 5.2 -> 4.5, 1000 6.3 -> 5.9-6.0, 1280 7.6 -> 7.4; equal at 1536, slower from 2048 on (16 passes 11.4 against 11.15 ms).
    CVTMI_TRY(flat_search_rows(h, S, ns, q, nq, k, S.f_sd.as<float>(), S.f_si.as<int64_t>(), st, g_flat_u8_sample_passes.load()));
    CVTMI_TRY(stage(ns, n, S.f_sd.as<float>(), S.f_si.as<int64_t>(), dist, rows));
    h->f_last_worst = (long long)worst;
    if (worst > (uint32_t)cap) return CVTMI_OK;
    *done = true;
    return CVTMI_OK;
}

// uint8 L2 as a threshold filter (flat_u8_tfilter.hip: batches, and every search with k > 128).  Queries it could not answer (sample not
// filled, list over, masses of ties at the k-th place; every query of a pass in which a wave's record region ran over) are re-run by the
// row-per-lane kernels under the flags as a predicate, their lists written over the filter's -- nothing on this path waits for the device.
// *done = false: not applicable (no operand copy / no room for the scratch): the round-5 paths answer the call
static int flat_search_bigk_u8(cvtmi_flat_t h, FlatScratch &S, const uint8_t *q, int64_t nq, int k, float *dist, int64_t *rows, hipStream_t st, bool *done)
{
    *done = false;
    const int64_t n = h->n;
    const int D = h->D;
    if (h->f_pack_n != n) return CVTMI_OK;   // no operand copy (flat_prepare could not build it)
    if (S.fs_scratch.reserve(flat_u8_tfilter_scratch(D, n, nq, k)) != CVTMI_OK) { (void)hipGetLastError(); return CVTMI_OK; }
    CVTMI_TRY(S.fs_redo.reserve((size_t)(nq + 1) * sizeof(uint32_t)));
    uint32_t *flags = S.fs_redo.as<uint32_t>();
    CVTMI_TRY(launch_flat_u8_tfilter(D, h->f_pack.p, h->norms.as<int32_t>(), n, q, nq, k, S.fs_scratch.p, dist, rows, flags, st));
    CVTMI_TRY(flat_search_rows(h, S, n, q, nq, k, dist, rows, st, INT64_MAX, flags + 1));
    h->f_last_worst = 0;
    *done = true;
    return CVTMI_OK;
}

// which of the pipelines a search of nq queries takes (the dispatch rules, in one place: flat_prepare builds what they need)
struct FlatRoute { bool stream, tfilter, filt_f32, filt_u8, big_u8; };
// the tuning values a search dispatches on, read ONCE per call: flat_prepare and flat_search_leased must see the same route even if
// another thread calls cvtmi_set_tuning between the two
struct FlatTuning {
    int variant, f32_stream;
    static FlatTuning now() { return { g_flat_variant.load(), g_flat_f32_stream.load() }; }
};
static FlatRoute flat_route(const cvtmi_flat_s *h, const void *q, int64_t nq, int k, const FlatTuning &tun)
{
    const int g_flat_variant = tun.variant, g_flat_f32_stream = tun.f32_stream;  // (this call's snapshot shadows the globals)
    FlatRoute r = { false, false, false, false, false };
    const bool aligned = ((uintptr_t)q & 15) == 0;
    // fp32: one stream over the rows (flat_f32_stream.hip).  flat_variant 2 asks for the older sample + filter pipeline, 1 for the exact kernels
    const bool f32_fast = ((g_flat_variant == 0 && g_flat_f32_stream == 1) || (g_flat_variant != 1 && g_flat_f32_stream == 2)) && aligned &&
                          h->fs_bias.p && h->fs_stats.p;
    r.stream = f32_fast && flat_f32_stream_applies(h->metric, h->D, h->n, k);
    r.tfilter = f32_fast && flat_f32_tfilter_applies(h->metric, h->D, h->n, nq, k);   // batches as a threshold filter (round 6), widths up to 512-d
    r.filt_f32 = g_flat_variant != 1 && aligned && nq <= 65535 &&
                 flat_filter_applies(h->metric, h->D, g_flat_variant == 2 ? std::max<int64_t>(h->n, 131072) : h->n,
                                     g_flat_variant == 2 ? std::max<int64_t>(nq, 16) : nq, k) && h->n >= 2 * 65536;
    // uint8: large batches go through the filter pipeline with the software-pipelined (LDS-DMA) kernel -- measured at 10 M x 512-d:
    // 4096 queries 27.4 -> 21.0 ms, 512 queries 4.6 -> 3.8 ms; smaller batches are one stream over the raw rows (flat_search_rows).
    // flat_variant 2 forces the pipeline wherever it applies, 1 forbids it.
    // (from 256 queries at every width: 10 M x 128-d nq = 256 / 512 / 1000 1.52 / 2.78 / 5.5 ms in streaming passes, 1.03 / 2.08 / 3.25 here;
    //  256-d nq = 256 1.87 against 1.26; between 257 and ~400 queries the two are within 5 %)
    // Round 5 (tools/sweep_u8_dispatch.py, profiles/r05_u8_dispatch_sweep.txt): once the pipeline's sample could go through the streaming
    // kernel on tables of any size (flat_search_filtered_u8: at least 262 144 sample rows) it beats the passes from ~1.3e11 row bytes x
    // queries on, at every width and table size measured (128 / 256 / 512-d, 0.6 .. 10 M rows, k = 10 / 64) -- 10 M x 512-d from 129
    // queries (2.2 -> 1.6 ms), 2 M x 512-d from 129 as well (256 queries: 1.47 ms under the old rule, which took the pipeline with a
    // row-tile sample, 0.49 now), 1 M x 128-d from ~1000; below that the two are within 5-20 % with the passes ahead.
    const bool u8_auto = g_flat_variant == 0 && flat_u8_gfilter_shape(h->D) && nq >= g_flat_u8_filter_min_nq.load() &&
                         h->n >= g_flat_u8_filter_min_rows.load() && k <= 64 &&
                         (double)h->n * (double)h->D * (double)nq >= 1e9 * (double)g_flat_u8_filter_min_work.load();
    r.filt_u8 = (g_flat_variant == 2 || u8_auto) && h->metric == CVTMI_METRIC_L2U8 && aligned && nq <= 65535 * 256 && h->norms.p &&
                flat_u8_filter_applies(h->D, std::max<int64_t>(h->n, 262144), std::max<int64_t>(nq, 256), k) && h->n >= 2 * 65536;
    // k > 128 (round 6, flat_u8_tfilter.hip): the stream and the pipeline above stop at 128 / 64 neighbours, the exact kernels behind them
    // take one query per workgroup (2 M x 512-d, 1000 queries: k = 128 3.9 ms, k = 129 139 ms)
    r.big_u8 = (g_flat_variant == 0 || (g_flat_variant == 2 && k > 128)) && h->metric == CVTMI_METRIC_L2U8 && aligned && h->norms.p && flat_u8_tfilter_applies(h->D, h->n, nq, k);
    return r;
}

// which pipeline a flat search would take under the current tuning values, for inspection and for the CPU tests that pin the
// dispatch rules (include/cvtmi.h)
extern "C" int cvtmi_flat_describe_dispatch(int metric, int D, int64_t n_rows, int64_t nq, int k, int out[4])
{
    if (!out || metric < 0 || metric > 2 || D < 1 || n_rows < 0 || nq < 1 || k < 1) return fail(CVTMI_EINVAL, "cvtmi_flat_describe_dispatch: bad arguments");
    cvtmi_flat_s h;   // nothing of it touches a device; the buffers a route asks about count as present
    static char present[16];
    h.metric = metric; h.D = D; h.n = n_rows;
    h.fs_bias.p = present; h.fs_stats.p = present; h.norms.p = present;
    alignas(16) static const char aligned_q[16] = {};
    const FlatRoute r = flat_route(&h, aligned_q, nq, k, FlatTuning::now());
    h.fs_bias.p = nullptr; h.fs_stats.p = nullptr; h.norms.p = nullptr;
    out[0] = r.tfilter ? 2 : (r.stream ? 1 : 0);
    out[1] = r.filt_f32 ? 1 : 0;
    out[2] = r.big_u8 ? 2 : (r.filt_u8 ? 1 : 0);
    out[3] = (metric == CVTMI_METRIC_L2U8 && !r.filt_u8 && !r.big_u8 && flat_u8_mstream_applies(D, n_rows, std::min<int64_t>(nq, 128), k)) ? 1 : 0;
    return CVTMI_OK;
}

// The lazily built parts of the index a route needs -- the host copy of the row statistics (fp32 stream), the operand copies of
// the filter pipelines -- are built under the EXCLUSIVE lock, once per index state, and the stream is drained before the lock
// is given back.  Called before the search takes its shared lock.
static int flat_prepare(cvtmi_flat_t h, const void *q, int64_t nq, int k, hipStream_t st, const FlatTuning &tun)
{
    for (int attempt = 0; attempt < 2; ++attempt) {
        FlatRoute r;
        bool need_fs, need_f32, need_u8, need_rm;
        int want_nch = 0;
        {
            std::shared_lock<std::shared_timed_mutex> rd(h->rw);
            r = flat_route(h, q, nq, k, tun);
            need_fs = (r.stream || r.tfilter) && h->fs_stats_n != h->n;
            // the threshold filter reads the copy, and so do the stream kernels for small batches on tables of its size
            const bool tf = (r.tfilter || (r.stream && h->D % 16 == 0 && flat_f32_tfilter_nch(h->D) == h->D / 16 && h->n >= flat_f32_tfilter_min_rows())) &&
                            !h->fs_nonfinite;
            want_nch = tf ? flat_f32_tfilter_nch(h->D) : h->D / 16;
            need_f32 = (h->f_pack_n != h->n || h->f_pack_nch != want_nch) &&
                       (tf ? !need_fs : (r.filt_f32 && !(r.stream && !need_fs && !h->fs_nonfinite)));   // (the stream answers: no copy needed)
            need_u8 = (r.filt_u8 || r.big_u8) && h->f_pack_n != h->n;
            need_rm = tf && g_flat_f32_rows_copy.load() != 0 && h->D >= g_flat_f32_rows_copy.load() && h->D % 4 == 0 && h->f_rows_n != h->n &&
                      !h->f_rows_failed;
            if (!need_fs && !need_f32 && !need_u8 && !need_rm) return CVTMI_OK;
        }
        FlatMutation mut(h, st);
        const int64_t n = h->n;
        if (need_fs && h->fs_stats_n != n) {   // once per index state: do the rows hold non-finite values?
            uint32_t stats[2] = { 0, 0 };
            CVTMI_HIP(hipMemcpyAsync(stats, h->fs_stats.p, sizeof stats, hipMemcpyDeviceToHost, st));
            CVTMI_HIP(stream_wait(st));
            h->fs_nonfinite = stats[1] != 0;
            h->fs_stats_n = n;
            continue;   // the route may not need an operand copy after all
        }
        if (need_f32 && (h->f_pack_n != n || h->f_pack_nch != want_nch)) {   // bf16 operand copy of the rows (same bytes as the fp32 rows)
            // rows appended since the copy was made (the reference adds video by video): only those are packed, the buffers grow by halves
            int64_t row0 = (h->f_pack.p && h->f_bias.p && h->f_istats.p && h->f_pack_nch == want_nch && h->f_pack_n > 0 && h->f_pack_n < n) ? h->f_pack_n : 0;
            h->f_pack_n = -1;
            const size_t need_p = flat_pack_bytes(want_nch, n), need_b = (size_t)((n + 31) / 32) * 32 * sizeof(uint32_t);
            if (row0 > 0) {
                const size_t keep_p = flat_pack_bytes(want_nch, row0), keep_b = (size_t)((row0 + 31) / 32) * 32 * sizeof(uint32_t);
                if (need_p > h->f_pack.cap && h->f_pack.grow(std::max(need_p, h->f_pack.cap + h->f_pack.cap / 2), keep_p, st) != CVTMI_OK) { (void)hipGetLastError(); row0 = 0; }
                if (row0 > 0 && need_b > h->f_bias.cap) CVTMI_TRY(h->f_bias.grow(std::max(need_b, h->f_bias.cap + h->f_bias.cap / 2), keep_b, st));
            }
            if (row0 == 0) {
                if (h->f_pack.reserve(need_p) != CVTMI_OK) return CVTMI_OK;   // no room: the exact path answers
                CVTMI_TRY(h->f_bias.reserve(need_b));
                CVTMI_TRY(h->f_istats.reserve(16));
            }
            CVTMI_TRY(launch_flat_pack(h->data.as<float>(), n, h->D, want_nch, h->metric, h->f_pack.as<uint4>(), h->f_bias.as<uint32_t>(),
                                       h->f_istats.as<uint32_t>(), st, row0));
            uint32_t stats[2] = { 0, 0 };
            CVTMI_HIP(hipMemcpyAsync(stats, h->f_istats.p, sizeof stats, hipMemcpyDeviceToHost, st));
            CVTMI_HIP(stream_wait(st));
            h->f_nonfinite = stats[1] != 0 || !(__builtin_bit_cast(float, stats[0]) <= 3.0e38f);
            h->f_pack_n = n;
            h->f_pack_nch = want_nch;
        }
        if (need_rm && h->f_rows_n != n && !need_fs) {   // row-major copy for the exact finish: the rows appended since it was made, the buffer grows by halves
            int64_t row0 = (h->f_rows.p && h->f_rows_n > 0 && h->f_rows_n < n) ? h->f_rows_n : 0;
            h->f_rows_n = -1;
            const size_t need_b = (size_t)n * h->D * sizeof(float);
            bool ok = true;
            if (row0 > 0 && need_b > h->f_rows.cap &&
                h->f_rows.grow(std::max(need_b, h->f_rows.cap + h->f_rows.cap / 2), (size_t)row0 * h->D * sizeof(float), st) != CVTMI_OK) { (void)hipGetLastError(); row0 = 0; }
            if (row0 == 0 && h->f_rows.reserve(need_b) != CVTMI_OK) { (void)hipGetLastError(); ok = false; h->f_rows_failed = true; }   // no room: the finish gathers from the blocked rows
            if (ok) {
                CVTMI_TRY(launch_flat_unblock(h->data.as<float>(), row0, n, h->D, h->f_rows.as<float>(), st));
                CVTMI_HIP(stream_wait(st));
                h->f_rows_n = n;
            }
        }
        if (need_u8 && h->f_pack_n != n) {    // operand-ordered copy of the rows (x - 128 as int8)
            // rows appended since the copy was made: only their tiles are packed (from the last, partly filled one on), the buffer grows by halves
            int64_t row0 = (h->f_pack.p && h->f_pack_n > 0 && h->f_pack_n < n) ? h->f_pack_n / 32 * 32 : 0;
            h->f_pack_n = -1;
            const size_t need_p = flat_u8_pack_bytes(h->D, n);
            if (row0 > 0 && need_p > h->f_pack.cap &&
                h->f_pack.grow(std::max(need_p, h->f_pack.cap + h->f_pack.cap / 2), flat_u8_pack_bytes(h->D, row0), st) != CVTMI_OK) { (void)hipGetLastError(); row0 = 0; }
            if (row0 == 0 && h->f_pack.reserve(need_p) != CVTMI_OK) return CVTMI_OK;
            CVTMI_TRY(launch_flat_u8_pack(h->data.as<uint8_t>(), n, h->D, h->f_pack.as<uint4>(), st, row0));
            CVTMI_HIP(stream_wait(st));
            h->f_pack_n = n;
        }
        return CVTMI_OK;
    }
    return CVTMI_OK;
}

// the search proper, on a leased scratch set, under the shared lock
static int flat_search_leased(cvtmi_flat_t h, FlatScratch &S, const void *q, int64_t nq, int k, void *dist, int64_t *labels, hipStream_t st,
                              const FlatTuning &tun)
{
    bool done = false;
    long long worst0 = 0;
    h->f_last_worst = worst0;
    int how = 0;
    const FlatRoute r = flat_route(h, q, nq, k, tun);
    if (r.stream || r.tfilter) {
        int how_s = 2;
        CVTMI_TRY(flat_search_streamed(h, S, reinterpret_cast<const float *>(q), nq, k, reinterpret_cast<float *>(dist), labels, st, &done, &how_s));
        if (done) how = how_s;
    }
    if (!done && r.filt_f32)
        CVTMI_TRY(flat_search_filtered(h, S, reinterpret_cast<const float *>(q), nq, k, reinterpret_cast<float *>(dist), labels, st, &done));
    if (!done && r.big_u8) {
        CVTMI_TRY(flat_search_bigk_u8(h, S, reinterpret_cast<const uint8_t *>(q), nq, k, reinterpret_cast<float *>(dist), labels, st, &done));
        if (done) how = 4;
    }
    if (!done && r.filt_u8)
        CVTMI_TRY(flat_search_filtered_u8(h, S, reinterpret_cast<const uint8_t *>(q), nq, k, reinterpret_cast<float *>(dist), labels, st, &done));
    h->f_last_filtered = done ? (how ? how : 1) : 0;
    if (!done) CVTMI_TRY(flat_search_rows(h, S, h->n, q, nq, k, reinterpret_cast<float *>(dist), labels, st));
    if (!h->identity) CVTMI_TRY(launch_gather_labels(labels, nq * k, h->labels.as<int64_t>(), st));
    else if (h->id_base != 0) CVTMI_TRY(launch_offset_labels(labels, nq * k, h->id_base, st));
    return CVTMI_OK;
}

int cvtmi_flat_search_dev(cvtmi_flat_t h, const void *q, int64_t nq, int k, void *dist, int64_t *labels, void *stream)
{
    CHECK_H(h);
    if (nq < 0 || (nq > 0 && (!q || !dist || !labels))) return fail(CVTMI_EINVAL, "cvtmi_flat_search: bad arguments");
    if (k < 1 || k > CVTMI_K_MAX) return fail(CVTMI_EUNSUPPORTED, "cvtmi_flat_search: k=%d outside 1..%d", k, CVTMI_K_MAX);
    if (nq == 0) return CVTMI_OK;
    hipStream_t st = (hipStream_t)stream;
    const FlatTuning tun = FlatTuning::now();
    CVTMI_TRY(flat_prepare(h, q, nq, k, st, tun));
    std::shared_lock<std::shared_timed_mutex> rd(h->rw);
    FlatLease lease;
    CVTMI_TRY(lease.open(h, st, false));
    return flat_search_leased(h, *lease.s, q, nq, k, dist, labels, st, tun);
}

int cvtmi_flat_set_id_base(cvtmi_flat_t h, int64_t base)
{
    if (!h) return fail(CVTMI_EINVAL, "cvtmi_flat_set_id_base: null");
    h->id_base = base;
    return CVTMI_OK;
}

// row-sharded exhaustive search (the flat twin of cvtmi_opq_search_sharded_dev): local search into the communicator's slot,
// one all-gather, merge.  uint8 L2: the int32 distances travel and merge as their bit patterns (shard.hip).
int cvtmi_flat_search_sharded_dev(cvtmi_flat_t h, cvtmi_comm_t c, const void *q, int64_t nq, int k, void *dist, int64_t *labels, void *stream)
{
    CVTMI_TRY(comm_validate(c));
    if (!h) return fail(CVTMI_EINVAL, "cvtmi_flat_search_sharded: null handle");
    if (nq < 0 || (nq > 0 && (!dist || !labels))) return fail(CVTMI_EINVAL, "cvtmi_flat_search_sharded: bad arguments");
    if (k < 1 || k > CVTMI_K_MAX) return fail(CVTMI_EUNSUPPORTED, "cvtmi_flat_search_sharded: k=%d outside 1..%d", k, CVTMI_K_MAX);
    if (nq == 0) return CVTMI_OK;
    Serial serial_c(*comm_sync(c), (hipStream_t)stream);
    CHECK_H(h);
    if (comm_world(c) == 1 && !comm_has_transport(c)) return cvtmi_flat_search_dev(h, q, nq, k, dist, labels, stream);
    int rc = comm_device(c) != h->device ? fail(CVTMI_EINVAL, "cvtmi_flat_search_sharded: handle and communicator live on different devices") : CVTMI_OK;
    float *sd = nullptr;
    int64_t *si = nullptr;
    if (rc == CVTMI_OK) rc = !q ? fail(CVTMI_EINVAL, "cvtmi_flat_search_sharded: null queries") : sharded_local_failure(c);
    if (rc == CVTMI_OK) rc = comm_local_slot(c, nq, k, &sd, &si);
    if (rc == CVTMI_OK) rc = cvtmi_flat_search_dev(h, q, nq, k, sd, si, stream);
    return comm_exchange_merge(c, nq, k, rc, reinterpret_cast<float *>(dist), labels, (hipStream_t)stream);
}

int cvtmi_flat_search_sharded(cvtmi_flat_t h, cvtmi_comm_t c, const void *q, int64_t nq, int k, void *dist, int64_t *labels)
{
    CVTMI_TRY(comm_validate(c));
    CHECK_H(h);
    if (nq < 0 || (nq > 0 && (!q || !dist || !labels))) return fail(CVTMI_EINVAL, "cvtmi_flat_search_sharded: bad arguments");
    if (k < 1 || k > CVTMI_K_MAX) return fail(CVTMI_EUNSUPPORTED, "cvtmi_flat_search_sharded: k=%d outside 1..%d", k, CVTMI_K_MAX);
    if (nq == 0) return CVTMI_OK;
    Tmp dq, dd, di;
    CVTMI_TRY(dq.upload(q, (size_t)nq * h->row_bytes));
    CVTMI_TRY(dd.alloc((size_t)nq * k * 4));
    CVTMI_TRY(di.alloc((size_t)nq * k * 8));
    CVTMI_TRY(cvtmi_flat_search_sharded_dev(h, c, dq.p, nq, k, dd.p, di.as<int64_t>(), nullptr));
    CVTMI_HIP(hipMemcpy(dist, dd.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost));
    CVTMI_HIP(hipMemcpy(labels, di.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost));
    {   // (as in cvtmi_opq_search_sharded)
        Serial serial_c(*comm_sync(c), nullptr);
        CVTMI_TRY(comm_take_deferred(c));
    }
    return CVTMI_OK;
}

int cvtmi_flat_search_sharded_all(cvtmi_flat_t *handles, cvtmi_comm_t *comms, int ndev, const void *q, int64_t nq, int k, void *dist,
                                  int64_t *labels)
{
    if (!handles || !comms || ndev < 1) return fail(CVTMI_EINVAL, "cvtmi_flat_search_sharded_all: bad arguments");
    if (nq < 0 || (nq > 0 && (!q || !dist || !labels))) return fail(CVTMI_EINVAL, "cvtmi_flat_search_sharded_all: bad arguments");
    if (k < 1 || k > CVTMI_K_MAX) return fail(CVTMI_EUNSUPPORTED, "cvtmi_flat_search_sharded_all: k=%d outside 1..%d", k, CVTMI_K_MAX);
    std::vector<int> devices(ndev);
    for (int d = 0; d < ndev; ++d) {
        CVTMI_TRY(comm_validate(comms[d]));
        if (!handles[d]) return fail(CVTMI_EINVAL, "cvtmi_flat_search_sharded_all: null handle %d", d);
        if (comm_world(comms[d]) != ndev || comm_rank(comms[d]) != d || comm_device(comms[d]) != handles[d]->device)
            return fail(CVTMI_EINVAL, "cvtmi_flat_search_sharded_all: communicator %d does not belong to handle %d", d, d);
        devices[d] = handles[d]->device;
    }
    if (nq == 0) return CVTMI_OK;
    return sharded_all(comms, ndev, q, (size_t)nq * handles[0]->row_bytes, nq, k, dist, labels, devices.data(),
                       [&](int d, const void *qd, float *sd, int64_t *si) { return cvtmi_flat_search_dev(handles[d], qd, nq, k, sd, si, nullptr); });
}

int cvtmi_flat_last_search(cvtmi_flat_t h, int *filtered, int64_t *max_candidates)
{
    if (!h) return fail(CVTMI_EINVAL, "cvtmi_flat_last_search: null handle");
    if (filtered) *filtered = h->f_last_filtered;
    if (max_candidates) *max_candidates = h->f_last_worst;
    return CVTMI_OK;
}

// host pointers in and out.  Every call runs on the stream of its own scratch set (staging buffers included), so callers on
// several threads -- the reference's searchKnn is a pure read, brutoforce.hpp:73-93 -- proceed side by side.
int cvtmi_flat_search(cvtmi_flat_t h, const void *q, int64_t nq, int k, void *dist, int64_t *labels)
{
    CHECK_H(h);
    if (nq < 0 || (nq > 0 && (!q || !dist || !labels))) return fail(CVTMI_EINVAL, "cvtmi_flat_search: bad arguments");
    if (k < 1 || k > CVTMI_K_MAX) return fail(CVTMI_EUNSUPPORTED, "cvtmi_flat_search: k=%d outside 1..%d", k, CVTMI_K_MAX);
    if (nq == 0) return CVTMI_OK;
    alignas(16) static const char aligned_probe[16] = {};
    const FlatTuning tun = FlatTuning::now();
    CVTMI_TRY(flat_prepare(h, aligned_probe, nq, k, nullptr, tun));   // (the staged queries are 16-byte aligned)
    std::shared_lock<std::shared_timed_mutex> rd(h->rw);
    FlatLease lease;
    CVTMI_TRY(lease.open(h, nullptr, true));
    FlatScratch &S = *lease.s;
    hipStream_t st = lease.st;
    CVTMI_TRY(S.io_q.reserve((size_t)nq * h->row_bytes));
    // Small calls (the brute_force CLI's shape, one searchKnn per query: brute_force_search/src/brute_force.cpp:86): the three copies
    // are a tenth of such a call.  The queries go up from a page-locked staging area (a truly asynchronous copy), and the last kernel of
    // the search writes the lists straight into that area (device-visible host memory) -- no copy engine on the way back.
    const size_t qn = (size_t)nq * h->row_bytes, dn = (size_t)nq * k * 4, in = (size_t)nq * k * 8;
    if (g_flat_small_zero_copy.load() && qn <= ((size_t)64 << 10) && dn + in <= ((size_t)768 << 10)) {
        const size_t qoff = (qn + 255) & ~(size_t)255, doff = (dn + 255) & ~(size_t)255;
        CVTMI_TRY(S.io_pin.reserve(std::max(qoff + doff + in, (size_t)1 << 20)));
        void *pin_dev = nullptr;
        if (hipHostGetDevicePointer(&pin_dev, S.io_pin.p, 0) == hipSuccess && pin_dev) {
            char *pin = S.io_pin.as<char>(), *pd = static_cast<char *>(pin_dev);
            memcpy(pin, q, qn);
            CVTMI_HIP(hipMemcpyAsync(S.io_q.p, pin, qn, hipMemcpyHostToDevice, st));
            CVTMI_TRY(flat_search_leased(h, S, S.io_q.p, nq, k, pd + qoff, reinterpret_cast<int64_t *>(pd + qoff + doff), st, tun));
            CVTMI_HIP(stream_wait(st));
            memcpy(dist, pin + qoff, dn);
            memcpy(labels, pin + qoff + doff, in);
            return CVTMI_OK;
        }
        (void)hipGetLastError();
    }
    CVTMI_TRY(S.io_d.reserve((size_t)nq * k * 4));
    CVTMI_TRY(S.io_i.reserve((size_t)nq * k * 8));
    CVTMI_HIP(hipMemcpyAsync(S.io_q.p, q, (size_t)nq * h->row_bytes, hipMemcpyHostToDevice, st));
    CVTMI_TRY(flat_search_leased(h, S, S.io_q.p, nq, k, S.io_d.p, S.io_i.as<int64_t>(), st, tun));
    CVTMI_HIP(hipMemcpyAsync(dist, S.io_d.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, st));
    CVTMI_HIP(hipMemcpyAsync(labels, S.io_i.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost, st));
    CVTMI_HIP(stream_wait(st));
    return CVTMI_OK;
}

// ================================================================ SQ8 =========================
int cvtmi_sq8_train_dev(const float *x, int64_t n, int d, int l2norm, float *vmin, float *vdiff, void *stream)
{
    if (n < 0 || d < 1 || !vmin || !vdiff || (n > 0 && !x)) return fail(CVTMI_EINVAL, "cvtmi_sq8_train: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    Tmp den, keys;
    if (l2norm && n > 0 && !sq8_single_pass(d, x, nullptr, nullptr, nullptr, n)) CVTMI_TRY(den.alloc((size_t)n * sizeof(float)));
    CVTMI_TRY(keys.alloc((size_t)d * 2 * sizeof(uint32_t)));
    CVTMI_TRY(launch_sq8_train(x, n, d, l2norm, den.as<float>(), keys.as<uint32_t>(), keys.as<uint32_t>() + d, vmin, vdiff,
                               st));
    CVTMI_HIP(stream_wait(st));  // the temporaries die with this frame
    return CVTMI_OK;
}

int cvtmi_sq8_train(const float *x, int64_t n, int d, int l2norm, float *vmin, float *vdiff)
{
    if (n < 0 || d < 1 || !vmin || !vdiff || (n > 0 && !x)) return fail(CVTMI_EINVAL, "cvtmi_sq8_train: bad arguments");
    Tmp dx, dmin, ddiff;
    CVTMI_TRY(dx.upload(x, (size_t)n * d * sizeof(float)));
    CVTMI_TRY(dmin.alloc((size_t)d * sizeof(float)));
    CVTMI_TRY(ddiff.alloc((size_t)d * sizeof(float)));
    CVTMI_TRY(cvtmi_sq8_train_dev(dx.as<float>(), n, d, l2norm, dmin.as<float>(), ddiff.as<float>(), nullptr));
    CVTMI_HIP(hipMemcpy(vmin, dmin.p, (size_t)d * sizeof(float), hipMemcpyDeviceToHost));
    CVTMI_HIP(hipMemcpy(vdiff, ddiff.p, (size_t)d * sizeof(float), hipMemcpyDeviceToHost));
    return CVTMI_OK;
}

int cvtmi_sq8_encode_dev(const float *vmin, const float *vdiff, int d, float *x, int64_t n, int l2norm, uint8_t *codes,
                         void *stream)
{
    if (n < 0 || d < 1 || !vmin || !vdiff || (n > 0 && (!x || !codes))) return fail(CVTMI_EINVAL, "cvtmi_sq8_encode: bad arguments");
    if (n == 0) return CVTMI_OK;
    hipStream_t st = (hipStream_t)stream;
    Tmp den;
    const bool two_pass = l2norm && !sq8_single_pass(d, x, codes, vmin, vdiff, n);
    if (two_pass) CVTMI_TRY(den.alloc((size_t)n * sizeof(float)));
    CVTMI_TRY(launch_sq8_encode_rows(vmin, vdiff, d, x, n, l2norm ? 1 : 0, l2norm == 2 ? 0 : 1, codes, den.as<float>(), st));
    if (two_pass) CVTMI_HIP(stream_wait(st));  // the temporary dies with this frame
    return CVTMI_OK;
}

}  // extern "C"

// Small SQ8 calls through the host-pointer entries -- the reference encodes and decodes ONE feature vector per call (int8_quan.cc:72-132) --
// used to pay four device allocations, four copies and four frees per call (65-80 us, 270 at 2048-d).  They now run out of a page-locked
// scratch area: the model, the rows and the results live in device-visible host memory, the kernels read and write it directly
// (everything is touched once), and nothing is allocated per call.  The areas are kept per device for the life of the process (a handful
// of 1 MB buffers; the SQ8 entries have no handle that could own them).
namespace {
struct Sq8HostScratch {
    PinBuf pin;
    hipStream_t st = nullptr;
    int device = -1;
    bool busy = false;
};
std::mutex g_sq8_host_mu;
std::vector<Sq8HostScratch *> g_sq8_host_pool;   // never shrinks, never freed (process lifetime)
constexpr size_t SQ8_HOST_SMALL = (size_t)1 << 20;
struct Sq8HostLease {
    Sq8HostScratch *s = nullptr;
    int open()
    {
        int dev = 0;
        CVTMI_HIP(hipGetDevice(&dev));
        {
            std::lock_guard<std::mutex> g(g_sq8_host_mu);
            for (Sq8HostScratch *c : g_sq8_host_pool)
                if (!c->busy && c->device == dev) { s = c; break; }
            if (!s) {
                s = new (std::nothrow) Sq8HostScratch();
                if (!s) return fail(CVTMI_ENOMEM, "sq8: out of host memory");
                s->device = dev;
                g_sq8_host_pool.push_back(s);
            }
            s->busy = true;
        }
        if (!s->st) CVTMI_HIP(hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking));
        return s->pin.reserve(SQ8_HOST_SMALL + 4096);
    }
    ~Sq8HostLease()
    {
        if (!s) return;
        std::lock_guard<std::mutex> g(g_sq8_host_mu);
        s->busy = false;
    }
};
inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
}  // namespace

extern "C" {

int cvtmi_sq8_encode(const float *vmin, const float *vdiff, int d, float *x, int64_t n, int l2norm, uint8_t *codes)
{
    if (n < 0 || d < 1 || !vmin || !vdiff || (n > 0 && (!x || !codes))) return fail(CVTMI_EINVAL, "cvtmi_sq8_encode: bad arguments");
    if (n == 0) return CVTMI_OK;
    {
        const size_t mb = up256((size_t)d * sizeof(float)), xb = up256((size_t)n * d * sizeof(float)), cb = up256((size_t)n * d);
        const size_t nb = up256((size_t)n * sizeof(float));   // row norms of the widths that take two passes
        if (g_sq8_host_small.load() && 2 * mb + xb + cb + nb <= SQ8_HOST_SMALL) {
            Sq8HostLease lease;
            CVTMI_TRY(lease.open());
            void *pd_ = nullptr;
            if (hipHostGetDevicePointer(&pd_, lease.s->pin.p, 0) == hipSuccess && pd_) {
                char *pin = lease.s->pin.as<char>(), *pd = static_cast<char *>(pd_);
                memcpy(pin, vmin, (size_t)d * sizeof(float));
                memcpy(pin + mb, vdiff, (size_t)d * sizeof(float));
                memcpy(pin + 2 * mb, x, (size_t)n * d * sizeof(float));
                CVTMI_TRY(launch_sq8_encode_rows(reinterpret_cast<float *>(pd), reinterpret_cast<float *>(pd + mb), d, reinterpret_cast<float *>(pd + 2 * mb), n,
                                                 l2norm ? 1 : 0, l2norm == 2 ? 0 : 1, reinterpret_cast<uint8_t *>(pd + 2 * mb + xb),
                                                 reinterpret_cast<float *>(pd + 2 * mb + xb + cb), lease.s->st));
                CVTMI_HIP(stream_wait(lease.s->st));
                memcpy(codes, pin + 2 * mb + xb, (size_t)n * d);
                if (l2norm == 1) memcpy(x, pin + 2 * mb, (size_t)n * d * sizeof(float));
                return CVTMI_OK;
            }
            (void)hipGetLastError();
        }
    }
    Tmp dmin, ddiff, dx, dc;
    CVTMI_TRY(dmin.upload(vmin, (size_t)d * sizeof(float)));
    CVTMI_TRY(ddiff.upload(vdiff, (size_t)d * sizeof(float)));
    CVTMI_TRY(dx.upload(x, (size_t)n * d * sizeof(float)));
    CVTMI_TRY(dc.alloc((size_t)n * d));
    CVTMI_TRY(cvtmi_sq8_encode_dev(dmin.as<float>(), ddiff.as<float>(), d, dx.as<float>(), n, l2norm, dc.as<uint8_t>(), nullptr));
    CVTMI_HIP(hipMemcpy(codes, dc.p, (size_t)n * d, hipMemcpyDeviceToHost));
    if (l2norm == 1) CVTMI_HIP(hipMemcpy(x, dx.p, (size_t)n * d * sizeof(float), hipMemcpyDeviceToHost));
    return CVTMI_OK;
}

// ================================================================ PCA =========================
int cvtmi_pca_project_dev(const float *mean, const float *vectors, int din, int dout, const float *x, int64_t n, int l2norm,
                          float *y, void *stream)
{
    if (n < 0 || !mean || !vectors || (n > 0 && (!x || !y))) return fail(CVTMI_EINVAL, "cvtmi_pca_project: bad arguments");
    return launch_pca_project(mean, vectors, din, dout, x, n, l2norm, y, (hipStream_t)stream);
}

int cvtmi_pca_project(const float *mean, const float *vectors, int din, int dout, const float *x, int64_t n, int l2norm, float *y)
{
    if (n < 0 || din < 1 || dout < 1 || !mean || !vectors || (n > 0 && (!x || !y))) return fail(CVTMI_EINVAL, "cvtmi_pca_project: bad arguments");
    if (n == 0) return CVTMI_OK;
    Tmp dm, de, dx, dy;
    CVTMI_TRY(dm.upload(mean, (size_t)din * sizeof(float)));
    CVTMI_TRY(de.upload(vectors, (size_t)dout * din * sizeof(float)));
    CVTMI_TRY(dx.upload(x, (size_t)n * din * sizeof(float)));
    CVTMI_TRY(dy.alloc((size_t)n * dout * sizeof(float)));
    CVTMI_TRY(cvtmi_pca_project_dev(dm.as<float>(), de.as<float>(), din, dout, dx.as<float>(), n, l2norm, dy.as<float>(), nullptr));
    CVTMI_HIP(hipMemcpy(y, dy.p, (size_t)n * dout * sizeof(float), hipMemcpyDeviceToHost));
    return CVTMI_OK;
}

static int sq8_decode_dev_mode(const float *vmin, const float *vdiff, int d, const uint8_t *codes, int64_t n, float *x, void *stream, int mode)
{
    if (n < 0 || d < 1 || !vmin || !vdiff || (n > 0 && (!x || !codes))) return fail(CVTMI_EINVAL, "cvtmi_sq8_decode: bad arguments");
    return launch_sq8_decode(vmin, vdiff, d, codes, n, x, (hipStream_t)stream, mode);
}
static int sq8_decode_host_mode(const float *vmin, const float *vdiff, int d, const uint8_t *codes, int64_t n, float *x, int mode)
{
    if (n < 0 || d < 1 || !vmin || !vdiff || (n > 0 && (!x || !codes))) return fail(CVTMI_EINVAL, "cvtmi_sq8_decode: bad arguments");
    if (n == 0) return CVTMI_OK;
    {   // small calls: out of the page-locked scratch area (see Sq8HostScratch)
        const size_t mb = up256((size_t)d * sizeof(float)), xb = up256((size_t)n * d * sizeof(float)), cb = up256((size_t)n * d);
        if (g_sq8_host_small.load() && 2 * mb + xb + cb <= SQ8_HOST_SMALL) {
            Sq8HostLease lease;
            CVTMI_TRY(lease.open());
            void *pd_ = nullptr;
            if (hipHostGetDevicePointer(&pd_, lease.s->pin.p, 0) == hipSuccess && pd_) {
                char *pin = lease.s->pin.as<char>(), *pd = static_cast<char *>(pd_);
                memcpy(pin, vmin, (size_t)d * sizeof(float));
                memcpy(pin + mb, vdiff, (size_t)d * sizeof(float));
                memcpy(pin + 2 * mb, codes, (size_t)n * d);
                CVTMI_TRY(sq8_decode_dev_mode(reinterpret_cast<float *>(pd), reinterpret_cast<float *>(pd + mb), d, reinterpret_cast<uint8_t *>(pd + 2 * mb), n,
                                              reinterpret_cast<float *>(pd + 2 * mb + cb), lease.s->st, mode));
                CVTMI_HIP(stream_wait(lease.s->st));
                memcpy(x, pin + 2 * mb + cb, (size_t)n * d * sizeof(float));
                return CVTMI_OK;
            }
            (void)hipGetLastError();
        }
    }
    Tmp dmin, ddiff, dx, dc;
    CVTMI_TRY(dmin.upload(vmin, (size_t)d * sizeof(float)));
    CVTMI_TRY(ddiff.upload(vdiff, (size_t)d * sizeof(float)));
    CVTMI_TRY(dc.upload(codes, (size_t)n * d));
    CVTMI_TRY(dx.alloc((size_t)n * d * sizeof(float)));
    CVTMI_TRY(sq8_decode_dev_mode(dmin.as<float>(), ddiff.as<float>(), d, dc.as<uint8_t>(), n, dx.as<float>(), nullptr, mode));
    CVTMI_HIP(hipMemcpy(x, dx.p, (size_t)n * d * sizeof(float), hipMemcpyDeviceToHost));
    return CVTMI_OK;
}

int cvtmi_sq8_decode_dev(const float *vmin, const float *vdiff, int d, const uint8_t *codes, int64_t n, float *x, void *stream)
{
    return sq8_decode_dev_mode(vmin, vdiff, d, codes, n, x, stream, 0);
}
int cvtmi_sq8_decode(const float *vmin, const float *vdiff, int d, const uint8_t *codes, int64_t n, float *x)
{
    return sq8_decode_host_mode(vmin, vdiff, d, codes, n, x, 0);
}
int cvtmi_sq8_decode_faiss_dev(const float *vmin, const float *vdiff, int d, const uint8_t *codes, int64_t n, float *x, void *stream)
{
    return sq8_decode_dev_mode(vmin, vdiff, d, codes, n, x, stream, 1);
}
int cvtmi_sq8_decode_faiss(const float *vmin, const float *vdiff, int d, const uint8_t *codes, int64_t n, float *x)
{
    return sq8_decode_host_mode(vmin, vdiff, d, codes, n, x, 1);
}

}  // extern "C"

// ================================================================ codebook training ===========
static uint64_t splitmix64(uint64_t &s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int cvtmi_kmeans_dev(const float *x, int64_t ld, int64_t n, int d, int k, int niter, uint64_t seed, float *centroids,
                     int32_t *assign, int *iters_done, void *stream)
{
    if (!x || !centroids || n < 1 || d < 1 || k < 1 || ld < d) return fail(CVTMI_EINVAL, "cvtmi_kmeans: bad arguments");
    if (n < k) return fail(CVTMI_EINVAL, "cvtmi_kmeans: fewer rows (%lld) than centroids (%d)", (long long)n, k);
    if (d > 512) return fail(CVTMI_EUNSUPPORTED, "cvtmi_kmeans: d=%d > 512", d);
    hipStream_t st = (hipStream_t)stream;
    // seeding: k distinct rows, index = splitmix64() % n, redraw on repeats (host side, k values)
    std::vector<int64_t> rows((size_t)k);
    {
        std::vector<uint8_t> taken((size_t)n, 0);
        uint64_t s = seed;
        for (int c = 0; c < k; ++c) {
            int64_t r;
            do { r = (int64_t)(splitmix64(s) % (uint64_t)n); } while (taken[(size_t)r]);
            taken[(size_t)r] = 1;
            rows[(size_t)c] = r;
        }
    }
    Tmp drows, dassign, dchanged;
    CVTMI_TRY(drows.upload(rows.data(), (size_t)k * sizeof(int64_t)));
    CVTMI_TRY(launch_kmeans_gather(x, ld, d, drows.as<int64_t>(), k, centroids, st));
    int32_t *as = assign;
    if (!as) {
        CVTMI_TRY(dassign.alloc((size_t)n * sizeof(int32_t)));
        as = dassign.as<int32_t>();
    }
    CVTMI_TRY(launch_kmeans_fill(as, n, -2, st));
    CVTMI_TRY(dchanged.alloc(sizeof(unsigned long long)));
    const int max_iter = niter > 0 ? niter : 100;
    int it = 0;
    for (;;) {
        CVTMI_HIP(hipMemsetAsync(dchanged.p, 0, sizeof(unsigned long long), st));
        CVTMI_TRY(launch_kmeans_assign(x, ld, n, d, centroids, k, as, dchanged.as<unsigned long long>(), st));
        unsigned long long changed = 0;
        CVTMI_HIP(hipMemcpyAsync(&changed, dchanged.p, sizeof changed, hipMemcpyDeviceToHost, st));
        CVTMI_HIP(stream_wait(st));
        if (changed == 0 || it >= max_iter) break;
        CVTMI_TRY(launch_kmeans_update(x, ld, n, d, as, k, centroids, st));
        ++it;
    }
    if (iters_done) *iters_done = it;
    CVTMI_HIP(stream_wait(st));  // the temporaries die with this frame
    return CVTMI_OK;
}

int cvtmi_kmeans(const float *x, int64_t n, int d, int k, int niter, uint64_t seed, float *centroids, int32_t *assign,
                 int *iters_done)
{
    if (!x || !centroids || n < 1 || d < 1 || k < 1) return fail(CVTMI_EINVAL, "cvtmi_kmeans: bad arguments");
    Tmp dx, dc, da;
    CVTMI_TRY(dx.upload(x, (size_t)n * d * sizeof(float)));
    CVTMI_TRY(dc.alloc((size_t)k * d * sizeof(float)));
    CVTMI_TRY(da.alloc((size_t)n * sizeof(int32_t)));
    CVTMI_TRY(cvtmi_kmeans_dev(dx.as<float>(), d, n, d, k, niter, seed, dc.as<float>(), da.as<int32_t>(), iters_done, nullptr));
    CVTMI_HIP(hipMemcpy(centroids, dc.p, (size_t)k * d * sizeof(float), hipMemcpyDeviceToHost));
    if (assign) CVTMI_HIP(hipMemcpy(assign, da.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return CVTMI_OK;
}

int cvtmi_opq_train_dev(const float *x, int64_t n, int D, int coarseK, int M, int K, int niter, uint64_t seed, float *coarse,
                        float *books, void *stream)
{
    if (!x || !coarse || !books || n < 1 || D < 1 || M < 1 || M > 16 || D % M != 0 || K < 1 || K > 256 || coarseK < 1)
        return fail(CVTMI_EINVAL, "cvtmi_opq_train: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int step = D / M;
    Tmp assign, res;
    CVTMI_TRY(assign.alloc((size_t)n * sizeof(int32_t)));
    CVTMI_TRY(res.alloc((size_t)n * D * sizeof(float)));
    CVTMI_TRY(cvtmi_kmeans_dev(x, D, n, D, coarseK, niter, seed, coarse, assign.as<int32_t>(), nullptr, stream));
    CVTMI_TRY(launch_kmeans_residual(x, n, D, coarse, assign.as<int32_t>(), res.as<float>(), st));
    for (int m = 0; m < M; ++m)
        CVTMI_TRY(cvtmi_kmeans_dev(res.as<float>() + m * step, D, n, step, K, niter, seed, books + (size_t)m * K * step,
                                   assign.as<int32_t>(), nullptr, stream));
    CVTMI_HIP(stream_wait(st));
    return CVTMI_OK;
}

int cvtmi_opq_train(const float *x, int64_t n, int D, int coarseK, int M, int K, int niter, uint64_t seed, float *coarse,
                    float *books)
{
    if (!x || !coarse || !books || n < 1 || D < 1) return fail(CVTMI_EINVAL, "cvtmi_opq_train: bad arguments");
    Tmp dx, dc, db;
    CVTMI_TRY(dx.upload(x, (size_t)n * D * sizeof(float)));
    CVTMI_TRY(dc.alloc((size_t)coarseK * D * sizeof(float)));
    CVTMI_TRY(db.alloc((size_t)K * D * sizeof(float)));
    CVTMI_TRY(cvtmi_opq_train_dev(dx.as<float>(), n, D, coarseK, M, K, niter, seed, dc.as<float>(), db.as<float>(), nullptr));
    CVTMI_HIP(hipMemcpy(coarse, dc.p, (size_t)coarseK * D * sizeof(float), hipMemcpyDeviceToHost));
    CVTMI_HIP(hipMemcpy(books, db.p, (size_t)K * D * sizeof(float), hipMemcpyDeviceToHost));
    return CVTMI_OK;
}

// ================================================================ HNSW search ==================
// The graph is immutable once loaded and searchKnn is a pure read in the reference (hnswalg.h:688-728): searches on one handle run side
// by side, each on a leased scratch set (visited bits, spilled queues, re-rank lists, host staging) and the stream of its caller (the
// host-pointer entries: the set's own stream).
struct HnswScratch {
    DevBuf s_vis, s_cand, s_err, s_rr_d, s_rr_id, io_q, io_d, io_l;
    hipStream_t own = nullptr;
    hipEvent_t done = nullptr;
    hipStream_t last = nullptr;
    bool pending = false, busy = false;
    void release_all()
    {
        for (DevBuf *b : { &s_vis, &s_cand, &s_err, &s_rr_d, &s_rr_id, &io_q, &io_d, &io_l }) b->release();
        if (own) (void)hipStreamDestroy(own);
        if (done) (void)hipEventDestroy(done);
        own = nullptr; done = nullptr;
    }
};
struct cvtmi_hnsw_s {
    uint32_t magic = 0x484e5357u;
    int device = 0, metric = 0, D = 0;
    HnswDevGraph g{};
    DevBuf vec, links0, labels, upper_off, upper;
    std::mutex pool_mu;
    std::vector<HnswScratch *> pool;
    int slots_per_cu_max = 32, cus = 256;
};
struct HnswLease {
    cvtmi_hnsw_s *h = nullptr;
    HnswScratch *s = nullptr;
    hipStream_t st = nullptr;
    int open(cvtmi_hnsw_s *handle, hipStream_t stream, bool host)
    {
        h = handle; st = stream;
        {
            std::lock_guard<std::mutex> g(h->pool_mu);
            HnswScratch *any = nullptr;
            for (HnswScratch *c : h->pool) {
                if (c->busy) continue;
                if (!host && c->pending && c->last == stream) { s = c; break; }   // same stream as before: nothing to wait for
                if (!any) any = c;
            }
            if (!s) s = any;
            if (!s) {
                s = new (std::nothrow) HnswScratch();
                if (!s) return fail(CVTMI_ENOMEM, "hnsw search: out of host memory");
                h->pool.push_back(s);
            }
            s->busy = true;
        }
        if (host) {
            if (!s->own && hipStreamCreateWithFlags(&s->own, hipStreamNonBlocking) != hipSuccess) { s->busy = false; s = nullptr; return fail(CVTMI_EHIP, "hipStreamCreate failed"); }
            st = s->own;
        }
        if (s->pending && s->last != st) (void)hipStreamWaitEvent(st, s->done, 0);
        return CVTMI_OK;
    }
    ~HnswLease()
    {
        if (!s) return;
        if (!s->done) (void)hipEventCreateWithFlags(&s->done, hipEventDisableTiming);
        if (s->done && hipEventRecord(s->done, st) == hipSuccess) { s->last = st; s->pending = true; }
        std::lock_guard<std::mutex> g(h->pool_mu);
        s->busy = false;
    }
    HnswLease() = default;
    HnswLease(const HnswLease &) = delete;
    HnswLease &operator=(const HnswLease &) = delete;
};
#define CHECK_HN(h) do { if (!(h) || (h)->magic != 0x484e5357u) return fail(CVTMI_EINVAL, "bad hnsw handle"); CVTMI_TRY(use_device((h)->device)); } while (0)

int cvtmi_hnsw_load(const void *file, int64_t bytes, int metric, int D, cvtmi_hnsw_t *out)
{
    if (!file || !out || D < 1 || (metric != CVTMI_METRIC_IP && metric != CVTMI_METRIC_L2F))
        return fail(CVTMI_EINVAL, "cvtmi_hnsw_load: bad arguments (metric must be IP or L2F)");
    if (bytes < 96) return fail(CVTMI_EINVAL, "cvtmi_hnsw_load: not a saveIndex file (too short)");
    const uint8_t *f = static_cast<const uint8_t *>(file), *p = f;
    uint64_t offsetLevel0, max_elements, cur_count, size_per, label_off, offsetData, maxM, maxM0, M, efc;
    int32_t maxlevel; uint32_t enterpoint; double mult;
    auto rd = [&](void *dst, size_t nb) { memcpy(dst, p, nb); p += nb; };
    rd(&offsetLevel0, 8); rd(&max_elements, 8); rd(&cur_count, 8); rd(&size_per, 8); rd(&label_off, 8); rd(&offsetData, 8);
    rd(&maxlevel, 4); rd(&enterpoint, 4); rd(&maxM, 8); rd(&maxM0, 8); rd(&M, 8); rd(&mult, 8); rd(&efc, 8);
    (void)M; (void)mult; (void)efc;
    if (size_per != 4 + 4 * maxM0 + 4 * (uint64_t)D + 8 || offsetData != 4 + 4 * maxM0 || label_off != offsetData + 4 * (uint64_t)D ||
        offsetLevel0 != 0 || cur_count > max_elements || maxM0 > 4096 || maxM > 4096)
        return fail(CVTMI_EINVAL, "cvtmi_hnsw_load: header does not describe %d-d fp32 vectors (size_data_per_element=%llu)", D,
                    (unsigned long long)size_per);
    // the header is untrusted: the product below must not wrap, and the counts size host allocations
    if (max_elements > ((uint64_t)bytes - 96) / size_per) return fail(CVTMI_EINVAL, "cvtmi_hnsw_load: truncated level-0 block");
    if (cur_count > 0 && maxlevel < 0) return fail(CVTMI_EINVAL, "cvtmi_hnsw_load: negative maxlevel");
    const int64_t n = (int64_t)cur_count;
    const uint8_t *l0 = p;
    p += max_elements * size_per;
    std::vector<float> vec;
    std::vector<uint32_t> links0, upper;
    std::vector<int64_t> labels, uoff;
    std::vector<int32_t> levels;  // upper levels a node has link blocks for
    const uint64_t links_per = 4 * maxM + 4;
    try {
        vec.resize((size_t)n * D);
        links0.resize((size_t)n * (maxM0 + 1));
        labels.resize((size_t)n);
        uoff.assign((size_t)n, -1);
        levels.assign((size_t)n, 0);
        for (int64_t i = 0; i < n; ++i) {
            const uint8_t *e = l0 + (uint64_t)i * size_per;
            memcpy(&links0[(size_t)i * (maxM0 + 1)], e, 4 * (maxM0 + 1));
            if (links0[(size_t)i * (maxM0 + 1)] > maxM0) return fail(CVTMI_EINVAL, "cvtmi_hnsw_load: corrupt link count");
            memcpy(&vec[(size_t)i * D], e + offsetData, 4 * (size_t)D);
            uint64_t lab; memcpy(&lab, e + label_off, 8);
            labels[(size_t)i] = (int64_t)lab;
        }
        for (uint64_t i = 0; i < max_elements; ++i) {
            if ((uint64_t)(f + bytes - p) < 4) return fail(CVTMI_EINVAL, "cvtmi_hnsw_load: truncated link lists");
            uint32_t sz; memcpy(&sz, p, 4); p += 4;
            if (sz) {
                if ((uint64_t)(f + bytes - p) < sz || sz % links_per != 0) return fail(CVTMI_EINVAL, "cvtmi_hnsw_load: corrupt link list");
                if ((int64_t)i < n) {
                    uoff[(size_t)i] = (int64_t)upper.size();
                    levels[(size_t)i] = (int32_t)(sz / links_per);
                    upper.resize(upper.size() + sz / 4);
                    memcpy(&upper[(size_t)uoff[(size_t)i]], p, sz);
                }
                p += sz;
            }
        }
    } catch (const std::exception &) {
        return fail(CVTMI_ENOMEM, "cvtmi_hnsw_load: out of host memory for %llu elements", (unsigned long long)cur_count);
    }
    // every link must point inside the graph, and a link at level L at a node that HAS a level-L block: the kernel
    // follows them without further checks (hnsw.hip: a.upper + a.upper_off[cur] + (level - 1) * (maxM + 1))
    for (int64_t i = 0; i < n; ++i) {
        const uint32_t *l = &links0[(size_t)i * (maxM0 + 1)];
        for (uint32_t j = 1; j <= l[0]; ++j) if (l[j] >= (uint64_t)n) return fail(CVTMI_EINVAL, "cvtmi_hnsw_load: link out of range");
        for (int32_t lv = 1; lv <= levels[(size_t)i]; ++lv) {
            const uint32_t *u = &upper[(size_t)uoff[(size_t)i] + (size_t)(lv - 1) * (maxM + 1)];
            if (u[0] > maxM) return fail(CVTMI_EINVAL, "cvtmi_hnsw_load: corrupt upper link count");
            for (uint32_t j = 1; j <= u[0]; ++j) {
                if (u[j] >= (uint64_t)n) return fail(CVTMI_EINVAL, "cvtmi_hnsw_load: link out of range");
                if (levels[u[j]] < lv) return fail(CVTMI_EINVAL, "cvtmi_hnsw_load: level-%d link to a node without that level", lv);
            }
        }
    }
    if (n > 0 && (enterpoint >= (uint64_t)n || levels[enterpoint] < maxlevel))
        return fail(CVTMI_EINVAL, "cvtmi_hnsw_load: bad entry point");
    int dev = 0;
    CVTMI_HIP(hipGetDevice(&dev));  // no device: fails here, there is no CPU path
    cvtmi_hnsw_s *h = new (std::nothrow) cvtmi_hnsw_s();
    if (!h) return fail(CVTMI_ENOMEM, "cvtmi_hnsw_load: out of host memory");
    h->device = dev; h->metric = metric; h->D = D;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) h->cus = prop.multiProcessorCount;
    }
    auto up = [&](DevBuf &b, const void *src, size_t nb) -> int {
        CVTMI_TRY(b.reserve(nb ? nb : 16));
        if (nb) CVTMI_HIP(hipMemcpy(b.p, src, nb, hipMemcpyHostToDevice));
        return CVTMI_OK;
    };
    int rc = up(h->vec, vec.data(), vec.size() * 4);
    if (rc == CVTMI_OK) rc = up(h->links0, links0.data(), links0.size() * 4);
    if (rc == CVTMI_OK) rc = up(h->labels, labels.data(), labels.size() * 8);
    if (rc == CVTMI_OK) rc = up(h->upper_off, uoff.data(), uoff.size() * 8);
    if (rc == CVTMI_OK) rc = up(h->upper, upper.data(), upper.size() * 4);
    if (rc != CVTMI_OK) { cvtmi_hnsw_destroy(h); return rc; }
    h->g.vec = h->vec.as<float>(); h->g.links0 = h->links0.as<uint32_t>(); h->g.labels = h->labels.as<int64_t>();
    h->g.upper_off = h->upper_off.as<int64_t>(); h->g.upper = h->upper.as<uint32_t>();
    h->g.n = n; h->g.D = D; h->g.maxM = (int)maxM; h->g.maxM0 = (int)maxM0; h->g.maxlevel = n > 0 ? maxlevel : 0;
    h->g.enterpoint = enterpoint;
    *out = h;
    return CVTMI_OK;
}

int cvtmi_hnsw_destroy(cvtmi_hnsw_t h)
{
    if (!h) return CVTMI_OK;
    CHECK_HN(h);
    h->vec.release(); h->links0.release(); h->labels.release(); h->upper_off.release(); h->upper.release();
    (void)hipDeviceSynchronize();   // searches still in flight on other streams read the graph
    for (HnswScratch *c : h->pool) { c->release_all(); delete c; }
    h->pool.clear();
    h->magic = 0;
    delete h;
    return CVTMI_OK;
}

int64_t cvtmi_hnsw_ntotal(cvtmi_hnsw_t h) { return (h && h->magic == 0x484e5357u) ? h->g.n : -1; }

// scratch of one traversal launch: `slots` concurrent queries (one wave each), a visited bit per node and the spilled queues per slot
struct HnswPlan { int slots; int64_t words, gcap; };
static int hnsw_plan(cvtmi_hnsw_t h, HnswScratch &S, int lds_dim, int64_t nq, int k, int ef, HnswPlan &pl, hipStream_t st)
{
    const int efe = ef > k ? ef : k;
    int per_cu = (159 * 1024) / hnsw_lds_bytes(lds_dim, efe);  // query slots (one wave each) a CU's 160 KB of LDS hold
    per_cu = per_cu > 32 ? 32 : (per_cu < 1 ? 1 : per_cu);
    if (const int cap = g_hnsw_slots_cap.load(); cap > 0 && per_cu > cap) per_cu = cap;   // cvtmi_set_tuning("hnsw_slots"): measurement hook
    // (filling the rounds of a batch evenly with fewer slots per CU was measured: no effect -- throughput grows with the traversals in
    //  flight all the way to 32 per CU: 12 / 16 / 20 / 24 / 28 / 32 slots -> 144 / 164 / 178 / 184 / 192 / 201 K queries/s over codes at ef = 1000)
    pl.slots = h->cus * per_cu;
    if (pl.slots > nq) pl.slots = (int)nq;
    pl.words = (h->g.n + 31) / 32 + 1;
    int64_t gcap = (int64_t)efe * h->g.maxM0 * 2;
    if (gcap > h->g.n) gcap = h->g.n;
    gcap = gcap > hnsw_lcap() ? gcap - hnsw_lcap() : 0;
    pl.gcap = gcap + 64;
    CVTMI_TRY(S.s_vis.reserve((size_t)pl.slots * pl.words * 4));
    CVTMI_TRY(S.s_cand.reserve((size_t)pl.slots * (pl.gcap + efe + 1) * 8));  // per slot: spilled top queue + spilled candidates
    CVTMI_TRY(S.s_err.reserve(16));
    CVTMI_HIP(hipMemsetAsync(S.s_err.p, 0, 8, st));  // [0] overflow flag, [1] query counter
    return CVTMI_OK;
}
static int hnsw_check_overflow(HnswScratch &S, const char *who, int ef, hipStream_t st)
{
    int err = 0;
    CVTMI_HIP(hipMemcpyAsync(&err, S.s_err.p, 4, hipMemcpyDeviceToHost, st));
    CVTMI_HIP(stream_wait(st));
    if (err) return fail(CVTMI_EUNSUPPORTED, "%s: candidate queue overflow (ef=%d)", who, ef);
    return CVTMI_OK;
}

static int hnsw_search_leased(cvtmi_hnsw_t h, HnswScratch &S, const float *q, int64_t nq, int k, int ef, float *dist, int64_t *labels, hipStream_t st)
{
    HnswPlan pl;
    CVTMI_TRY(hnsw_plan(h, S, h->D, nq, k, ef, pl, st));
    CVTMI_TRY(launch_hnsw_search(h->g, h->metric, q, nq, k, ef, dist, labels, S.s_vis.as<uint32_t>(), S.s_cand.p, pl.slots, pl.words,
                                 pl.gcap, S.s_err.as<int>(), st));
    return hnsw_check_overflow(S, "cvtmi_hnsw_search", ef, st);
}

static int hnsw_search_args(cvtmi_hnsw_t h, const char *who, const void *q, int64_t nq, int k, int ef, const void *dist, const void *labels)
{
    (void)h;
    if (nq < 0 || (nq > 0 && (!q || !dist || !labels))) return fail(CVTMI_EINVAL, "%s: bad arguments", who);
    if (k < 1 || k > hnsw_ef_max()) return fail(CVTMI_EUNSUPPORTED, "%s: k=%d outside 1..%d", who, k, hnsw_ef_max());
    if (ef < 1 || ef > hnsw_ef_max()) return fail(CVTMI_EUNSUPPORTED, "%s: ef=%d outside 1..%d", who, ef, hnsw_ef_max());
    if (nq > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "%s: nq too large", who);
    return CVTMI_OK;
}

int cvtmi_hnsw_search_dev(cvtmi_hnsw_t h, const float *q, int64_t nq, int k, int ef, float *dist, int64_t *labels, void *stream)
{
    CHECK_HN(h);
    CVTMI_TRY(hnsw_search_args(h, "cvtmi_hnsw_search", q, nq, k, ef, dist, labels));
    if (nq == 0) return CVTMI_OK;
    HnswLease lease;
    CVTMI_TRY(lease.open(h, (hipStream_t)stream, false));
    return hnsw_search_leased(h, *lease.s, q, nq, k, ef, dist, labels, lease.st);
}

// host pointers in and out: staged through the leased set's own buffers, on its own stream
template <typename F> static int hnsw_host_call(cvtmi_hnsw_t h, const float *q, int64_t nq, int k, float *dist, int64_t *labels, F &&run)
{
    HnswLease lease;
    CVTMI_TRY(lease.open(h, nullptr, true));
    HnswScratch &S = *lease.s;
    hipStream_t st = lease.st;
    const size_t qb = (size_t)nq * h->D * sizeof(float), db = (size_t)nq * k * 4, lb = (size_t)nq * k * 8;
    CVTMI_TRY(S.io_q.reserve(qb));
    CVTMI_TRY(S.io_d.reserve(db));
    CVTMI_TRY(S.io_l.reserve(lb));
    CVTMI_HIP(hipMemcpyAsync(S.io_q.p, q, qb, hipMemcpyHostToDevice, st));
    CVTMI_TRY(run(S, S.io_q.as<float>(), S.io_d.as<float>(), S.io_l.as<int64_t>(), st));
    CVTMI_HIP(hipMemcpyAsync(dist, S.io_d.p, db, hipMemcpyDeviceToHost, st));
    CVTMI_HIP(hipMemcpyAsync(labels, S.io_l.p, lb, hipMemcpyDeviceToHost, st));
    CVTMI_HIP(stream_wait(st));
    return CVTMI_OK;
}

int cvtmi_hnsw_search(cvtmi_hnsw_t h, const float *q, int64_t nq, int k, int ef, float *dist, int64_t *labels)
{
    CHECK_HN(h);
    CVTMI_TRY(hnsw_search_args(h, "cvtmi_hnsw_search", q, nq, k, ef, dist, labels));
    if (nq == 0) return CVTMI_OK;
    return hnsw_host_call(h, q, nq, k, dist, labels, [&](HnswScratch &S, const float *dq, float *dd, int64_t *dl, hipStream_t st) {
        return hnsw_search_leased(h, S, dq, nq, k, ef, dd, dl, st);
    });
}

// HNSW over OPQ-compressed vectors: the graph of `h`, distances = ADC over the codes held by `opq` (one code
// row per graph node, appended in internal-id order).  Queries are rotated and their tables built by the OPQ
// handle's own kernels, into a scratch set leased from the OPQ handle (so a later cvtmi_opq_add waits for this search);
// the OPQ handle is held shared for the duration, like a search of its own.
static int hnsw_search_adc_leased(cvtmi_hnsw_t h, HnswScratch &S, cvtmi_opq_t opq, const float *q, int64_t nq, int rotate, int k, int ef,
                                  float *dist, int64_t *labels, hipStream_t st, int raw_ids)
{
    if (!opq) return fail(CVTMI_EINVAL, "cvtmi_hnsw_search_adc: null OPQ handle");
    CVTMI_TRY(hnsw_search_args(h, "cvtmi_hnsw_search_adc", q, nq, k, ef, dist, labels));
    if (opq->m.coarseK != 1) return fail(CVTMI_EUNSUPPORTED, "cvtmi_hnsw_search_adc: needs an OPQ model with coarseK == 1");
    if (opq->m.D != h->D) return fail(CVTMI_EINVAL, "cvtmi_hnsw_search_adc: OPQ model is %d-d, graph is %d-d", opq->m.D, h->D);
    if (opq->device != h->device) return fail(CVTMI_EINVAL, "cvtmi_hnsw_search_adc: handles live on different devices");
    if (nq == 0) return CVTMI_OK;
    std::shared_lock<std::shared_timed_mutex> rd(opq->rw);
    if (opq->n != h->g.n) return fail(CVTMI_EINVAL, "cvtmi_hnsw_search_adc: %lld code rows for %lld graph nodes", (long long)opq->n,
                                      (long long)h->g.n);
    OpqLease ol;
    CVTMI_TRY(ol.open(opq, st, false));
    OpqScratch &OS = *ol.s;
    const float *q_rot = q;
    if (rotate && (opq->m.perm || opq->m.R)) {
        CVTMI_TRY(OS.s_qrot.reserve((size_t)nq * opq->m.D * sizeof(float)));
        CVTMI_TRY(opq_rotate_impl(opq, q, nq, OS.s_qrot.as<float>(), st));
        q_rot = OS.s_qrot.as<float>();
    }
    CVTMI_TRY(OS.s_lut.reserve((size_t)nq * opq->m.M * opq->m.K * sizeof(float)));
    CVTMI_TRY(launch_lut(opq->m, q_rot, nq, nullptr, OS.s_lut.as<float>(), st));
    HnswPlan pl;
    const int state_floats = hnsw_adc_state_floats(opq->m.M * opq->m.K);   // one reading of the tuning flag for the slot count AND the launch
    CVTMI_TRY(hnsw_plan(h, S, state_floats, nq, k, ef, pl, st));
    CVTMI_TRY(launch_hnsw_search_adc(h->g, OS.s_lut.as<float>(), opq->codes.as<uint8_t>(), opq->m.M, opq->m.K, nq, k, ef, dist,
                                     labels, S.s_vis.as<uint32_t>(), S.s_cand.p, pl.slots, pl.words, pl.gcap, S.s_err.as<int>(), st, raw_ids,
                                     state_floats));
    return hnsw_check_overflow(S, "cvtmi_hnsw_search_adc", ef, st);
}

int cvtmi_hnsw_search_adc_dev(cvtmi_hnsw_t h, cvtmi_opq_t opq, const float *q, int64_t nq, int rotate, int k, int ef, float *dist,
                              int64_t *labels, void *stream)
{
    CHECK_HN(h);
    HnswLease lease;
    CVTMI_TRY(lease.open(h, (hipStream_t)stream, false));
    return hnsw_search_adc_leased(h, *lease.s, opq, q, nq, rotate, k, ef, dist, labels, lease.st, 0);
}

// ADC traversal with a result list of `rerank` nodes, then their exact fp32 distances (the graph's own vectors, the summation
// order of the reference's distance functions) and the k smallest; equal exact distances keep their ADC order
static int hnsw_search_adc_rerank_leased(cvtmi_hnsw_t h, HnswScratch &S, cvtmi_opq_t opq, const float *q, int64_t nq, int rotate, int k, int ef,
                                         int rerank, float *dist, int64_t *labels, hipStream_t st)
{
    if (k < 1 || k > CVTMI_K_MAX) return fail(CVTMI_EUNSUPPORTED, "cvtmi_hnsw_search_adc_rerank: k=%d outside 1..%d", k, CVTMI_K_MAX);
    if (rerank < k || rerank > hnsw_ef_max()) return fail(CVTMI_EINVAL, "cvtmi_hnsw_search_adc_rerank: rerank=%d outside k..%d", rerank, hnsw_ef_max());
    if (nq < 0 || (nq > 0 && (!q || !dist || !labels))) return fail(CVTMI_EINVAL, "cvtmi_hnsw_search_adc_rerank: bad arguments");
    if (nq == 0) return CVTMI_OK;
    CVTMI_TRY(S.s_rr_d.reserve((size_t)nq * rerank * sizeof(float)));
    CVTMI_TRY(S.s_rr_id.reserve((size_t)nq * rerank * sizeof(int64_t)));
    CVTMI_TRY(hnsw_search_adc_leased(h, S, opq, q, nq, rotate, rerank, ef, S.s_rr_d.as<float>(), S.s_rr_id.as<int64_t>(), st, 1));
    CVTMI_TRY(launch_hnsw_rerank(h->g, h->metric, q, nq, rerank, S.s_rr_id.as<int64_t>(), S.s_rr_d.as<float>(), st));
    CVTMI_TRY(launch_topk_select(S.s_rr_d.as<float>(), S.s_rr_id.as<int64_t>(), nq, rerank, k, dist, labels, st));
    return launch_gather_labels(labels, nq * k, h->g.labels, st);
}

int cvtmi_hnsw_search_adc_rerank_dev(cvtmi_hnsw_t h, cvtmi_opq_t opq, const float *q, int64_t nq, int rotate, int k, int ef, int rerank,
                                     float *dist, int64_t *labels, void *stream)
{
    CHECK_HN(h);
    HnswLease lease;
    CVTMI_TRY(lease.open(h, (hipStream_t)stream, false));
    return hnsw_search_adc_rerank_leased(h, *lease.s, opq, q, nq, rotate, k, ef, rerank, dist, labels, lease.st);
}

int cvtmi_hnsw_search_adc_rerank(cvtmi_hnsw_t h, cvtmi_opq_t opq, const float *q, int64_t nq, int rotate, int k, int ef, int rerank,
                                 float *dist, int64_t *labels)
{
    CHECK_HN(h);
    if (nq < 0 || k < 1 || (nq > 0 && (!q || !dist || !labels))) return fail(CVTMI_EINVAL, "cvtmi_hnsw_search_adc_rerank: bad arguments");
    if (nq == 0) return CVTMI_OK;
    return hnsw_host_call(h, q, nq, k, dist, labels, [&](HnswScratch &S, const float *dq, float *dd, int64_t *dl, hipStream_t st) {
        return hnsw_search_adc_rerank_leased(h, S, opq, dq, nq, rotate, k, ef, rerank, dd, dl, st);
    });
}

int cvtmi_hnsw_search_adc(cvtmi_hnsw_t h, cvtmi_opq_t opq, const float *q, int64_t nq, int rotate, int k, int ef, float *dist,
                          int64_t *labels)
{
    CHECK_HN(h);
    if (nq < 0 || k < 1 || (nq > 0 && (!q || !dist || !labels))) return fail(CVTMI_EINVAL, "cvtmi_hnsw_search_adc: bad arguments");
    if (nq == 0) return CVTMI_OK;
    return hnsw_host_call(h, q, nq, k, dist, labels, [&](HnswScratch &S, const float *dq, float *dd, int64_t *dl, hipStream_t st) {
        return hnsw_search_adc_leased(h, S, opq, dq, nq, rotate, k, ef, dd, dl, st, 0);
    });
}
