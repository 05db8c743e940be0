// shard.hip -- the exchange step of the row-sharded search (north star): one process per GPU, every rank
// scans its contiguous block of code rows, then ONE all-gather of the per-shard (distance, id) top-k lists over
// RCCL / xGMI and a k-way merge on every rank.  Same shape as the only distributed search in the reference tree
// (FLANN-MPI: local search, id += offset, reduce with ResultsMerger; retrieval/vlindex/lib/FLANN/mpi/index.h:74-108,
// :196-226) with the tree reduce replaced by a single collective: the message is tiny (12 bytes per result, 12 MB per
// rank at 10 K queries x top-100) and xGMI is a full point-to-point mesh, so one all-gather costs one latency.
//
// Buffer layout: the communicator owns `world` slots of
//     [status word, padded to 16 B][nq * k fp32 distances, padded to 16 B][nq * k int64 ids, padded to 16 B]
// The local search writes straight into slot `rank` (no pack kernel, no copy), ncclAllGather runs IN PLACE over the
// slots as one byte message per rank, and topk_merge_kernel<true> reads the gathered lists where they landed.  Rank
// order is ascending id range, which is the merge's tie rule.  Everything is enqueued on the caller's stream.
// (uint8 L2 searches send their int32 distances through the same fp32 fields: non-negative ints order like their bit
// patterns read as floats, and nothing does arithmetic on them.)
//
// Failure: a rank whose local search failed still enters the collective -- with its error code in the status word --
// so nobody is left waiting; after the all-gather every rank reads the `world` status words back (one small copy + a
// stream synchronisation; cvtmi_set_tuning("comm_check_status", 0) skips it) and ALL ranks return CVTMI_ECOMM.
//
// One process can also drive every GPU of the node (cvtmi_comm_create_all: ncclCommInitAll; the searches of the
// devices are enqueued one after the other, the all-gathers go out as one ncclGroupStart / ncclGroupEnd group).
//
// RCCL is bound at run time (dlopen of librccl.so.1 on the first communicator): processes that never shard -- the CPU
// boundary tests, the single-GPU CLIs -- do not load it, and a process that already has RCCL mapped (torch) shares it.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include "host_util.h"
#include "kernels.h"
#include "shard.h"

namespace cvtmi {

namespace {

struct RcclApi {
    void *so = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string why;
};

RcclApi *rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
      try {   // nothing may throw across the C ABI (std::string can)
        const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        for (const char *nm : names) {
            api.so = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (api.so) break;
        }
        if (!api.so) {
            const char *e = dlerror();   // (one call: dlerror() clears the state it reports)
            api.why = e ? e : "librccl.so.1 not found";
            return;
        }
        auto sym = [&](const char *nm) -> void * {
            void *p = dlsym(api.so, nm);
            if (!p && api.why.empty()) api.why = std::string("symbol missing in librccl: ") + nm;
            return p;
        };
        api.GetUniqueId = reinterpret_cast<decltype(&ncclGetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(&ncclCommInitRank)>(sym("ncclCommInitRank"));
        api.CommInitAll = reinterpret_cast<decltype(&ncclCommInitAll)>(sym("ncclCommInitAll"));
        api.GroupStart = reinterpret_cast<decltype(&ncclGroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(&ncclGroupEnd)>(sym("ncclGroupEnd"));
        api.AllGather = reinterpret_cast<decltype(&ncclAllGather)>(sym("ncclAllGather"));
        api.CommDestroy = reinterpret_cast<decltype(&ncclCommDestroy)>(sym("ncclCommDestroy"));
        api.GetErrorString = reinterpret_cast<decltype(&ncclGetErrorString)>(sym("ncclGetErrorString"));
      } catch (...) {
        api.so = nullptr;
      }
    });
    return &api;
}

int rccl_ready(RcclApi **out)
{
    RcclApi *a = rccl();
    if (!a->so || !a->why.empty()) return fail(CVTMI_ECOMM, "RCCL is not available: %s", a->why.empty() ? "librccl could not be loaded" : a->why.c_str());
    *out = a;
    return CVTMI_OK;
}

#define CVTMI_NCCL(api, expr)                                                                                      \
    do {                                                                                                           \
        ncclResult_t r__ = (expr);                                                                                 \
        if (r__ != ncclSuccess)                                                                                    \
            return ::cvtmi::fail(CVTMI_ECOMM, "%s failed: %s (%s:%d)", #expr, (api)->GetErrorString(r__), __FILE__, __LINE__); \
    } while (0)

int g_force_rccl = 0;  // cvtmi_set_tuning("comm_force_rccl"): world == 1 communicators go through RCCL too (tests on a 1-GPU box)
// cvtmi_set_tuning("comm_check_status"): what happens to the ranks' status words after every all-gather.
//   2 (default) deferred: a kernel behind the merge looks at them on the device; if any rank failed it overwrites the call's results with
//     (+inf, -1) and leaves (code, rank) in a host-mapped word of the communicator -- no stream synchronisation.  The error is returned by
//     the NEXT call on the communicator, by cvtmi_comm_status, and by the host-pointer entries at once (they synchronise for their copy anyway).
//   1 immediate: read back + stream synchronisation inside the call (every rank returns CVTMI_ECOMM from the failing call itself).
//   0 ignored (only this rank's own failure is reported).
int g_check_status = 2;
constexpr size_t kSlotHeader = 16;

constexpr uint32_t kCommMagic = 0x434f4d4du;

inline size_t align16(size_t b) { return (b + 15) & ~(size_t)15; }

}  // namespace

void comm_set_force_rccl(int v) { g_force_rccl = v; }
void comm_set_check_status(int v) { g_check_status = v; }

}  // namespace cvtmi

using namespace cvtmi;

struct cvtmi_comm_s {
    uint32_t magic = kCommMagic;
    int device = 0, rank = 0, world = 1;
    ncclComm_t nccl = nullptr;       // RCCL transport
    cvtmi_allgather_fn fn = nullptr;  // caller-supplied transport
    void *ctx = nullptr;
    HandleSync sync;
    DevBuf gather;  // world slots, see the layout note on top
    int64_t n_collectives = 0, last_bytes_per_rank = 0;
    uint32_t *h_status = nullptr;   // [world] host copy of the status words of the last exchange (comm_check_status = 1)
    uint32_t *h_sticky = nullptr;   // host-mapped [2]: (code, rank + 1) of the first failure a deferred check saw; 0 = none
    uint32_t *d_sticky = nullptr;   // its device address
};

namespace cvtmi {

static int comm_check(cvtmi_comm_t c)
{
    if (!c || c->magic != kCommMagic) return fail(CVTMI_EINVAL, "bad communicator handle");
    int cur = -1;
    CVTMI_HIP(hipGetDevice(&cur));
    if (cur != c->device) CVTMI_HIP(hipSetDevice(c->device));
    return CVTMI_OK;
}

size_t comm_slot_bytes(int64_t nq, int k)
{
    return kSlotHeader + align16((size_t)nq * k * sizeof(float)) + align16((size_t)nq * k * sizeof(int64_t));
}

int comm_validate(cvtmi_comm_t c)
{
    if (!c || c->magic != kCommMagic) return fail(CVTMI_EINVAL, "bad communicator handle");
    return CVTMI_OK;
}
int comm_world(cvtmi_comm_t c) { return c ? c->world : 1; }
int comm_rank(cvtmi_comm_t c) { return c ? c->rank : 0; }
int comm_device(cvtmi_comm_t c) { return c ? c->device : -1; }
bool comm_has_transport(cvtmi_comm_t c) { return c && (c->nccl || c->fn); }
HandleSync *comm_sync(cvtmi_comm_t c) { return &c->sync; }

// this rank's slot for a [nq][k] result: where the local search writes
int comm_local_slot(cvtmi_comm_t c, int64_t nq, int k, float **dist, int64_t **ids)
{
    CVTMI_TRY(comm_check(c));
    const size_t slot = comm_slot_bytes(nq, k);
    CVTMI_TRY(c->gather.reserve(std::max<size_t>(slot * c->world, 16)));
    uint8_t *mine = c->gather.as<uint8_t>() + (size_t)c->rank * slot + kSlotHeader;
    *dist = reinterpret_cast<float *>(mine);
    *ids = reinterpret_cast<int64_t *>(mine + align16((size_t)nq * k * sizeof(float)));
    return CVTMI_OK;
}

// the status word of this rank's slot (enqueued); the slot exists afterwards even if comm_local_slot never ran.
// The ONE failure that cannot travel through the collective is this rank running out of device memory for the gather buffer
// itself (world x slot bytes): there is nothing to all-gather into, the call returns CVTMI_ENOMEM before the collective and the
// peers stay inside theirs (RCCL has no time-out).  cvtmi_comm_reserve takes that allocation out of the search path: call it
// once per communicator with the largest (nq, k) it will see, at a point where a failure can still be reported out of band.
static int comm_post_status(cvtmi_comm_t c, int64_t nq, int k, int status, hipStream_t st)
{
    const size_t slot = comm_slot_bytes(nq, k);
    CVTMI_TRY(c->gather.reserve(std::max<size_t>(slot * c->world, 16)));
    CVTMI_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c->gather.as<uint8_t>() + (size_t)c->rank * slot), status, 1, st));
    return CVTMI_OK;
}
static int comm_allgather(cvtmi_comm_t c, size_t slot, hipStream_t st)
{
    uint8_t *base = c->gather.as<uint8_t>();
    if (c->nccl) {
        RcclApi *api = nullptr;
        CVTMI_TRY(rccl_ready(&api));
        // in place: sendbuff == recvbuff + rank * sendcount
        CVTMI_NCCL(api, api->AllGather(base + (size_t)c->rank * slot, base, slot, ncclUint8, c->nccl, st));
    } else if (c->fn) {
        const int rc = c->fn(c->ctx, base + (size_t)c->rank * slot, base, slot, (void *)st);
        if (rc != 0) return fail(CVTMI_ECOMM, "caller-supplied all-gather failed with %d", rc);
    }  // world == 1 without a transport: the one slot is already the gathered buffer
    c->n_collectives += (c->nccl || c->fn) ? 1 : 0;
    c->last_bytes_per_rank = (int64_t)slot;
    return CVTMI_OK;
}
// deferred check (comm_check_status = 2): runs behind the merge.  Any non-zero status word: the results are void on every rank (the failing
// rank's slot holds lists of an earlier search) -- they are overwritten with the padding pattern (+inf, -1) -- and the first failure is
// left where the host finds it without asking the device.
__global__ void comm_status_fold_kernel(const uint8_t *__restrict__ gather, size_t slot, int world, uint32_t *sticky, float *dist, int64_t *ids,
                                        int64_t count)
{
    uint32_t code = 0, who = 0;
    for (int r = world - 1; r >= 0; --r) {
        const uint32_t v = *reinterpret_cast<const uint32_t *>(gather + (size_t)r * slot);
        if (v) { code = v; who = (uint32_t)r + 1; }
    }
    if (!code) return;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        dist[i] = __uint_as_float(0x7f800000u);
        ids[i] = -1;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && __hip_atomic_load(&sticky[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) {
        __hip_atomic_store(&sticky[0], code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&sticky[1], who, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
static int comm_sticky_ready(cvtmi_comm_t c)
{
    if (c->h_sticky) return CVTMI_OK;
    void *hp = nullptr, *dp = nullptr;
    CVTMI_HIP(hipHostMalloc(&hp, 16, hipHostMallocMapped));
    memset(hp, 0, 16);
    if (hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) { (void)hipHostFree(hp); return fail(CVTMI_EHIP, "communicator: no device view of pinned memory"); }
    c->h_sticky = static_cast<uint32_t *>(hp);
    c->d_sticky = static_cast<uint32_t *>(dp);
    return CVTMI_OK;
}
// a failure a deferred check has recorded since the last report: CVTMI_ECOMM once, then forgotten
int comm_take_deferred(cvtmi_comm_t c)
{
    if (!c || !c->h_sticky) return CVTMI_OK;
    const uint32_t who = __atomic_load_n(&c->h_sticky[1], __ATOMIC_ACQUIRE);
    if (!who) return CVTMI_OK;
    const int code = (int)c->h_sticky[0], r = (int)who - 1;
    c->h_sticky[0] = 0u;
    __atomic_store_n(&c->h_sticky[1], 0u, __ATOMIC_RELEASE);
    return fail(CVTMI_ECOMM, "row-sharded search: rank %d reported error %d%s in an earlier search on this communicator (its results were voided: +inf / -1)",
                r, code, r == c->rank ? " (this rank)" : "");
}

// every rank's status word -> host; non-zero anywhere: CVTMI_ECOMM on every rank
static int comm_check_statuses(cvtmi_comm_t c, size_t slot, int own_status, hipStream_t st)
{
    if (g_check_status != 1) return own_status == CVTMI_OK ? CVTMI_OK : fail(CVTMI_ECOMM, "the local search of rank %d failed with %d", c->rank, own_status);
    if (!c->h_status) {
        c->h_status = new (std::nothrow) uint32_t[c->world];
        if (!c->h_status) return fail(CVTMI_ENOMEM, "communicator: out of host memory");
    }
    CVTMI_HIP(hipMemcpy2DAsync(c->h_status, sizeof(uint32_t), c->gather.p, slot, sizeof(uint32_t), (size_t)c->world, hipMemcpyDeviceToHost, st));
    CVTMI_HIP(stream_wait(st));
    for (int r = 0; r < c->world; ++r)
        if (c->h_status[r] != 0u)
            return fail(CVTMI_ECOMM, "row-sharded search: rank %d reported error %d%s", r, (int)c->h_status[r], r == c->rank ? " (this rank)" : "");
    return CVTMI_OK;
}

// all-gather of the slots (one collective) + merge of the world lists per query into dist / ids.  status: what the local
// search of this rank returned -- it travels in the slot header, and a failure anywhere fails the call on every rank
int comm_exchange_merge(cvtmi_comm_t c, int64_t nq, int k, int status, float *dist, int64_t *ids, hipStream_t st)
{
    CVTMI_TRY(comm_check(c));
    if (nq <= 0) return CVTMI_OK;
    const size_t slot = comm_slot_bytes(nq, k);
    const std::string own = status != CVTMI_OK ? std::string(cvtmi_last_error()) : std::string();
    const bool deferred = g_check_status == 2 && c->world > 1;
    if (deferred) CVTMI_TRY(comm_sticky_ready(c));
    // what an earlier search left behind is reported AFTER this rank has entered the collective: the other ranks are on their way into it
    const int earlier = comm_take_deferred(c);
    const std::string earlier_msg = earlier != CVTMI_OK ? std::string(cvtmi_last_error()) : std::string();
    CVTMI_TRY(comm_post_status(c, nq, k, status, st));
    CVTMI_TRY(comm_allgather(c, slot, st));
    const int rc = comm_check_statuses(c, slot, status, st);
    if (rc != CVTMI_OK) {
        // (a failure an EARLIER search left behind was taken off the communicator above: it is reported with this one, not dropped)
        const std::string now = status != CVTMI_OK ? own : std::string(cvtmi_last_error());
        const char *sep = earlier != CVTMI_OK ? "; and before that: " : "";
        if (status != CVTMI_OK)
            return fail(CVTMI_ECOMM, "row-sharded search: the local search of rank %d failed with %d: %s%s%s", c->rank, status, now.c_str(), sep, earlier_msg.c_str());
        return fail(CVTMI_ECOMM, "%s%s%s", now.c_str(), sep, earlier_msg.c_str());
    }
    uint8_t *base = c->gather.as<uint8_t>() + kSlotHeader;
    const size_t ids_off = align16((size_t)nq * k * sizeof(float));
    CVTMI_TRY(launch_topk_merge_gathered(reinterpret_cast<const float *>(base), reinterpret_cast<const int64_t *>(base + ids_off),
                                         (int64_t)(slot / sizeof(float)), (int64_t)(slot / sizeof(int64_t)), nq, c->world, k, dist, ids, st));
    if (deferred) {
        const int64_t count = nq * k;
        hipLaunchKernelGGL(comm_status_fold_kernel, dim3((unsigned)std::min<int64_t>(256, (count + 255) / 256)), dim3(256), 0, st, c->gather.as<uint8_t>(), slot,
                           c->world, c->d_sticky, dist, ids, count);
        CVTMI_HIP(hipGetLastError());
    }
    if (earlier != CVTMI_OK) return fail(CVTMI_ECOMM, "%s", earlier_msg.c_str());
    return CVTMI_OK;
}

// The same exchange for the communicators of ONE process (cvtmi_comm_create_all), one per device: status words, ONE group of
// all-gathers (ncclGroupStart .. ncclGroupEnd: issued from a single thread they would otherwise wait for each other), status
// check, merge on the device of comms[0] only (dist / ids live there).  Everything on the devices' null streams.
int comm_exchange_merge_all(cvtmi_comm_t *comms, int ndev, int64_t nq, int k, const int *status, float *dist, int64_t *ids)
{
    if (nq <= 0) return CVTMI_OK;
    const size_t slot = comm_slot_bytes(nq, k);
    RcclApi *api = nullptr;
    CVTMI_TRY(rccl_ready(&api));
    // every device is driven by this process: the local statuses are all here, on the host.  A failed local search fails the call
    // before any collective is issued (nobody else could be left waiting), whatever "comm_check_status" says -- the read-back
    // below only looks at comms[0]'s copy, and with it switched off a failure on device d > 0 would merge that device's stale slot.
    for (int d = 0; d < ndev; ++d)
        if (status[d] != CVTMI_OK)
            return fail(CVTMI_ECOMM, "row-sharded search: the local search on device %d (rank %d) failed with %d: %s", comms[d] ? comms[d]->device : -1, d,
                        status[d], std::string(cvtmi_last_error()).c_str());
    for (int d = 0; d < ndev; ++d) {
        CVTMI_TRY(comm_check(comms[d]));
        CVTMI_TRY(comm_post_status(comms[d], nq, k, status[d], nullptr));
    }
    CVTMI_NCCL(api, api->GroupStart());
    for (int d = 0; d < ndev; ++d) {
        cvtmi_comm_t c = comms[d];
        if (comm_check(c) != CVTMI_OK) { (void)api->GroupEnd(); return CVTMI_EHIP; }
        uint8_t *base = c->gather.as<uint8_t>();
        ncclResult_t r = api->AllGather(base + (size_t)c->rank * slot, base, slot, ncclUint8, c->nccl, nullptr);
        if (r != ncclSuccess) { (void)api->GroupEnd(); return fail(CVTMI_ECOMM, "ncclAllGather (device %d) failed: %s", c->device, api->GetErrorString(r)); }
        c->n_collectives += 1;
        c->last_bytes_per_rank = (int64_t)slot;
    }
    CVTMI_NCCL(api, api->GroupEnd());
    cvtmi_comm_t c0 = comms[0];
    CVTMI_TRY(comm_check(c0));
    // (no read-back: every rank of this communicator set is driven by this process, and their statuses were checked above)
    uint8_t *base = c0->gather.as<uint8_t>() + kSlotHeader;
    const size_t ids_off = align16((size_t)nq * k * sizeof(float));
    return launch_topk_merge_gathered(reinterpret_cast<const float *>(base), reinterpret_cast<const int64_t *>(base + ids_off),
                                      (int64_t)(slot / sizeof(float)), (int64_t)(slot / sizeof(int64_t)), nq, c0->world, k, dist, ids, nullptr);
}

}  // namespace cvtmi

extern "C" {

int cvtmi_comm_status(cvtmi_comm_t c)
{
    CVTMI_TRY(comm_validate(c));
    Serial serial(*comm_sync(c), nullptr);
    return comm_take_deferred(c);
}

int cvtmi_comm_reserve(cvtmi_comm_t c, int64_t nq, int k)
{
    CVTMI_TRY(comm_validate(c));
    if (nq < 0 || k < 1) return fail(CVTMI_EINVAL, "cvtmi_comm_reserve: bad arguments");
    Serial serial(*comm_sync(c), nullptr);
    CVTMI_TRY(comm_check(c));
    return c->gather.reserve(std::max<size_t>(comm_slot_bytes(nq, k) * c->world, 16));
}

int cvtmi_comm_unique_id(void *id)
{
    if (!id) return fail(CVTMI_EINVAL, "cvtmi_comm_unique_id: null");
    static_assert(sizeof(ncclUniqueId) == CVTMI_COMM_ID_BYTES, "CVTMI_COMM_ID_BYTES must match ncclUniqueId");
    RcclApi *api = nullptr;
    CVTMI_TRY(rccl_ready(&api));
    ncclUniqueId u;
    CVTMI_NCCL(api, api->GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return CVTMI_OK;
}

int cvtmi_comm_create(const void *id, int rank, int world, cvtmi_comm_t *out)
{
    if (!out) return fail(CVTMI_EINVAL, "cvtmi_comm_create: null out");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(CVTMI_EINVAL, "cvtmi_comm_create: rank %d of %d", rank, world);
    if ((world > 1 || g_force_rccl) && !id) return fail(CVTMI_EINVAL, "cvtmi_comm_create: null id");
    int dev = 0;
    CVTMI_HIP(hipGetDevice(&dev));
    cvtmi_comm_s *c = new (std::nothrow) cvtmi_comm_s();
    if (!c) return fail(CVTMI_ENOMEM, "cvtmi_comm_create: out of host memory");
    c->device = dev; c->rank = rank; c->world = world;
    // the word of the deferred status check is allocated HERE, before anybody is inside a collective: a rank that cannot get it fails at
    // creation, not on its way into a search's all-gather (where its peers would wait for ever)
    if (world > 1) { const int rs = comm_sticky_ready(c); if (rs != CVTMI_OK) { delete c; return rs; } }
    if (world > 1 || g_force_rccl) {
        RcclApi *api = nullptr;
        int rc = rccl_ready(&api);
        if (rc != CVTMI_OK) { if (c->h_sticky) (void)hipHostFree(c->h_sticky); delete c; return rc; }
        ncclUniqueId u;
        memcpy(&u, id, sizeof u);
        ncclResult_t r = api->CommInitRank(&c->nccl, world, u, rank);  // collective: every rank of the job is in here now
        if (r != ncclSuccess) {
            if (c->h_sticky) (void)hipHostFree(c->h_sticky);
            delete c;
            return fail(CVTMI_ECOMM, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, dev, api->GetErrorString(r));
        }
    }
    *out = c;
    return CVTMI_OK;
}

int cvtmi_comm_create_all(int ndev, const int *devices, cvtmi_comm_t *comms)
{
    if (ndev < 1 || !comms) return fail(CVTMI_EINVAL, "cvtmi_comm_create_all: bad arguments");
    for (int d = 0; d < ndev; ++d) comms[d] = nullptr;
    int avail = 0;
    CVTMI_HIP(hipGetDeviceCount(&avail));
    std::vector<int> devs(ndev);
    for (int d = 0; d < ndev; ++d) {
        devs[d] = devices ? devices[d] : d;
        if (devs[d] < 0 || devs[d] >= avail) return fail(CVTMI_EINVAL, "cvtmi_comm_create_all: device %d of %d", devs[d], avail);
        for (int e = 0; e < d; ++e)
            if (devs[e] == devs[d]) return fail(CVTMI_EINVAL, "cvtmi_comm_create_all: device %d listed twice", devs[d]);
    }
    RcclApi *api = nullptr;
    CVTMI_TRY(rccl_ready(&api));
    if (!api->CommInitAll || !api->GroupStart || !api->GroupEnd) return fail(CVTMI_ECOMM, "RCCL lacks ncclCommInitAll / ncclGroupStart");
    std::vector<ncclComm_t> nc(ndev, nullptr);
    ncclResult_t r = api->CommInitAll(nc.data(), ndev, devs.data());
    if (r != ncclSuccess) return fail(CVTMI_ECOMM, "ncclCommInitAll(%d devices) failed: %s", ndev, api->GetErrorString(r));
    for (int d = 0; d < ndev; ++d) {
        cvtmi_comm_s *c = new (std::nothrow) cvtmi_comm_s();
        if (!c) {
            for (int e = 0; e < ndev; ++e) { if (comms[e]) { comms[e]->nccl = nullptr; delete comms[e]; comms[e] = nullptr; } (void)api->CommDestroy(nc[e]); }
            return fail(CVTMI_ENOMEM, "cvtmi_comm_create_all: out of host memory");
        }
        c->device = devs[d]; c->rank = d; c->world = ndev; c->nccl = nc[d];
        comms[d] = c;
    }
    return CVTMI_OK;
}

int cvtmi_comm_create_custom(cvtmi_allgather_fn fn, void *ctx, int rank, int world, cvtmi_comm_t *out)
{
    if (!out) return fail(CVTMI_EINVAL, "cvtmi_comm_create_custom: null out");
    *out = nullptr;
    if (!fn || world < 1 || rank < 0 || rank >= world) return fail(CVTMI_EINVAL, "cvtmi_comm_create_custom: bad arguments");
    int dev = 0;
    CVTMI_HIP(hipGetDevice(&dev));
    cvtmi_comm_s *c = new (std::nothrow) cvtmi_comm_s();
    if (!c) return fail(CVTMI_ENOMEM, "cvtmi_comm_create_custom: out of host memory");
    c->device = dev; c->rank = rank; c->world = world; c->fn = fn; c->ctx = ctx;
    if (world > 1) { const int rs = comm_sticky_ready(c); if (rs != CVTMI_OK) { delete c; return rs; } }   // (as in cvtmi_comm_create)
    *out = c;
    return CVTMI_OK;
}

int cvtmi_comm_destroy(cvtmi_comm_t c)
{
    if (!c) return CVTMI_OK;
    if (c->magic != kCommMagic) return fail(CVTMI_EINVAL, "cvtmi_comm_destroy: bad handle");
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    if (c->nccl) {
        RcclApi *api = rccl();
        if (api->CommDestroy) (void)api->CommDestroy(c->nccl);
    }
    c->gather.release();
    c->sync.destroy();
    delete[] c->h_status;
    if (c->h_sticky) (void)hipHostFree(c->h_sticky);
    c->magic = 0;
    delete c;
    return CVTMI_OK;
}

int cvtmi_comm_info(cvtmi_comm_t c, int *rank, int *world, int *transport, int64_t *collectives, int64_t *bytes_per_rank)
{
    if (!c || c->magic != kCommMagic) return fail(CVTMI_EINVAL, "cvtmi_comm_info: bad handle");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (transport) *transport = c->nccl ? 1 : (c->fn ? 2 : 0);
    if (collectives) *collectives = c->n_collectives;
    if (bytes_per_rank) *bytes_per_rank = c->last_bytes_per_rank;
    return CVTMI_OK;
}

int cvtmi_comm_slot_bytes(int64_t nq, int k, size_t *bytes)
{
    if (nq < 0 || k < 1 || !bytes) return fail(CVTMI_EINVAL, "cvtmi_comm_slot_bytes: bad arguments");
    *bytes = comm_slot_bytes(nq, k);
    return CVTMI_OK;
}

int cvtmi_shard_range(int64_t n_total, int rank, int world, int64_t *begin, int64_t *end)
{
    if (n_total < 0 || world < 1 || rank < 0 || rank >= world || !begin || !end) return fail(CVTMI_EINVAL, "cvtmi_shard_range: bad arguments");
    const int64_t base = n_total / world, rem = n_total % world;
    *begin = rank * base + std::min<int64_t>(rank, rem);
    *end = *begin + base + (rank < rem ? 1 : 0);
    return CVTMI_OK;
}

int cvtmi_shard_merge_topk_dev(cvtmi_comm_t c, const float *local_dist, const int64_t *local_ids, int64_t nq, int k, float *dist,
                               int64_t *ids, void *stream)
{
    CVTMI_TRY(comm_check(c));
    if (nq < 0 || k < 1 || k > CVTMI_K_MAX || (nq > 0 && (!local_dist || !local_ids || !dist || !ids)))
        return fail(CVTMI_EINVAL, "cvtmi_shard_merge_topk: bad arguments");
    if (nq == 0) return CVTMI_OK;
    hipStream_t st = (hipStream_t)stream;
    Serial serial(c->sync, st);
    float *sd = nullptr;
    int64_t *si = nullptr;
    CVTMI_TRY(comm_local_slot(c, nq, k, &sd, &si));
    CVTMI_HIP(hipMemcpyAsync(sd, local_dist, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToDevice, st));
    CVTMI_HIP(hipMemcpyAsync(si, local_ids, (size_t)nq * k * sizeof(int64_t), hipMemcpyDeviceToDevice, st));
    return comm_exchange_merge(c, nq, k, CVTMI_OK, dist, ids, st);
}

}  // extern "C"
