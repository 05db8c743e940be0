// shard.hip -- the exchange step of the row-sharded search (north star): one process per GPU, every rank
// scans its contiguous block of code rows, then ONE all-gather of the per-shard (distance, id) top-k lists over
// RCCL / xGMI and a k-way merge on every rank.  Same shape as the only distributed search in the reference tree
// (FLANN-MPI: local search, id += offset, reduce with ResultsMerger; retrieval/vlindex/lib/FLANN/mpi/index.h:74-108,
// :196-226) with the tree reduce replaced by a single collective: the message is tiny (12 bytes per result, 12 MB per
// rank at 10 K queries x top-100) and xGMI is a full point-to-point mesh, so one all-gather costs one latency.
//
// Buffer layout: the communicator owns `world` slots of
//     [nq * k fp32 distances, padded to 16 B][nq * k int64 ids, padded to 16 B]
// The local search writes straight into slot `rank` (no pack kernel, no copy), ncclAllGather runs IN PLACE over the
// slots as one byte message per rank, and topk_merge_kernel<true> reads the gathered lists where they landed.  Rank
// order is ascending id range, which is the merge's tie rule.  Everything is enqueued on the caller's stream.
//
// RCCL is bound at run time (dlopen of librccl.so.1 on the first communicator): processes that never shard -- the CPU
// boundary tests, the single-GPU CLIs -- do not load it, and a process that already has RCCL mapped (torch) shares it.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <algorithm>
#include <new>

#include "host_util.h"
#include "kernels.h"
#include "shard.h"

namespace cvtmi {

namespace {

struct RcclApi {
    void *so = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
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
        const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        for (const char *nm : names) {
            api.so = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (api.so) break;
        }
        if (!api.so) { api.why = dlerror() ? dlerror() : "librccl.so.1 not found"; return; }
        auto sym = [&](const char *nm) -> void * {
            void *p = dlsym(api.so, nm);
            if (!p && api.why.empty()) api.why = std::string("symbol missing in librccl: ") + nm;
            return p;
        };
        api.GetUniqueId = reinterpret_cast<decltype(&ncclGetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(&ncclCommInitRank)>(sym("ncclCommInitRank"));
        api.AllGather = reinterpret_cast<decltype(&ncclAllGather)>(sym("ncclAllGather"));
        api.CommDestroy = reinterpret_cast<decltype(&ncclCommDestroy)>(sym("ncclCommDestroy"));
        api.GetErrorString = reinterpret_cast<decltype(&ncclGetErrorString)>(sym("ncclGetErrorString"));
    });
    return &api;
}

int rccl_ready(RcclApi **out)
{
    RcclApi *a = rccl();
    if (!a->so || !a->why.empty()) return fail(CVTMI_ECOMM, "RCCL is not available: %s", a->why.c_str());
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

constexpr uint32_t kCommMagic = 0x434f4d4du;

inline size_t align16(size_t b) { return (b + 15) & ~(size_t)15; }

}  // namespace

void comm_set_force_rccl(int v) { g_force_rccl = v; }

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

size_t comm_slot_bytes(int64_t nq, int k) { return align16((size_t)nq * k * sizeof(float)) + align16((size_t)nq * k * sizeof(int64_t)); }

int comm_world(cvtmi_comm_t c) { return c ? c->world : 1; }
int comm_device(cvtmi_comm_t c) { return c ? c->device : -1; }
bool comm_has_transport(cvtmi_comm_t c) { return c && (c->nccl || c->fn); }
HandleSync *comm_sync(cvtmi_comm_t c) { return &c->sync; }

// this rank's slot for a [nq][k] result: where the local search writes
int comm_local_slot(cvtmi_comm_t c, int64_t nq, int k, float **dist, int64_t **ids)
{
    CVTMI_TRY(comm_check(c));
    const size_t slot = comm_slot_bytes(nq, k);
    CVTMI_TRY(c->gather.reserve(std::max<size_t>(slot * c->world, 16)));
    uint8_t *mine = c->gather.as<uint8_t>() + (size_t)c->rank * slot;
    *dist = reinterpret_cast<float *>(mine);
    *ids = reinterpret_cast<int64_t *>(mine + align16((size_t)nq * k * sizeof(float)));
    return CVTMI_OK;
}

// all-gather of the slots (one collective) + merge of the world lists per query into dist / ids
int comm_exchange_merge(cvtmi_comm_t c, int64_t nq, int k, float *dist, int64_t *ids, hipStream_t st)
{
    CVTMI_TRY(comm_check(c));
    if (nq <= 0) return CVTMI_OK;
    const size_t slot = comm_slot_bytes(nq, k);
    uint8_t *base = c->gather.as<uint8_t>();
    if (c->gather.cap < slot * c->world) return fail(CVTMI_ESTATE, "comm_exchange_merge: slot not prepared");
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
    const size_t ids_off = align16((size_t)nq * k * sizeof(float));
    return launch_topk_merge_gathered(reinterpret_cast<const float *>(base), reinterpret_cast<const int64_t *>(base + ids_off),
                                      (int64_t)(slot / sizeof(float)), (int64_t)(slot / sizeof(int64_t)), nq, c->world, k, dist, ids, st);
}

}  // namespace cvtmi

extern "C" {

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
    if (world > 1 || g_force_rccl) {
        RcclApi *api = nullptr;
        int rc = rccl_ready(&api);
        if (rc != CVTMI_OK) { delete c; return rc; }
        ncclUniqueId u;
        memcpy(&u, id, sizeof u);
        ncclResult_t r = api->CommInitRank(&c->nccl, world, u, rank);  // collective: every rank of the job is in here now
        if (r != ncclSuccess) {
            delete c;
            return fail(CVTMI_ECOMM, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, dev, api->GetErrorString(r));
        }
    }
    *out = c;
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
    if (nq < 0 || k < 1 || k > 128 || (nq > 0 && (!local_dist || !local_ids || !dist || !ids)))
        return fail(CVTMI_EINVAL, "cvtmi_shard_merge_topk: bad arguments");
    if (nq == 0) return CVTMI_OK;
    hipStream_t st = (hipStream_t)stream;
    Serial serial(c->sync, st);
    float *sd = nullptr;
    int64_t *si = nullptr;
    CVTMI_TRY(comm_local_slot(c, nq, k, &sd, &si));
    CVTMI_HIP(hipMemcpyAsync(sd, local_dist, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToDevice, st));
    CVTMI_HIP(hipMemcpyAsync(si, local_ids, (size_t)nq * k * sizeof(int64_t), hipMemcpyDeviceToDevice, st));
    return comm_exchange_merge(c, nq, k, dist, ids, st);
}

}  // extern "C"
