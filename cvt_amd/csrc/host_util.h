// host_util.h -- host-side helpers shared by the C-ABI glue (api.hip, shard.hip): device buffers and the
// per-handle serialisation of calls.
#pragma once
#include <atomic>
#include <chrono>
#include <mutex>

#include "common.h"

namespace cvtmi {

// growable device buffer
// Every allocation carries kDevSlack bytes beyond `cap`: streaming kernels may READ up to that far past the end of a buffer
// instead of clamping each lane's row index (the ADC scan's last 64-row chunk; what is read there is never used).
constexpr size_t kDevSlack = 4096;
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return CVTMI_OK;
        if (p) { CVTMI_HIP(hipFree(p)); p = nullptr; cap = 0; }
        hipError_t e = hipMalloc(&p, bytes + kDevSlack);
        if (e != hipSuccess) { p = nullptr; return fail(CVTMI_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
        cap = bytes;
        return CVTMI_OK;
    }
    // keeps the first `keep` bytes
    int grow(size_t bytes, size_t keep, hipStream_t st)
    {
        if (bytes <= cap) return CVTMI_OK;
        void *np = nullptr;
        hipError_t e = hipMalloc(&np, bytes + kDevSlack);
        if (e != hipSuccess) return fail(CVTMI_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        if (p && keep) {
            CVTMI_HIP(hipMemcpyAsync(np, p, keep, hipMemcpyDeviceToDevice, st));
            CVTMI_HIP(hipStreamSynchronize(st));
        }
        if (p) CVTMI_HIP(hipFree(p));
        p = np; cap = bytes;
        return CVTMI_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

template <class T>
inline int dev_alloc_copy(T **out, const T *host, size_t count)
{
    *out = nullptr;
    if (count == 0) return CVTMI_OK;
    CVTMI_HIP(hipMalloc((void **)out, count * sizeof(T)));
    CVTMI_HIP(hipMemcpy(*out, host, count * sizeof(T), hipMemcpyHostToDevice));
    return CVTMI_OK;
}

// temporary device allocation for the host-pointer entry points
// pinned host staging area of a handle (grow-only): host-pointer entry points copy through it instead of handing pageable
// memory to the runtime -- no per-call hipMalloc / hipFree (the free synchronises the device), and a stable DMA path
struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return CVTMI_OK;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
        if (e != hipSuccess) { p = nullptr; return fail(CVTMI_ENOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
        cap = bytes;
        return CVTMI_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct Tmp {
    void *p = nullptr;
    ~Tmp() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes)
    {
        if (bytes == 0) bytes = 16;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) { p = nullptr; return fail(CVTMI_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
        return CVTMI_OK;
    }
    int upload(const void *host, size_t bytes)
    {
        CVTMI_TRY(alloc(bytes));
        if (bytes) CVTMI_HIP(hipMemcpy(p, host, bytes, hipMemcpyHostToDevice));
        return CVTMI_OK;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Per-handle serialisation.  Every search borrows scratch buffers that belong to the handle (tables, partial top-k
// lists, visited bitmaps, lazily built copies of the rows), so two calls on one handle must not overlap -- neither on the
// host (two threads inside the library) nor on the device (two streams).  Serial holds the handle's lock for the duration
// of the host-side call, and when a call arrives on another stream than the previous one it makes that stream wait for
// the previous call's work (an event recorded when each outermost call returns).  Calls nest (host-pointer entries call
// their _dev twins), hence the recursive lock and the depth count.
// Waiting for a stream from a host-pointer entry: a blocking hipStreamSynchronize parks the thread on an interrupt, which costs tens of
// microseconds to wake from -- as much again as a whole small search.  Short waits therefore poll (hipStreamQuery) for up to
// cvtmi_set_tuning("host_spin_us") microseconds (default 200) before they block.
extern std::atomic<int> g_host_spin_us;
inline hipError_t stream_wait(hipStream_t st)
{
    const int budget = g_host_spin_us.load(std::memory_order_relaxed);
    if (budget > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0;; ++i) {
            const hipError_t e = hipStreamQuery(st);
            if (e != hipErrorNotReady) return e;
            if ((i & 7) == 7 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(budget)) break;
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
    }
    return hipStreamSynchronize(st);
}

struct HandleSync {
    std::recursive_mutex mu;
    hipEvent_t done = nullptr;
    hipStream_t last = nullptr;
    int depth = 0;
    bool pending = false;
    void destroy() { if (done) (void)hipEventDestroy(done); done = nullptr; }
};

struct Serial {
    HandleSync &s;
    hipStream_t st;
    Serial(HandleSync &sync, hipStream_t stream) : s(sync), st(stream)
    {
        s.mu.lock();
        if (s.depth++ == 0 && s.pending && s.last != st) (void)hipStreamWaitEvent(st, s.done, 0);
    }
    ~Serial()
    {
        if (--s.depth == 0) {
            if (!s.done) (void)hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
            if (s.done && hipEventRecord(s.done, st) == hipSuccess) { s.last = st; s.pending = true; }
        }
        s.mu.unlock();
    }
    Serial(const Serial &) = delete;
    Serial &operator=(const Serial &) = delete;
};

}  // namespace cvtmi
