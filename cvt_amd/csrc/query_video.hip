// query_video.hip -- the reference's own query semantics, IVFOPQ::Query / QueryThrehold
// (opq/src/IVFOPQ.cpp:213-320 / :322-422):
//   * per query frame keep the nk nearest coarse lists (:238-260: the first nk lists enter a
//     max-heap unconditionally, later ones replace the top when strictly closer -> the nk smallest
//     (distance, list) pairs; the visiting order does not matter because the scores are min-reduced);
//   * per probed list: residual LUT (:273-291) and ADC scan of that list's entries (:300-306);
//   * matchScore[f][videoId] = min(score, current), starting from threhold = 1.0 (:5, :262, :308).
// Scores are sums of squares (>= +0), so the fp32 min is an unsigned-integer atomicMin on the bits.
#include <algorithm>

#include "block_topk.h"
#include "kernels.h"

namespace cvtmi {

constexpr int PROBE_CAP = 512;
constexpr int PROBE_TRIG = 384;

__global__ __launch_bounds__(kBlock) void coarse_probe_kernel(const float *__restrict__ q_rot, int D,
                                                              const float *__restrict__ coarse, int coarseK, int nprobe,
                                                              int32_t *__restrict__ probe)
{
    extern __shared__ __attribute__((aligned(16))) float qv[];  // D floats
    __shared__ TopKShared<1, PROBE_CAP> tk;
    const int64_t qi = blockIdx.x;
    const int tid = threadIdx.x;
    for (int d = tid; d < D; d += kBlock) qv[d] = q_rot[qi * D + d];
    topk_init(tk);
    __syncthreads();
    int tile = 0;
    for (int base = 0; base < coarseK; base += kBlock, ++tile) {
        const int c = base + tid;
        uint32_t key[1][1] = { { KEY_MAX } };
        uint32_t pay[1] = { (uint32_t)c };
        if (c < coarseK) {
            const float *cp = coarse + (int64_t)c * D;
            float acc = 0.0f;
            for (int d = 0; d < D; ++d) {
                const float t = __fsub_rn(qv[d], cp[d]);
                acc = __fadd_rn(acc, __fmul_rn(t, t));
            }
            const uint32_t kk = __float_as_uint(acc);
            key[0][0] = kk == KEY_MAX ? KEY_MAX - 1 : kk;
        }
        topk_tile<1, 1, PROBE_CAP, PROBE_TRIG>(tk, nprobe, tile, key, pay);
    }
    __syncthreads();
    topk_compact(tk, nprobe);
    const int cnt = tk.cnt[0];
    for (int i = tid; i < nprobe; i += kBlock) probe[qi * nprobe + i] = i < cnt ? (int32_t)(uint32_t)tk.buf[0][i] : -1;
}

// Few queries (the reference's own Query call: a handful of frames) against many lists: one workgroup per query walks all coarseK
// centroids by itself -- 9 frames x 8192 lists kept 9 CUs busy for 169 us.  Here the centroids are cut into `splits` ranges, workgroup
// (split, query) keeps the nprobe nearest of its range (same distance arithmetic, same (distance, list) order), and one workgroup per
// query merges the splits' short lists.  The nprobe smallest (distance, list) pairs of a union are the nprobe smallest of the parts'.
__global__ __launch_bounds__(kBlock) void coarse_probe_split_kernel(const float *__restrict__ q_rot, int D, const float *__restrict__ coarse, int coarseK,
                                                                    int nprobe, int chunk, uint32_t *__restrict__ part_key, int32_t *__restrict__ part_id)
{
    extern __shared__ __attribute__((aligned(16))) float qv[];  // D floats
    __shared__ TopKShared<1, PROBE_CAP> tk;
    const int64_t qi = blockIdx.y;
    const int split = blockIdx.x, splits = gridDim.x, tid = threadIdx.x;
    for (int d = tid; d < D; d += kBlock) qv[d] = q_rot[qi * D + d];
    topk_init(tk);
    __syncthreads();
    const int c0 = split * chunk, c1 = c0 + chunk < coarseK ? c0 + chunk : coarseK;
    int tile = 0;
    for (int base = c0; base < c1; base += kBlock, ++tile) {
        const int c = base + tid;
        uint32_t key[1][1] = { { KEY_MAX } };
        uint32_t pay[1] = { (uint32_t)c };
        if (c < c1) {
            const float *cp = coarse + (int64_t)c * D;
            float acc = 0.0f;
            if ((D & 3) == 0) {   // (rows are 16-byte aligned when D is a multiple of 4: four dimensions per load, same operation order)
                for (int d = 0; d < D; d += 4) {
                    const float4 v = *reinterpret_cast<const float4 *>(cp + d);
                    const float t0 = __fsub_rn(qv[d], v.x), t1 = __fsub_rn(qv[d + 1], v.y), t2 = __fsub_rn(qv[d + 2], v.z), t3 = __fsub_rn(qv[d + 3], v.w);
                    acc = __fadd_rn(acc, __fmul_rn(t0, t0)); acc = __fadd_rn(acc, __fmul_rn(t1, t1));
                    acc = __fadd_rn(acc, __fmul_rn(t2, t2)); acc = __fadd_rn(acc, __fmul_rn(t3, t3));
                }
            } else {
                for (int d = 0; d < D; ++d) {
                    const float t = __fsub_rn(qv[d], cp[d]);
                    acc = __fadd_rn(acc, __fmul_rn(t, t));
                }
            }
            const uint32_t kk = __float_as_uint(acc);
            key[0][0] = kk == KEY_MAX ? KEY_MAX - 1 : kk;
        }
        topk_tile<1, 1, PROBE_CAP, PROBE_TRIG>(tk, nprobe, tile, key, pay);
    }
    __syncthreads();
    topk_compact(tk, nprobe);
    const int cnt = tk.cnt[0];
    const int64_t o = (qi * splits + split) * nprobe;
    for (int i = tid; i < nprobe; i += kBlock) {
        part_key[o + i] = i < cnt ? (uint32_t)(tk.buf[0][i] >> 32) : KEY_MAX;
        part_id[o + i] = i < cnt ? (int32_t)(uint32_t)tk.buf[0][i] : -1;
    }
}
__global__ __launch_bounds__(kBlock) void coarse_probe_merge_kernel(const uint32_t *__restrict__ part_key, const int32_t *__restrict__ part_id, int splits, int nprobe,
                                                                    int32_t *__restrict__ probe)
{
    __shared__ TopKShared<1, PROBE_CAP> tk;
    const int64_t qi = blockIdx.x;
    const int tid = threadIdx.x, total = splits * nprobe;
    topk_init(tk);
    __syncthreads();
    int tile = 0;
    for (int base = 0; base < total; base += kBlock, ++tile) {
        const int e = base + tid;
        uint32_t key[1][1] = { { KEY_MAX } };
        uint32_t pay[1] = { 0u };
        if (e < total && part_id[qi * total + e] >= 0) {
            key[0][0] = part_key[qi * total + e];
            pay[0] = (uint32_t)part_id[qi * total + e];
        }
        topk_tile<1, 1, PROBE_CAP, PROBE_TRIG>(tk, nprobe, tile, key, pay);
    }
    __syncthreads();
    topk_compact(tk, nprobe);
    const int cnt = tk.cnt[0];
    for (int i = tid; i < nprobe; i += kBlock) probe[qi * nprobe + i] = i < cnt ? (int32_t)(uint32_t)tk.buf[0][i] : -1;
}

// Batched form (nq >= 64): a workgroup serves 16 queries against 64-centroid tiles staged in LDS -- a centroid row is read
// from memory once per 16 queries, with coalesced 16-byte loads, instead of once per query at a 4 D-byte stride.  Wave w owns
// queries 4 w .. 4 w + 3, lane = centroid of the tile; distances in the reference's order (d ascending, separate multiply and add).
constexpr int CPT_Q = 16;
constexpr int CPT_TILE = 64;
constexpr int CPT_CAP = 320;
constexpr int CPT_TRIG = 192;

__global__ __launch_bounds__(kBlock) void coarse_probe_tile_kernel(const float *__restrict__ q_rot, int64_t nq, int D,
                                                                   const float *__restrict__ coarse, int coarseK, int nprobe,
                                                                   int32_t *__restrict__ probe)
{
    extern __shared__ __attribute__((aligned(16))) float cpt_sm[];
    float *qv = cpt_sm;                       // [CPT_Q][D]
    float *ct = cpt_sm + CPT_Q * D;           // [D][CPT_TILE + 1]
    __shared__ TopKShared<CPT_Q, CPT_CAP> tk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t q0 = (int64_t)blockIdx.x * CPT_Q;
    for (int e = tid; e < CPT_Q * D; e += kBlock) {
        const int64_t qi = q0 + e / D;
        qv[e] = q_rot[(qi < nq ? qi : nq - 1) * D + e % D];
    }
    topk_init(tk);
    __syncthreads();
    int tile = 0;
    for (int base = 0; base < coarseK; base += CPT_TILE, ++tile) {
        // stage the tile transposed: ct[d][c]; consecutive threads read consecutive floats of the centroid block
        const int rows = coarseK - base < CPT_TILE ? coarseK - base : CPT_TILE;
        for (int e = tid; e < CPT_TILE * D; e += kBlock) {
            const int c = e / D, d = e - c * D;
            ct[d * (CPT_TILE + 1) + c] = c < rows ? coarse[(int64_t)(base + c) * D + d] : 0.0f;
        }
        __syncthreads();
        float acc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
        const float *qw = qv + (wave * 4) * D;
        for (int d = 0; d < D; ++d) {
            const float cv = ct[d * (CPT_TILE + 1) + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t = __fsub_rn(qw[j * D + d], cv);  // wave-uniform address: one broadcast read
                acc[j] = __fadd_rn(acc[j], __fmul_rn(t, t));
            }
        }
        uint32_t key[1][CPT_Q];
        uint32_t pay[1] = { (uint32_t)(base + lane) };
#pragma unroll
        for (int q = 0; q < CPT_Q; ++q) key[0][q] = KEY_MAX;
        if (lane < rows) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t kk = __float_as_uint(acc[j]);
                const uint32_t kv = kk == KEY_MAX ? KEY_MAX - 1 : kk;
#pragma unroll
                for (int q = 0; q < CPT_Q; ++q)
                    if (q == wave * 4 + j) key[0][q] = kv;
            }
        }
        topk_tile<CPT_Q, 1, CPT_CAP, CPT_TRIG>(tk, nprobe, tile, key, pay);  // ends with barriers: the tile can be overwritten
    }
    __syncthreads();
    topk_compact<CPT_Q, CPT_CAP>(tk, nprobe);
    for (int e = tid; e < CPT_Q * nprobe; e += kBlock) {
        const int q = e / nprobe, i = e - q * nprobe;
        if (q0 + q < nq) probe[(q0 + q) * nprobe + i] = i < tk.cnt[q] ? (int32_t)(uint32_t)tk.buf[q][i] : -1;
    }
}

static int g_probe_variant = 0;  // cvtmi_set_tuning("probe_variant"): 0 choose, 1 exact kernels only, 2 matrix-core filter wherever it applies
void set_probe_variant(int v) { g_probe_variant = v; }

// scratch the few-queries form needs behind the probe array: per (query, split) nprobe keys + nprobe list ids
constexpr int PROBE_SPLITS_MAX = 64;
// below this many queries the split form runs; from 256 on the matrix-core filter (assign_mfma.hip) applies and is faster (8192 lists:
// 256 frames 0.167 ms against 0.178, 1000 frames 0.21 against 0.56 -- once its score kernel stopped running on nq / 256 workgroups)
constexpr int64_t PROBE_SPLIT_NQ = 256;
size_t coarse_probe_scratch_bytes(int64_t nq, int nprobe)
{
    const size_t b = (size_t)nq * PROBE_SPLITS_MAX * nprobe * 8;
    return nq < PROBE_SPLIT_NQ && b <= ((size_t)64 << 20) ? b : 0;
}

int launch_coarse_probe(const OpqModelDev &m, const float *q_rot, int64_t nq, int nprobe, int32_t *probe, hipStream_t st, void *scratch)
{
    // few queries, many lists: every CU takes a range of the centroids (64 frames x 8192 lists on the 16-queries-per-workgroup kernel
    // below: four workgroups, 2.05 ms)
    if (scratch && g_probe_variant == 0 && m.coarseK >= 1024 && nq > 0 && coarse_probe_scratch_bytes(nq, nprobe) > 0 && nprobe >= 1 && nprobe <= 128) {
        int splits = (m.coarseK + kBlock - 1) / kBlock;
        if (splits > PROBE_SPLITS_MAX) splits = PROBE_SPLITS_MAX;
        const int chunk = ((m.coarseK + splits - 1) / splits + kBlock - 1) / kBlock * kBlock;
        splits = (m.coarseK + chunk - 1) / chunk;
        uint32_t *pk = reinterpret_cast<uint32_t *>(scratch);
        int32_t *pid = reinterpret_cast<int32_t *>(pk + (size_t)nq * splits * nprobe);
        hipLaunchKernelGGL(coarse_probe_split_kernel, dim3((unsigned)splits, (unsigned)nq), dim3(kBlock), (size_t)m.D * sizeof(float), st, q_rot, m.D, m.coarse,
                           m.coarseK, nprobe, chunk, pk, pid);
        hipLaunchKernelGGL(coarse_probe_merge_kernel, dim3((unsigned)nq), dim3(kBlock), 0, st, pk, pid, splits, nprobe, probe);
        CVTMI_HIP(hipGetLastError());
        return CVTMI_OK;
    }
    if (g_probe_variant != 1 && coarse_probe_filter_applies(q_rot, g_probe_variant == 2 ? std::max<int64_t>(nq, 256) : nq, m.D, m.coarse, m.coarseK, nprobe))
        return launch_coarse_probe_filtered(q_rot, nq, m.D, m.coarse, m.coarseK, nprobe, probe, st);
    if (nq >= 64 && nprobe <= 128 && m.D <= 256) {
        const size_t lds = ((size_t)CPT_Q * m.D + (size_t)m.D * (CPT_TILE + 1)) * sizeof(float);
        const int64_t blocks = (nq + CPT_Q - 1) / CPT_Q;
        if (blocks > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "query_video: nq too large");
        CVTMI_HIP(hipFuncSetAttribute((const void *)coarse_probe_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(coarse_probe_tile_kernel, dim3((unsigned)blocks), dim3(kBlock), lds, st, q_rot, nq, m.D, m.coarse, m.coarseK,
                           nprobe, probe);
        CVTMI_HIP(hipGetLastError());
        return CVTMI_OK;
    }
    if (nq <= 0) return CVTMI_OK;
    if (nprobe < 1 || nprobe > 128) return fail(CVTMI_EUNSUPPORTED, "query_video: nprobe=%d outside 1..128", nprobe);
    if (nq > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "query_video: nq too large");
    hipLaunchKernelGGL(coarse_probe_kernel, dim3((unsigned)nq), dim3(kBlock), (size_t)m.D * sizeof(float), st, q_rot, m.D,
                       m.coarse, m.coarseK, nprobe, probe);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

// ------------------------------------------------------------------------------------------
// List-ordered (CSR) copy of the entries, built ON THE DEVICE: a stable counting sort by list id.
//   count    nb single-wave blocks, block b owns the contiguous entry range b; hist[b][l] = its entries of list l
//   prefix   per list: exclusive prefix of hist[.][l] over the blocks (in place) and the list total
//   offsets  exclusive scan of the totals -> list_off[L + 1]; longest list; (min, max) video id seen
//   scatter  the same blocks walk their range in order, 64 entries at a time: lanes holding the same list are found with
//            one ballot per key bit, the first of them claims the group's slots from the block's cursor (hist[b][l]), and
//            every lane copies its entry to  list_off[l] + cursor + rank  -- insertion order inside a list is kept, which is
//            the order m_ivfList holds them in and SaveIndex writes them in (IVFOPQ.cpp:167, :557-575).
// Entries whose list id is outside [0, L) (-1: a row no centroid could claim) are dropped, as before.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void csr_count_kernel(const int32_t *__restrict__ lists, int64_t n, int L, uint32_t *__restrict__ hist)
{
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < n ? r0 + per : n;
    uint32_t *h = hist + (int64_t)blockIdx.x * L;
    for (int64_t i = r0 + threadIdx.x; i < r1; i += 64) {
        const int l = lists ? lists[i] : 0;
        if (l >= 0 && l < L) atomicAdd(&h[l], 1u);
    }
}

__global__ __launch_bounds__(kBlock) void csr_prefix_kernel(uint32_t *__restrict__ hist, int nb, int L, int64_t *__restrict__ total)
{
    const int l = blockIdx.x * kBlock + threadIdx.x;
    if (l >= L) return;
    uint32_t run = 0;
    for (int b = 0; b < nb; ++b) {
        const uint32_t c = hist[(int64_t)b * L + l];
        hist[(int64_t)b * L + l] = run;
        run += c;
    }
    total[l] = run;
}

// one workgroup: list_off[0..L] = exclusive scan of total[0..L); stats[0] = longest list
__global__ __launch_bounds__(1024) void csr_offsets_kernel(const int64_t *__restrict__ total, int L, int64_t *__restrict__ list_off,
                                                           int64_t *__restrict__ stats)
{
    __shared__ int64_t part[1024];
    __shared__ int64_t pmax[1024];
    const int tid = threadIdx.x;
    const int per = (L + 1023) / 1024;
    const int l0 = tid * per, l1 = l0 + per < L ? l0 + per : L;
    int64_t s = 0, mx = 0;
    for (int l = l0; l < l1; ++l) { s += total[l]; mx = total[l] > mx ? total[l] : mx; }
    part[tid] = s; pmax[tid] = mx;
    __syncthreads();
    if (tid == 0) {
        int64_t run = 0, m = 0;
        for (int t = 0; t < 1024; ++t) { const int64_t v = part[t]; part[t] = run; run += v; m = pmax[t] > m ? pmax[t] : m; }
        list_off[L] = run;
        stats[0] = m;
    }
    __syncthreads();
    int64_t run = part[tid];
    for (int l = l0; l < l1; ++l) { list_off[l] = run; run += total[l]; }
}

__global__ __launch_bounds__(64) void csr_scatter_kernel(const int32_t *__restrict__ lists, const int32_t *__restrict__ videos,
                                                         const uint8_t *__restrict__ codes, int64_t n, int L, int M, int key_bits,
                                                         uint32_t *__restrict__ hist, const int64_t *__restrict__ list_off,
                                                         uint8_t *__restrict__ out_codes, int32_t *__restrict__ out_videos,
                                                         int32_t *__restrict__ vstats)
{
    const int lane = threadIdx.x;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < n ? r0 + per : n;
    uint32_t *cur = hist + (int64_t)blockIdx.x * L;
    int vmin = 0x7fffffff, vmax = -0x7fffffff - 1;
    for (int64_t base = r0; base < r1; base += 64) {  // wave-uniform trip count
        const int64_t i = base + lane;
        const bool have = i < r1;
        const int l = have ? (lists ? lists[i] : 0) : -1;
        const bool valid = have && l >= 0 && l < L;
        unsigned long long mask = __ballot(valid);
        for (int b = 0; b < key_bits; ++b) {
            const bool bit = (l >> b) & 1;
            const unsigned long long bal = __ballot(valid && bit);
            mask &= bit ? bal : ~bal;
        }
        uint32_t slot = 0;
        if (valid) {
            const int leader = __ffsll((long long)mask) - 1;
            uint32_t first = 0;
            if (lane == leader) first = atomicAdd(&cur[l], (uint32_t)__popcll(mask));  // this block's cursor inside list l
            first = (uint32_t)__shfl((int)first, leader);
            slot = first + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        }
        if (valid) {
            const int64_t o = list_off[l] + slot;
            if (M == 16) {
                reinterpret_cast<uint4 *>(out_codes)[o] = reinterpret_cast<const uint4 *>(codes)[i];
            } else {
                for (int m = 0; m < M; ++m) out_codes[o * M + m] = codes[i * M + m];
            }
            const int v = videos ? videos[i] : (int32_t)i;
            out_videos[o] = v;
            vmin = v < vmin ? v : vmin;
            vmax = v > vmax ? v : vmax;
        }
    }
    for (int o = 32; o >= 1; o >>= 1) {
        const int a = __shfl_xor(vmin, o), b = __shfl_xor(vmax, o);
        vmin = a < vmin ? a : vmin;
        vmax = b > vmax ? b : vmax;
    }
    if (lane == 0 && vmin <= vmax) { atomicMin(&vstats[0], vmin); atomicMax(&vstats[1], vmax); }
}

size_t csr_scratch_bytes(int64_t n, int L, int *nb_out)
{
    int nb = (int)std::min<int64_t>(512, std::max<int64_t>(1, (8 << 20) / std::max(L, 1)));
    nb = (int)std::min<int64_t>(nb, std::max<int64_t>(1, (n + 4095) / 4096));
    if (nb_out) *nb_out = nb;
    return (((size_t)nb * L * sizeof(uint32_t) + 7) & ~(size_t)7) + (size_t)L * sizeof(int64_t) + 64;   // the int64 part starts on 8 bytes
}

// scratch: csr_scratch_bytes(); list_off [L + 1]; stats_out (device): [0] longest list (int64), then int32 min / max video id at byte 8 / 12
int launch_csr_build(const int32_t *lists, const int32_t *videos, const uint8_t *codes, int64_t n, int L, int M, void *scratch,
                     int64_t *list_off, uint8_t *out_codes, int32_t *out_videos, void *stats_out, hipStream_t st)
{
    int nb = 1;
    const size_t sb = csr_scratch_bytes(n, L, &nb);
    uint32_t *hist = static_cast<uint32_t *>(scratch);
    int64_t *total = reinterpret_cast<int64_t *>(static_cast<char *>(scratch) + (((size_t)nb * L * sizeof(uint32_t) + 7) & ~(size_t)7));
    (void)sb;
    CVTMI_HIP(hipMemsetAsync(hist, 0, (size_t)nb * L * sizeof(uint32_t), st));
    int64_t *st64 = static_cast<int64_t *>(stats_out);
    int32_t *vst = reinterpret_cast<int32_t *>(st64 + 1);
    CVTMI_HIP(hipMemsetAsync(st64, 0, 8, st));
    CVTMI_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(vst), 0x7fffffff, 1, st));        // min video id
    CVTMI_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(vst + 1), (int)0x80000000u, 1, st));   // max video id
    if (n > 0) {
        hipLaunchKernelGGL(csr_count_kernel, dim3(nb), dim3(64), 0, st, lists, n, L, hist);
        CVTMI_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(csr_prefix_kernel, dim3((L + kBlock - 1) / kBlock), dim3(kBlock), 0, st, hist, nb, L, total);
    hipLaunchKernelGGL(csr_offsets_kernel, dim3(1), dim3(1024), 0, st, total, L, list_off, st64);
    CVTMI_HIP(hipGetLastError());
    if (n > 0) {
        int key_bits = 1;
        while ((1 << key_bits) < L) ++key_bits;
        hipLaunchKernelGGL(csr_scatter_kernel, dim3(nb), dim3(64), 0, st, lists, videos, codes, n, L, M, key_bits, hist, list_off,
                           out_codes, out_videos, vst);
        CVTMI_HIP(hipGetLastError());
    }
    return CVTMI_OK;
}

// ------------------------------------------------------------------------------------------
// Per-video query: one workgroup per (query, probed list, piece of that list)
// ------------------------------------------------------------------------------------------
__global__ void fill_u32_kernel(uint32_t *p, int64_t n, uint32_t v)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) p[i] = v;
}

// The table of (query, list) is built in LDS as in the reference (:273-291); a list longer than rows_per_piece is cut into
// pieces that each rebuild the table (98 KFLOP against 16 look-ups per row: amortised from a few hundred rows up), so the work of
// a probe is proportional to its list and no workgroup walks a long list alone.  Pieces past the end of their list exit at once.
__global__ __launch_bounds__(kBlock) void query_video_kernel(const float *__restrict__ q_rot, int D, int M, int K,
                                                             int step, const float *__restrict__ coarse,
                                                             const float *__restrict__ books, int nprobe,
                                                             const int32_t *__restrict__ probe,
                                                             const int64_t *__restrict__ list_off,
                                                             const uint8_t *__restrict__ codes,
                                                             const int32_t *__restrict__ video_id, int img_num,
                                                             float *match_score, int pieces, int rows_per_piece)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];  // res[D] + lut[M][256]
    float *res = sm;
    float *lut = sm + D;
    const int64_t pair = blockIdx.x / pieces;
    const int piece = (int)(blockIdx.x - pair * pieces);
    const int64_t qi = pair / nprobe;
    const int l = probe[pair];
    if (l < 0) return;  // workgroup-uniform
    const int64_t b = list_off[l] + (int64_t)piece * rows_per_piece;
    int64_t e = b + rows_per_piece;
    e = e < list_off[l + 1] ? e : list_off[l + 1];
    if (b >= e) return;  // workgroup-uniform: an empty list, or a piece past its end
    const int tid = threadIdx.x;
    for (int d = tid; d < D; d += kBlock) res[d] = __fsub_rn(q_rot[qi * D + d], coarse[(int64_t)l * D + d]);
    __syncthreads();
    for (int t = tid; t < M * 256; t += kBlock) {
        const int m = t >> 8, j = t & 255;
        float acc = __uint_as_float(0x7f800000u);
        if (j < K) {
            const float *c = books + ((int64_t)m * K + j) * step;
            acc = 0.0f;
            for (int kk = 0; kk < step; ++kk) {
                const float d = __fsub_rn(res[m * step + kk], c[kk]);
                acc = __fadd_rn(acc, __fmul_rn(d, d));
            }
        }
        lut[t] = acc;
    }
    __syncthreads();
    uint32_t *ms = reinterpret_cast<uint32_t *>(match_score) + qi * img_num;
    for (int64_t r = b + tid; r < e; r += kBlock) {
        const uint8_t *c = codes + r * M;
        float s = 0.0f;
        if (M == 16) {  // one 16-byte load per code row
            const uint4 v = *reinterpret_cast<const uint4 *>(c);
            const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int m = 0; m < 16; ++m) s = __fadd_rn(s, lut[m * 256 + ((w[m >> 2] >> (8 * (m & 3))) & 0xffu)]);
        } else {
            for (int m = 0; m < M; ++m) s = __fadd_rn(s, lut[m * 256 + c[m]]);
        }
        const int v = video_id[r];
        if (v >= 0 && v < img_num) atomicMin(&ms[v], __float_as_uint(s));
    }
}

int launch_query_video(const OpqModelDev &m, const float *q_rot, int64_t nq, int nprobe, const int32_t *probe,
                       const int64_t *list_off, const uint8_t *codes, const int32_t *video_id, int img_num,
                       float *match_score, int64_t longest_list, hipStream_t st)
{
    if (nq <= 0 || img_num <= 0) return CVTMI_OK;
    if (m.K > 256) return fail(CVTMI_EUNSUPPORTED, "query_video: K=%d > 256", m.K);
    const int64_t total = nq * img_num;
    int64_t fb = (total + kBlock - 1) / kBlock;
    if (fb > 4096) fb = 4096;
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)fb), dim3(kBlock), 0, st,
                       reinterpret_cast<uint32_t *>(match_score), total, 0x3f800000u /* 1.0f = threhold */);
    if (longest_list <= 0) return CVTMI_OK;
    const int rows_per_piece = 4096;
    const int64_t pieces = (longest_list + rows_per_piece - 1) / rows_per_piece;
    const int64_t blocks = nq * nprobe * pieces;
    if (blocks > 0x7fffffff || pieces > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "query_video: grid too large");
    const size_t lds = ((size_t)m.D + (size_t)m.M * 256) * sizeof(float);
    hipLaunchKernelGGL(query_video_kernel, dim3((unsigned)blocks), dim3(kBlock), lds, st, q_rot, m.D, m.M, m.K, m.step,
                       m.coarse, m.books, nprobe, probe, list_off, codes, video_id, img_num, match_score, (int)pieces, rows_per_piece);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

}  // namespace cvtmi
