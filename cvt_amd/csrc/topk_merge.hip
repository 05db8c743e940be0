// topk_merge.hip -- merge L sorted (distance, id) lists per query into the k smallest pairs.
// Used twice: across the row splits of one GPU's scan, and across GPUs after the RCCL all-gather
// of per-shard results (same shape as FLANN-MPI's ResultsMerger,
// retrieval/vlindex/lib/FLANN/mpi/index.h:74-108).
//
// One workgroup per query streams the L*k candidates through the shared selection buffer
// (block_topk.h).  The payload is the candidate's position in the input; lists are ordered by
// ascending id range and each list is (distance, id)-sorted, so position order equals id order among
// equal distances -- exactly the tie rule the selection needs.
#include "block_topk.h"
#include "kernels.h"

namespace cvtmi {

constexpr int MERGE_CAP = 1024;
constexpr int MERGE_TRIG = 768;
constexpr int MERGE_R = 4;

// GATHERED = false: candidate i of query q sits at in_d[q * n_cand + i].
// GATHERED = true : the all-gather layout of the row-sharded search -- list r (one per rank, k entries) of query q sits at
//                   in_d[r * stride_d + q * k + j] / in_id[r * stride_id + q * k + j], i = r * k + j.
template <bool GATHERED, int CAP = MERGE_CAP, int TRIG = MERGE_TRIG>
__global__ __launch_bounds__(kBlock) void topk_merge_kernel(const float *__restrict__ in_d,
                                                            const int64_t *__restrict__ in_id, int n_cand, int k,
                                                            float *__restrict__ out_d, int64_t *__restrict__ out_id,
                                                            int64_t stride_d, int64_t stride_id, const uint32_t *__restrict__ only_if)
{
    __shared__ TopKShared<1, CAP> tk;
    const int64_t q = blockIdx.x;
    if (only_if && only_if[q] == 0) return;   // (workgroup-uniform)
    const int tid = threadIdx.x;
    auto at = [&](int i) -> int64_t {  // element offset of candidate i inside the distance / id arrays (before the stride term)
        if (!GATHERED) return q * n_cand + i;
        return q * k + (i % k);
    };
    auto dist_of = [&](int i) -> float { return GATHERED ? in_d[(int64_t)(i / k) * stride_d + at(i)] : in_d[at(i)]; };
    auto id_of = [&](int i) -> int64_t { return GATHERED ? in_id[(int64_t)(i / k) * stride_id + at(i)] : in_id[at(i)]; };
    topk_init(tk);
    __syncthreads();
    int tile = 0;
    for (int base = 0; base < n_cand; base += kBlock * MERGE_R, ++tile) {
        uint32_t key[MERGE_R][1];
        uint32_t pay[MERGE_R];
#pragma unroll
        for (int r = 0; r < MERGE_R; ++r) {
            const int i = base + r * kBlock + tid;
            pay[r] = (uint32_t)i;
            key[r][0] = KEY_MAX;
            if (i < n_cand && (in_id == nullptr || id_of(i) >= 0)) {
                const uint32_t kk = f32_key(dist_of(i));
                key[r][0] = kk == KEY_MAX ? KEY_MAX - 1 : kk;  // keep the "not a candidate" code free
            }
        }
        topk_tile<1, MERGE_R, CAP, TRIG>(tk, k, tile, key, pay);
    }
    __syncthreads();
    topk_compact(tk, k);
    const int cnt = tk.cnt[0];
    for (int i = tid; i < k; i += kBlock) {
        if (i < cnt) {
            const uint32_t p = (uint32_t)tk.buf[0][i];
            out_d[q * k + i] = dist_of((int)p);
            out_id[q * k + i] = in_id ? id_of((int)p) : (int64_t)p;
        } else {
            out_d[q * k + i] = __uint_as_float(0x7f800000u);
            out_id[q * k + i] = -1;
        }
    }
}

// merge of the all-gathered per-rank lists (shard.hip): L lists of k per query, list r at base + r * stride
int launch_topk_merge_gathered(const float *in_d, const int64_t *in_id, int64_t stride_d, int64_t stride_id, int64_t nq, int L, int k,
                               float *out_d, int64_t *out_id, hipStream_t st)
{
    if (nq <= 0) return CVTMI_OK;
    if (k < 1 || k > kBigK) return fail(CVTMI_EUNSUPPORTED, "topk: k=%d outside 1..%d", k, kBigK);
    if (L < 1 || (int64_t)L * k > 0x7fffffff || nq > 0x7fffffff) return fail(CVTMI_EINVAL, "topk_merge: bad list count %d", L);
    if (k > MERGE_TRIG / 2)
        hipLaunchKernelGGL((topk_merge_kernel<true, kBigCap, kBigTrig>), dim3((unsigned)nq), dim3(kBlock), 0, st, in_d, in_id, L * k, k, out_d, out_id,
                           stride_d, stride_id, (const uint32_t *)nullptr);
    else
        hipLaunchKernelGGL(topk_merge_kernel<true>, dim3((unsigned)nq), dim3(kBlock), 0, st, in_d, in_id, L * k, k, out_d, out_id, stride_d,
                           stride_id, (const uint32_t *)nullptr);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

int launch_topk_merge(const float *in_d, const int64_t *in_id, int64_t nq, int L, int k, float *out_d, int64_t *out_id,
                      hipStream_t st, const uint32_t *only_if)
{
    if (L < 1 || (int64_t)L * k > 0x7fffffff) return fail(CVTMI_EINVAL, "topk_merge: bad list count %d", L);
    return launch_topk_select(in_d, in_id, nq, (int64_t)L * k, k, out_d, out_id, st, only_if);
}

// k smallest (value, id) of n_cand candidates per query; in_id == nullptr: id = position
int launch_topk_select(const float *in_d, const int64_t *in_id, int64_t nq, int64_t n_cand, int k, float *out_d,
                       int64_t *out_id, hipStream_t st, const uint32_t *only_if)
{
    if (nq <= 0) return CVTMI_OK;
    if (k < 1 || k > kBigK) return fail(CVTMI_EUNSUPPORTED, "topk: k=%d outside 1..%d", k, kBigK);
    if (n_cand < 0 || n_cand > 0x7fffffff) return fail(CVTMI_EINVAL, "topk: bad candidate count");
    if (nq > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "topk: nq too large");
    if (k > MERGE_TRIG / 2)
        hipLaunchKernelGGL((topk_merge_kernel<false, kBigCap, kBigTrig>), dim3((unsigned)nq), dim3(kBlock), 0, st, in_d, in_id, (int)n_cand, k, out_d,
                           out_id, (int64_t)0, (int64_t)0, only_if);
    else
        hipLaunchKernelGGL(topk_merge_kernel<false>, dim3((unsigned)nq), dim3(kBlock), 0, st, in_d, in_id, (int)n_cand, k, out_d, out_id,
                           (int64_t)0, (int64_t)0, only_if);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

}  // namespace cvtmi
