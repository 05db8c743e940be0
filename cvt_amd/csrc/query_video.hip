// query_video.hip -- the reference's own query semantics, IVFOPQ::Query / QueryThrehold
// (opq/src/IVFOPQ.cpp:213-320 / :322-422):
//   * per query frame keep the nk nearest coarse lists (:238-260: the first nk lists enter a
//     max-heap unconditionally, later ones replace the top when strictly closer -> the nk smallest
//     (distance, list) pairs; the visiting order does not matter because the scores are min-reduced);
//   * per probed list: residual LUT (:273-291) and ADC scan of that list's entries (:300-306);
//   * matchScore[f][videoId] = min(score, current), starting from threhold = 1.0 (:5, :262, :308).
// Scores are sums of squares (>= +0), so the fp32 min is an unsigned-integer atomicMin on the bits.
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

int launch_coarse_probe(const OpqModelDev &m, const float *q_rot, int64_t nq, int nprobe, int32_t *probe, hipStream_t st)
{
    if (nq <= 0) return CVTMI_OK;
    if (nprobe < 1 || nprobe > 128) return fail(CVTMI_EUNSUPPORTED, "query_video: nprobe=%d outside 1..128", nprobe);
    if (nq > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "query_video: nq too large");
    hipLaunchKernelGGL(coarse_probe_kernel, dim3((unsigned)nq), dim3(kBlock), (size_t)m.D * sizeof(float), st, q_rot, m.D,
                       m.coarse, m.coarseK, nprobe, probe);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

__global__ void fill_u32_kernel(uint32_t *p, int64_t n, uint32_t v)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) p[i] = v;
}

__global__ __launch_bounds__(kBlock) void query_video_kernel(const float *__restrict__ q_rot, int D, int M, int K,
                                                             int step, const float *__restrict__ coarse,
                                                             const float *__restrict__ books, int nprobe,
                                                             const int32_t *__restrict__ probe,
                                                             const int64_t *__restrict__ list_off,
                                                             const uint8_t *__restrict__ codes,
                                                             const int32_t *__restrict__ video_id, int img_num,
                                                             float *match_score)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];  // res[D] + lut[M][256]
    float *res = sm;
    float *lut = sm + D;
    const int64_t qi = blockIdx.x / nprobe;
    const int l = probe[blockIdx.x];
    if (l < 0) return;  // workgroup-uniform
    const int tid = threadIdx.x;
    for (int d = tid; d < D; d += kBlock) res[d] = __fsub_rn(q_rot[qi * D + d], coarse[(int64_t)l * D + d]);
    __syncthreads();
    for (int e = tid; e < M * 256; e += kBlock) {
        const int m = e >> 8, j = e & 255;
        float acc = __uint_as_float(0x7f800000u);
        if (j < K) {
            const float *c = books + ((int64_t)m * K + j) * step;
            acc = 0.0f;
            for (int kk = 0; kk < step; ++kk) {
                const float t = __fsub_rn(res[m * step + kk], c[kk]);
                acc = __fadd_rn(acc, __fmul_rn(t, t));
            }
        }
        lut[e] = acc;
    }
    __syncthreads();
    uint32_t *ms = reinterpret_cast<uint32_t *>(match_score) + qi * img_num;
    const int64_t b = list_off[l], e = list_off[l + 1];
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
                       float *match_score, hipStream_t st)
{
    if (nq <= 0 || img_num <= 0) return CVTMI_OK;
    if (m.K > 256) return fail(CVTMI_EUNSUPPORTED, "query_video: K=%d > 256", m.K);
    const int64_t total = nq * img_num;
    int64_t fb = (total + kBlock - 1) / kBlock;
    if (fb > 4096) fb = 4096;
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)fb), dim3(kBlock), 0, st,
                       reinterpret_cast<uint32_t *>(match_score), total, 0x3f800000u /* 1.0f = threhold */);
    const int64_t blocks = nq * nprobe;
    if (blocks > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "query_video: grid too large");
    const size_t lds = ((size_t)m.D + (size_t)m.M * 256) * sizeof(float);
    hipLaunchKernelGGL(query_video_kernel, dim3((unsigned)blocks), dim3(kBlock), lds, st, q_rot, m.D, m.M, m.K, m.step,
                       m.coarse, m.books, nprobe, probe, list_off, codes, video_id, img_num, match_score);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

}  // namespace cvtmi
