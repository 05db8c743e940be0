/*
 * cvt_oracle.c -- CPU restatement of the cvt OPQ-encode / ADC-search hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under cvt_amd/ (the product) may link, import or
 * call this file.  It is used by tests/, by __graft_entry__.smoke() and by bench.py's
 * `cpu_baseline` leg, always as the checker / the reported CPU baseline, never as the
 * thing that is shipped or measured as the GPU path.
 *
 * Every function restates, on flat arrays, the arithmetic of the reference lines it
 * cites (paths relative to /root/reference).  Operation ORDER is part of the contract:
 * the reference accumulates `tmp = a - b; acc += tmp * tmp` left to right in fp32 with
 * separate multiply and add, so this file must be compiled with -ffp-contract=off and
 * without -ffast-math (oracle/Makefile does that).
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - OPQ reorder / encode / LUT+ADC / video-min aggregation / top-k, and the three
 *     brute-force metrics, are pinned bit-for-bit against the reference's own sources
 *     compiled in place (oracle/_ref, recipe in oracle/Makefile) and against the golden
 *     vectors under tests/golden/ that were generated from those binaries.
 *   - Scalar quantisation (orc_sq8_*): the L2 normalisation (orc_sq8_l2norm; int8_quan.cc:46-56 ==
 *     utils/math_util.h:29-39) IS PINNED: the reference's own MathUtil::L2NormArray is compiled in place
 *     (oracle/_ref/libref_math.so) and its outputs are golden data (tests/golden/sq8_norm_golden.npz).
 *     The quantise / decode formulas (int8_quan.cc:72-94, :117-132) and training (faiss RS_minmax) stay
 *     RESTATED ONLY: the reference delegates those to faiss 1.5.3 (not vendored, not installable here)
 *     and holds no expected outputs; the functions below follow the in-tree formulas line by line.
 *   - orc_rotate_fma: the general d x d rotation is NOT in the reference (which only
 *     permutes); it is the specification of the MFMA GEMM kernel (k-ordered fmaf chain).
 *   - orc_pca_project: PARITY UNPINNED.  The reference calls cv::PCA::project (OpenCV 3.2 / 3.3,
 *     not vendored, not installed here) and holds no expected outputs; flavour 0 restates OpenCV's
 *     own gemm (double accumulators), flavour 1 is the specification of the MFMA kernel.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * "OPQ rotation" = dimension permutation.  opq/src/IVFOPQ.cpp:424-439 (reorder),
 * applied to every row by LoadSingleFeatFile :459-461.   y[i] = x[perm[i]].
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_reorder(const int32_t *perm, int D, const float *x, int64_t n, float *y)
{
    for (int64_t r = 0; r < n; ++r) {
        const float *src = x + r * D;
        float *dst = y + r * D;
        for (int i = 0; i < D; ++i) dst[i] = src[perm[i]];
    }
}

/* General rotation Y = X * R^T as the GPU kernel computes it: one rounding per product,
 * products folded in ascending k by fused multiply-add starting from +0
 * (MI355X f32 MFMA == k-ordered fmaf chain).  With R a 0/1 permutation matrix this is
 * exactly orc_reorder for finite inputs. */
ORC_API void orc_rotate_fma(const float *R, int D, const float *x, int64_t n, float *y)
{
    for (int64_t r = 0; r < n; ++r) {
        const float *src = x + r * D;
        for (int i = 0; i < D; ++i) {
            float acc = 0.0f;
            const float *row = R + (int64_t)i * D;
            for (int k = 0; k < D; ++k) acc = fmaf(row[k], src[k], acc);
            y[r * D + i] = acc;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * PCA projection + L2 normalisation.  cvtk::PCAUtils::reduceDim,
 * pca_train_project/pca_online/pca_utils.cc:25-35 (twin: project/pca_dimension.h:47-58):
 *     reduceMat = pca_.project(mat)                      -- OpenCV: (x - mean) * vectors^T
 *     per row: norm = row * row.t() (a 1 x d gemm), denomv = float(max(1e-12, (double)sqrt(norm))),
 *              row[j] /= denomv
 * cv::PCA::project (modules/core/src/pca.cpp) subtracts the mean in the type of `mean` (fp32 here)
 * and calls gemm(tmp, eigenvectors, 1, Mat(), 0, result, GEMM_2_T).  OpenCV's built-in gemm for
 * CV_32F keeps DOUBLE accumulators and walks k in ascending order (GEMMSingleMul / GEMMBlockMul
 * <float, double>, modules/core/src/matmul.cpp); a build with LAPACK/IPP calls sgemm instead.
 *   flavour 0: the built-in gemm: y[j] = float(sum_k double(t[k]) * double(E[j][k])), t = x - mean in fp32
 *   flavour 1: what the MI355X kernel is specified to compute: the k-ascending fmaf chain in fp32
 * The row norm is the same 1 x d gemm: double accumulation, one rounding to float.
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_pca_project(const float *mean, const float *vectors, int din, int dout, const float *x, int64_t n,
                             int l2norm, int flavour, float *y)
{
    float *t = (float *)malloc((size_t)din * sizeof(float));
    for (int64_t r = 0; r < n; ++r) {
        const float *src = x + r * din;
        float *dst = y + r * dout;
        for (int k = 0; k < din; ++k) t[k] = src[k] - mean[k];
        for (int j = 0; j < dout; ++j) {
            const float *e = vectors + (int64_t)j * din;
            if (flavour == 0) {
                double acc = 0.0;
                for (int k = 0; k < din; ++k) acc += (double)t[k] * (double)e[k];
                dst[j] = (float)acc;
            } else {
                float acc = 0.0f;
                for (int k = 0; k < din; ++k) acc = fmaf(t[k], e[k], acc);
                dst[j] = acc;
            }
        }
        if (l2norm) {
            double ss = 0.0;
            for (int j = 0; j < dout; ++j) ss += (double)dst[j] * (double)dst[j];
            const float norm = (float)ss;
            const double root = (double)sqrtf(norm);
            const float denomv = (float)(root > 1e-12 ? root : 1e-12);
            for (int j = 0; j < dout; ++j) dst[j] = dst[j] / denomv;
        }
    }
    free(t);
}

/* squared L2 distance, sequential fp32, no contraction.  IVFOPQ.cpp:117-122 / :147-154 */
static inline float sq_dist_seq(const float *a, const float *b, int n)
{
    float acc = 0.0f;
    for (int k = 0; k < n; ++k) {
        float t = a[k] - b[k];
        acc += t * t;
    }
    return acc;
}

/* Coarse assignment of Add(): IVFOPQ.cpp:110-129.  dismin starts at float(UINT_MAX),
 * strict '<' keeps the first minimum, vw stays -1 if nothing beats the start value. */
ORC_API void orc_coarse_assign(const float *x, int64_t n, int D, const float *coarse, int coarseK,
                               int32_t *out_vw)
{
    for (int64_t r = 0; r < n; ++r) {
        float best = (float)4294967295u; /* UINT_MAX -> 4294967296.0f */
        int32_t vw = -1;
        for (int i = 0; i < coarseK; ++i) {
            float d = sq_dist_seq(x + r * D, coarse + (int64_t)i * D, D);
            if (d < best) { best = d; vw = i; }
        }
        out_vw[r] = vw;
    }
}

/* PQ encode of Add(): residual :135-139, per-sub-quantiser argmin :141-161.
 * books layout = LoadModel's: [M][K][step] contiguous (:88-93).
 * A row whose coarse assignment is -1 (all-NaN row) is undefined behaviour in the
 * reference (m_ppCoarseCluster[-1]); here its residual is taken against list 0 and the
 * list id is reported as -1. */
ORC_API void orc_pq_encode(const float *x, int64_t n, int D, const float *coarse, int coarseK,
                           const float *books, int M, int K, int32_t *out_list, uint8_t *out_codes)
{
    const int step = D / M;
    float *res = (float *)malloc(sizeof(float) * (size_t)D);
    for (int64_t r = 0; r < n; ++r) {
        int32_t vw;
        orc_coarse_assign(x + r * D, 1, D, coarse, coarseK, &vw);
        const float *cen = coarse + (int64_t)(vw < 0 ? 0 : vw) * D;
        for (int i = 0; i < D; ++i) res[i] = x[r * D + i] - cen[i];
        for (int m = 0; m < M; ++m) {
            float best = (float)4294967295u;
            int bj = -1;
            const float *cb = books + (int64_t)m * K * step;
            for (int j = 0; j < K; ++j) {
                float d = sq_dist_seq(res + m * step, cb + (int64_t)j * step, step);
                if (d < best) { best = d; bj = j; }
            }
            out_codes[r * M + m] = (uint8_t)bj; /* -1 -> 255, IVFOPQ.cpp:161 */
        }
        if (out_list) out_list[r] = vw;
    }
    free(res);
}

/* Distance look-up table of Query(): IVFOPQ.cpp:273-291.  centroid may be NULL (= zeros
 * is NOT the same as skipping the subtraction for -0.0 inputs, so NULL means "subtract
 * an explicit +0.0f", which is what a zero coarse centroid does in the reference). */
ORC_API void orc_lut(const float *q, int D, const float *centroid, const float *books, int M, int K,
                     float *lut /* [M][K] */)
{
    const int step = D / M;
    float *res = (float *)malloc(sizeof(float) * (size_t)D);
    for (int i = 0; i < D; ++i) res[i] = q[i] - (centroid ? centroid[i] : 0.0f);
    for (int m = 0; m < M; ++m)
        for (int j = 0; j < K; ++j)
            lut[m * K + j] = sq_dist_seq(res + m * step, books + ((int64_t)m * K + j) * step, step);
    free(res);
}

/* ADC scan: IVFOPQ.cpp:300-306.  score = sum_{k<M} LUT[k][code[k]], fp32, from 0.0f. */
ORC_API void orc_adc_scan(const float *lut, int M, int K, const uint8_t *codes, int64_t n,
                          float *out_scores)
{
    for (int64_t r = 0; r < n; ++r) {
        float s = 0.0f;
        for (int k = 0; k < M; ++k) s += lut[k * K + codes[r * M + k]];
        out_scores[r] = s;
    }
}

/* ---- (dist,id) lexicographic ordering: std::pair<float,uint> operator<, used by
 * get_sort_results (opq/src/common.h:25-37) and by the max-heap in
 * brutoforce.hpp:73-93. ---- */
typedef struct { float d; int64_t id; } orc_pair;

static int pair_less(const orc_pair *a, const orc_pair *b)
{
    if (a->d < b->d) return 1;
    if (b->d < a->d) return 0;
    return a->id < b->id;
}
static int pair_cmp_qsort(const void *pa, const void *pb)
{
    const orc_pair *a = (const orc_pair *)pa, *b = (const orc_pair *)pb;
    if (pair_less(a, b)) return -1;
    if (pair_less(b, a)) return 1;
    return 0;
}

/* k smallest (score, index) pairs ascending == std::partial_sort_copy of common.h:34.
 * Writes min(k, n) entries, returns that count. */
ORC_API int64_t orc_topk_pairs(const float *scores, const int64_t *ids /* NULL -> 0..n-1 */,
                               int64_t n, int64_t k, float *out_d, int64_t *out_id)
{
    orc_pair *p = (orc_pair *)malloc(sizeof(orc_pair) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) { p[i].d = scores[i]; p[i].id = ids ? ids[i] : i; }
    qsort(p, (size_t)n, sizeof(orc_pair), pair_cmp_qsort);
    int64_t m = k < n ? k : n;
    for (int64_t i = 0; i < m; ++i) { out_d[i] = p[i].d; out_id[i] = p[i].id; }
    free(p);
    return m;
}

/* Bounded running selection used by the exhaustive-search oracles below: same result as
 * the reference's heap (k smallest lexicographic pairs) without materialising n pairs. */
typedef struct { orc_pair *h; int64_t k, size; } orc_heap; /* max-heap on pair_less */

static void heap_sift_up(orc_heap *hp, int64_t i)
{
    while (i > 0) {
        int64_t par = (i - 1) / 2;
        if (pair_less(&hp->h[par], &hp->h[i])) {
            orc_pair t = hp->h[par]; hp->h[par] = hp->h[i]; hp->h[i] = t; i = par;
        } else break;
    }
}
static void heap_sift_down(orc_heap *hp, int64_t i)
{
    for (;;) {
        int64_t l = 2 * i + 1, r = l + 1, big = i;
        if (l < hp->size && pair_less(&hp->h[big], &hp->h[l])) big = l;
        if (r < hp->size && pair_less(&hp->h[big], &hp->h[r])) big = r;
        if (big == i) break;
        orc_pair t = hp->h[big]; hp->h[big] = hp->h[i]; hp->h[i] = t; i = big;
    }
}
/* brutoforce.hpp:74-92: the first k rows are pushed unconditionally; afterwards a row is
 * pushed when dist <= current worst and the worst is popped when the heap exceeds k. */
static void heap_offer(orc_heap *hp, float d, int64_t id)
{
    if (hp->size < hp->k) {
        hp->h[hp->size].d = d; hp->h[hp->size].id = id; hp->size++;
        heap_sift_up(hp, hp->size - 1);
        return;
    }
    if (!(d <= hp->h[0].d)) return;
    orc_pair cand = { d, id };
    if (pair_less(&cand, &hp->h[0])) { hp->h[0] = cand; heap_sift_down(hp, 0); }
    /* else: pushed then immediately popped again as the new maximum -> no change */
}
static int64_t heap_drain_sorted(orc_heap *hp, float *out_d, int64_t *out_id)
{
    int64_t m = hp->size;
    qsort(hp->h, (size_t)m, sizeof(orc_pair), pair_cmp_qsort);
    for (int64_t i = 0; i < m; ++i) { out_d[i] = hp->h[i].d; out_id[i] = hp->h[i].id; }
    return m;
}

/* Exhaustive ADC top-k (north-star form of Query: coarseK = 1, nprobe = 1, one vector
 * per "video", no 1.0 clamp): LUT per query (IVFOPQ.cpp:279-291), scan (:300-306),
 * k smallest (score,id) (common.h:25-37).  q must already be rotated.  Output rows are
 * padded with (+inf, -1) when n < k. */
ORC_API void orc_adc_search(const float *q, int64_t nq, int D, const float *centroid,
                            const float *books, int M, int K, const uint8_t *codes, int64_t n,
                            int64_t id_base, int64_t k, float *out_d, int64_t *out_id)
{
    float *lut = (float *)malloc(sizeof(float) * (size_t)M * K);
    orc_heap hp; hp.h = (orc_pair *)malloc(sizeof(orc_pair) * (size_t)(k > 0 ? k : 1)); hp.k = k;
    for (int64_t qi = 0; qi < nq; ++qi) {
        orc_lut(q + qi * D, D, centroid, books, M, K, lut);
        hp.size = 0;
        for (int64_t r = 0; r < n; ++r) {
            float s = 0.0f;
            const uint8_t *c = codes + r * M;
            for (int kk = 0; kk < M; ++kk) s += lut[kk * K + c[kk]];
            heap_offer(&hp, s, id_base + r);
        }
        int64_t m = heap_drain_sorted(&hp, out_d + qi * k, out_id + qi * k);
        for (int64_t i = m; i < k; ++i) { out_d[qi * k + i] = INFINITY; out_id[qi * k + i] = -1; }
    }
    free(hp.h); free(lut);
}

/* Full reference Query()/QueryThrehold() semantics, IVFOPQ.cpp:232-315 / :341-417.
 *   - coarse scan keeps the nk nearest lists in a max-heap of (dist, i): the first nk
 *     lists are pushed unconditionally, later ones replace the top when strictly closer
 *     (:248-259);  lists are then visited in heap-pop order = farthest first (:264-267).
 *   - matchScore[f][video] starts at 1.0 (threhold, :5, :262) and takes the min over
 *     every code of that video met in the probed lists (:308).
 * Inverted lists are given CSR-style: list l owns entries [list_off[l], list_off[l+1]).
 * q must already be rotated. */
ORC_API void orc_query_video(const float *q, int64_t nq, int D, const float *coarse, int coarseK,
                             const float *books, int M, int K, int nk, const int64_t *list_off,
                             const uint8_t *codes, const int32_t *video_id, int img_num,
                             float *match_score /* [nq][img_num] */)
{
    float *lut = (float *)malloc(sizeof(float) * (size_t)M * K);
    orc_pair *hp = (orc_pair *)malloc(sizeof(orc_pair) * (size_t)(nk > 0 ? nk : 1));
    for (int64_t f = 0; f < nq; ++f) {
        orc_heap h; h.h = hp; h.k = nk; h.size = 0;
        for (int i = 0; i < coarseK; ++i) {
            float d = sq_dist_seq(q + f * D, coarse + (int64_t)i * D, D);
            if (i < nk) {
                hp[h.size].d = d; hp[h.size].id = i; h.size++; heap_sift_up(&h, h.size - 1);
            } else if (d < hp[0].d) {
                hp[0].d = d; hp[0].id = i; heap_sift_down(&h, 0);
            }
        }
        float *ms = match_score + f * img_num;
        for (int v = 0; v < img_num; ++v) ms[v] = 1.0f;
        while (h.size > 0) {
            int vw = (int)hp[0].id;
            hp[0] = hp[h.size - 1]; h.size--; heap_sift_down(&h, 0);
            orc_lut(q + f * D, D, coarse + (int64_t)vw * D, books, M, K, lut);
            for (int64_t j = list_off[vw]; j < list_off[vw + 1]; ++j) {
                float s = 0.0f;
                for (int kk = 0; kk < M; ++kk) s += lut[kk * K + codes[j * M + kk]];
                int v = video_id[j];
                if (s < ms[v]) ms[v] = s; /* min(score, current), :308 */
            }
        }
    }
    free(hp); free(lut);
}

/* Frame aggregation of the query main: multi_frame_index_test.cpp:60-67, then
 * get_sort_results (:68).  total[v] = sum_f matchScore[f][v] in frame order. */
ORC_API void orc_video_rank(const float *match_score, int64_t nq, int img_num, int64_t k,
                            float *total /* [img_num] */, float *out_d, int64_t *out_id)
{
    for (int v = 0; v < img_num; ++v) total[v] = 0.0f;
    for (int64_t f = 0; f < nq; ++f)
        for (int v = 0; v < img_num; ++v) total[v] += match_score[f * img_num + v];
    orc_topk_pairs(total, NULL, img_num, k, out_d, out_id);
}

/* ------------------------------------------------------------------------------------------
 * Exhaustive-search distance functions.
 * ---------------------------------------------------------------------------------------- */
/* brute_force_search/src/space_ip.hpp:25-34  (scalar path) */
static float ip_scalar(const float *a, const float *b, int n)
{
    float res = 0;
    for (int i = 0; i < n; ++i) res += a[i] * b[i];
    return 1.0f - res;
}
/* space_ip.hpp:84-131 / :168-206 (non-AVX branches of InnerProductSIMD4Ext / SIMD16Ext, what the
 * reference's own CMake flags build): ONE 4-lane accumulator walks the row 4 floats at a time
 * (16-blocks first, then 4-blocks: same accumulator, same order), lanes are then added left to
 * right.  `lanes` = 8 models the AVX branch of SIMD16Ext (:140-167) for D % 16 == 0. */
static float ip_lanes(const float *a, const float *b, int n, int lanes)
{
    float acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    int nl = (n / lanes) * lanes;
    for (int i = 0; i < nl; i += lanes)
        for (int l = 0; l < lanes; ++l) acc[l] += a[i + l] * b[i + l];
    float s = acc[0];
    for (int l = 1; l < lanes; ++l) s += acc[l];
    return 1.0f - s;
}
/* hnsw_sifts_retrieval/hnswlib/space_l2.h:26-37 */
static float l2_scalar(const float *a, const float *b, int n) { return sq_dist_seq(a, b, n); }
/* space_l2.h:40-73 (L2SqrSIMD16Ext, AVX branch hard-enabled by `#define USE_AVX` at :12): 8 lanes;
 * space_l2.h:123-151 (L2SqrSIMD4Ext): 4 lanes.  Lanes are added left to right. */
static float l2_lanes(const float *a, const float *b, int n, int lanes)
{
    float acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    int nl = (n / lanes) * lanes;
    for (int i = 0; i < nl; i += lanes)
        for (int l = 0; l < lanes; ++l) { float t = a[i + l] - b[i + l]; acc[l] += t * t; }
    float s = acc[0];
    for (int l = 1; l < lanes; ++l) s += acc[l];
    return s;
}
/* space_l2.h:186-219: integer L2 over groups of four uint8; a dim%4 tail is dropped. */
static int l2_u8(const uint8_t *a, const uint8_t *b, int n)
{
    int res = 0;
    int groups = n >> 2;
    for (int i = 0; i < groups * 4; ++i) { int t = (int)a[i] - (int)b[i]; res += t * t; }
    return res;
}

/* metric ids shared with include/cvtmi.h */
enum { ORC_IP = 0, ORC_L2F = 1, ORC_L2U8 = 2 };
/* flavour 0 forces the scalar loops; any other value applies the reference's own function
 * selection (InnerProductSpace ctor space_ip.hpp:217-225, L2Space ctor space_l2.h:159-164):
 *   IP : D%4==0 -> 4-lane SSE order (flavour 8 and D%16==0 -> the 8-lane AVX order instead)
 *   L2F: D%16==0 -> 8-lane AVX order, else D%4==0 -> 4-lane, else scalar */
ORC_API float orc_dist(int metric, int flavour, const void *a, const void *b, int D)
{
    if (metric == ORC_IP) {
        if (flavour == 0 || D % 4 != 0) return ip_scalar((const float *)a, (const float *)b, D);
        if (flavour == 8 && D % 16 == 0) return ip_lanes((const float *)a, (const float *)b, D, 8);
        return ip_lanes((const float *)a, (const float *)b, D, 4);
    }
    if (metric == ORC_L2F) {
        if (flavour == 0 || D % 4 != 0) return l2_scalar((const float *)a, (const float *)b, D);
        return l2_lanes((const float *)a, (const float *)b, D, D % 16 == 0 ? 8 : 4);
    }
    return (float)l2_u8((const uint8_t *)a, (const uint8_t *)b, D);
}

/* BruteforceSearch::searchKnn, brutoforce.hpp:73-93: k smallest (dist,label) pairs.
 * Integer distances are returned as float (exact below 2^24; D*255^2 <= 2^24 for D<=258,
 * so out_di carries the exact int for the uint8 metric). */
ORC_API void orc_flat_search(int metric, int flavour, int D, const void *data, const int64_t *labels,
                             int64_t n, const void *queries, int64_t nq, int64_t k, float *out_d,
                             int32_t *out_di, int64_t *out_label)
{
    const size_t row = (metric == ORC_L2U8) ? (size_t)D : (size_t)D * 4;
    orc_heap hp; hp.h = (orc_pair *)malloc(sizeof(orc_pair) * (size_t)(k > 0 ? k : 1)); hp.k = k;
    for (int64_t qi = 0; qi < nq; ++qi) {
        const char *qp = (const char *)queries + qi * row;
        hp.size = 0;
        for (int64_t r = 0; r < n; ++r) {
            const char *xp = (const char *)data + r * row;
            float d;
            if (metric == ORC_L2U8) d = (float)l2_u8((const uint8_t *)qp, (const uint8_t *)xp, D);
            else d = orc_dist(metric, flavour, qp, xp, D);
            heap_offer(&hp, d, labels ? labels[r] : r);
        }
        int64_t m = heap_drain_sorted(&hp, out_d + qi * k, out_label + qi * k);
        for (int64_t i = m; i < k; ++i) { out_d[qi * k + i] = INFINITY; out_label[qi * k + i] = -1; }
        if (out_di) {
            /* exact integer distances recomputed for the winners */
            /* fewer than k rows: the reference's queue simply ends; the C ABI pads a fixed-size row with label -1 and the +inf bit
             * pattern in the 4-byte distance field (include/cvtmi.h), for integer distances too */
            for (int64_t i = 0; i < k; ++i) out_di[qi * k + i] = 0x7f800000;
            if (metric == ORC_L2U8) {
                for (int64_t i = 0; i < m; ++i) {
                    int64_t lab = out_label[qi * k + i], rr = lab;
                    if (labels) { for (rr = 0; rr < n && labels[rr] != lab; ++rr) {} }
                    out_di[qi * k + i] = l2_u8((const uint8_t *)qp, (const uint8_t *)data + rr * row, D);
                }
            }
        }
    }
    free(hp.h);
}

/* ------------------------------------------------------------------------------------------
 * Scalar quantisation (normalisation pinned, quantise / decode / train restated -- see header).
 * ---------------------------------------------------------------------------------------- */
/* scalar_quantization/scalar_quantization/int8_quan.cc:46-56 (== utils/math_util.h:29-39):
 * float product, double accumulate, double sqrt, float(max(1e-12, norm)), float divide. */
ORC_API void orc_sq8_l2norm(float *v, int d)
{
    double accum = 0.0;
    for (int i = 0; i < d; ++i) accum += v[i] * v[i];
    accum = sqrt(accum);
    float den = (float)(accum > 1e-12 ? accum : 1e-12);
    for (int i = 0; i < d; ++i) v[i] = v[i] / den;
}

/* Int8Encode, int8_quan.cc:72-94, applied to each of n vectors (the reference encodes only
 * the first; callers pass n = 1 for the literal behaviour).  x is normalised IN PLACE when
 * l2norm != 0, as the reference does to its caller's buffer. */
ORC_API void orc_sq8_encode(const float *vmin, const float *vdiff, int d, float *x, int64_t n,
                            int l2norm, uint8_t *out)
{
    for (int64_t r = 0; r < n; ++r) {
        float *row = x + r * d;
        if (l2norm) orc_sq8_l2norm(row, d);
        for (int i = 0; i < d; ++i) {
            float xi = 0;
            if (vdiff[i] != 0) xi = (row[i] - vmin[i]) / vdiff[i];
            if (xi < 0) xi = 0;
            if (xi > 1.0) xi = 1.0;
            out[r * d + i] = (uint8_t)(int)(255 * xi);
        }
    }
}

/* Int8Decode(std::string&), int8_quan.cc:117-132: evaluated in double, stored as float. */
ORC_API void orc_sq8_decode(const float *vmin, const float *vdiff, int d, const uint8_t *codes,
                            int64_t n, float *out)
{
    for (int64_t r = 0; r < n; ++r)
        for (int i = 0; i < d; ++i)
            out[r * d + i] = (float)(vmin[i] + vdiff[i] * (codes[r * d + i] + 0.5) / 255.0);
}

/* Int8Decode(uint8_t*) / Int8DecodeFaiss, int8_quan.cc:96-115: faiss::ScalarQuantizer::decode.  faiss is not in /root/reference
 * (external dependency, 1.5.3 per the reference's build notes); its QT_8bit codec as published: Codec8bit::decode_component =
 * (code[i] + 0.5f) / 255.0f, QuantizerTemplate<Codec, false>::reconstruct_component = vmin[i] + xi * vdiff[i], all float.
 * volatile keeps the product and the sum two roundings whatever -ffp-contract says.  PARITY UNPINNED (no faiss to run). */
ORC_API void orc_sq8_decode_faiss(const float *vmin, const float *vdiff, int d, const uint8_t *codes,
                                  int64_t n, float *out)
{
    for (int64_t r = 0; r < n; ++r)
        for (int i = 0; i < d; ++i) {
            const float xi = (codes[r * d + i] + 0.5f) / 255.0f;
            volatile float prod = xi * vdiff[i];
            out[r * d + i] = vmin[i] + prod;
        }
}

/* sq_train.cpp:84-103: rows are L2-normalised (:84) and handed to
 * faiss::IndexScalarQuantizer(d, QT_8bit).train, whose default range statistic (RS_minmax,
 * rangestat_arg 0) is per-dimension min and max-min.  x is normalised in place. */
ORC_API void orc_sq8_train(float *x, int64_t n, int d, int l2norm, float *vmin, float *vdiff)
{
    for (int i = 0; i < d; ++i) { vmin[i] = HUGE_VALF; vdiff[i] = -HUGE_VALF; }
    for (int64_t r = 0; r < n; ++r) {
        float *row = x + r * d;
        if (l2norm) orc_sq8_l2norm(row, d);
        for (int i = 0; i < d; ++i) {
            if (row[i] < vmin[i]) vmin[i] = row[i];
            if (row[i] > vdiff[i]) vdiff[i] = row[i];
        }
    }
    for (int i = 0; i < d; ++i) vdiff[i] = vdiff[i] - vmin[i];
}

/* ------------------------------------------------------------------------------------------
 * Multi-way merge of per-shard top-k lists (the exchange step of SURVEY 8e; same shape as
 * FLANN-MPI's ResultsMerger, retrieval/vlindex/lib/FLANN/mpi/index.h:74-108).
 * in_d/in_id: [nq][L][k]; entries with id < 0 are padding.  Output [nq][k].
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_merge_topk(const float *in_d, const int64_t *in_id, int64_t nq, int64_t L, int64_t k,
                            float *out_d, int64_t *out_id)
{
    orc_pair *p = (orc_pair *)malloc(sizeof(orc_pair) * (size_t)(L * k > 0 ? L * k : 1));
    for (int64_t qi = 0; qi < nq; ++qi) {
        int64_t m = 0;
        for (int64_t i = 0; i < L * k; ++i) {
            int64_t id = in_id[qi * L * k + i];
            if (id < 0) continue;
            p[m].d = in_d[qi * L * k + i]; p[m].id = id; ++m;
        }
        qsort(p, (size_t)m, sizeof(orc_pair), pair_cmp_qsort);
        for (int64_t i = 0; i < k; ++i) {
            if (i < m) { out_d[qi * k + i] = p[i].d; out_id[qi * k + i] = p[i].id; }
            else { out_d[qi * k + i] = INFINITY; out_id[qi * k + i] = -1; }
        }
    }
    free(p);
}

/* ------------------------------------------------------------------------------------------
 * Codebook training: TrainPQ::CoarseQuan / ProdQuan (opq/train_codebook/train_PQ_codebook.cpp:150-244).
 * The reference delegates to yael's kmeans(d, n, k, niter = 0, v, nt, seed = 1, redo = 1, ...), a library
 * that is not vendored: PARITY UNPINNED.  What is restated here is the structure the reference fixes --
 * coarse k-means on the (permuted) vectors, residuals x - coarse[assign] (:190-197), one k-means per
 * sub-space on the residual columns (:214-233) -- around a plain, fully specified Lloyd iteration:
 *   init      k distinct rows drawn with splitmix64(seed) (index = next() % n, redraw on repeats);
 *   assign    nearest centroid, sequential fp32 distance, strict '<' keeps the first minimum
 *             (the arithmetic of IVFOPQ::Add, IVFOPQ.cpp:110-129);
 *   update    centroid = float(sum_double(members, ascending row order) / count); an empty cluster keeps
 *             its centroid;
 *   stop      when an assignment pass changes nothing, or after max_iter updates (niter = 0 -> 100);
 *             assignments returned are those of the last pass, made against the returned centroids
 *             whenever the loop ended on convergence.
 * x has leading dimension ld (floats per row), d <= ld columns are used.
 * ---------------------------------------------------------------------------------------- */
static uint64_t orc_splitmix64(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

ORC_API int orc_kmeans(const float *x, int64_t ld, int64_t n, int d, int k, int niter, uint64_t seed,
                       float *cent /* [k][d] */, int32_t *assign /* [n] */, int *iters_done)
{
    if (n < k || k < 1 || d < 1) return -1;
    const int max_iter = niter > 0 ? niter : 100;
    unsigned char *taken = (unsigned char *)calloc((size_t)n, 1);
    uint64_t st = seed;
    for (int c = 0; c < k; ++c) {
        int64_t r;
        do { r = (int64_t)(orc_splitmix64(&st) % (uint64_t)n); } while (taken[r]);
        taken[r] = 1;
        memcpy(cent + (int64_t)c * d, x + r * ld, sizeof(float) * (size_t)d);
    }
    free(taken);
    double *sum = (double *)malloc(sizeof(double) * (size_t)k * d);
    int64_t *cnt = (int64_t *)malloc(sizeof(int64_t) * (size_t)k);
    for (int64_t r = 0; r < n; ++r) assign[r] = -2;
    int it = 0;
    for (;;) {
        int64_t changed = 0;
        for (int64_t r = 0; r < n; ++r) {
            float best = (float)4294967295u;
            int32_t bi = -1;
            for (int c = 0; c < k; ++c) {
                float dd = sq_dist_seq(x + r * ld, cent + (int64_t)c * d, d);
                if (dd < best) { best = dd; bi = c; }
            }
            if (bi != assign[r]) { ++changed; assign[r] = bi; }
        }
        if (changed == 0 || it >= max_iter) break;
        memset(sum, 0, sizeof(double) * (size_t)k * d);
        memset(cnt, 0, sizeof(int64_t) * (size_t)k);
        for (int64_t r = 0; r < n; ++r) {
            const int32_t c = assign[r];
            if (c < 0) continue; /* a row no centroid claims (NaN) */
            for (int i = 0; i < d; ++i) sum[(int64_t)c * d + i] += (double)x[r * ld + i];
            ++cnt[c];
        }
        for (int c = 0; c < k; ++c)
            if (cnt[c] > 0)
                for (int i = 0; i < d; ++i) cent[(int64_t)c * d + i] = (float)(sum[(int64_t)c * d + i] / (double)cnt[c]);
        ++it;
    }
    if (iters_done) *iters_done = it;
    free(sum); free(cnt);
    return 0;
}

/* TrainPQ::IFVPQ (:144-148): x is already permuted (LoadFeatureSample :77-82).  coarse [coarseK][D],
 * books [M][K][D/M] (the SaveCodebook layout, :283-287).  Every k-means uses the same seed, as the
 * reference passes seed = 1 to each. */
ORC_API int orc_opq_train(const float *x, int64_t n, int D, int coarseK, int M, int K, int niter, uint64_t seed,
                          float *coarse, float *books)
{
    const int step = D / M;
    int32_t *assign = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    float *res = (float *)malloc(sizeof(float) * (size_t)n * D);
    int rc = orc_kmeans(x, D, n, D, coarseK, niter, seed, coarse, assign, NULL);
    if (rc == 0) {
        for (int64_t r = 0; r < n; ++r) {
            const int32_t c = assign[r] < 0 ? 0 : assign[r];
            for (int i = 0; i < D; ++i) res[r * D + i] = x[r * D + i] - coarse[(int64_t)c * D + i];
        }
        for (int m = 0; m < M && rc == 0; ++m)
            rc = orc_kmeans(res + m * step, D, n, step, K, niter, seed, books + (int64_t)m * K * step, assign, NULL);
    }
    free(assign); free(res);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * OPQ rotation learning ("optional").  NOT in the reference: opq/ only ever permutes
 * dimensions (reorder_, IVFOPQ.cpp:424-439) and takes that permutation from a file.  What is specified
 * here is the non-parametric alternation of Ge et al. ("Optimized Product Quantization", 2013) over
 * pieces that exist on the path already -- the dense rotation (orc_rotate_fma), the fully specified
 * Lloyd iteration (orc_kmeans) -- plus an orthogonal Procrustes step:
 *     R = I;  repeat `outer` times:
 *         Xr = X R^T;  per sub-space m: (books[m], assign[m]) = kmeans(Xr[:, m], K, niter, seed)
 *         Y[r] = the concatenation of books[m][assign[m][r]]            (the rows' reconstructions)
 *         C = X^T Y  in double, rows in blocks of 1024: inside a block ascending, then the blocks
 *             ascending (the order the device kernel can reproduce: one workgroup per block)
 *         C = U S V^T (one-sided Jacobi, double);  R = V U^T            (argmin_R |X R^T - Y|_F, R orthogonal)
 *     books = kmeans of the final Xr = X R^T.
 * A Procrustes step that meets a rank-deficient C keeps the previous R.  Self-specified: the GPU path
 * (cvtmi_opq_learn_rotation) is held to it bit for bit.
 * ---------------------------------------------------------------------------------------- */
#define ORC_XTY_BLOCK 1024
ORC_API void orc_xty(const float *x, const float *y, int64_t n, int D, double *C /* [D][D] */)
{
    double *part = (double *)malloc(sizeof(double) * (size_t)D * D);
    for (int i = 0; i < D * D; ++i) C[i] = 0.0;
    for (int64_t b0 = 0; b0 < n; b0 += ORC_XTY_BLOCK) {
        const int64_t b1 = b0 + ORC_XTY_BLOCK < n ? b0 + ORC_XTY_BLOCK : n;
        for (int i = 0; i < D * D; ++i) part[i] = 0.0;
        for (int64_t r = b0; r < b1; ++r)
            for (int i = 0; i < D; ++i) {
                const double xi = (double)x[r * D + i];
                for (int j = 0; j < D; ++j) part[i * D + j] += xi * (double)y[r * D + j];   /* the product of two floats is exact in double */
            }
        for (int i = 0; i < D * D; ++i) C[i] += part[i];
    }
    free(part);
}

/* R = V U^T for C = U S V^T.  One-sided Jacobi on the columns of A = C (cyclic sweeps over the pairs p < q, a pair is rotated
 * when |a_p . a_q| > 1e-15 sqrt(|a_p|^2 |a_q|^2); at most 60 sweeps), V accumulates the rotations, U = the normalised columns.
 * Returns 1 (R untouched) when C is zero or holds a value that is not finite. */
ORC_API int orc_procrustes(const double *C, int D, float *R)
{
    double *A = (double *)malloc(sizeof(double) * (size_t)D * D), *V = (double *)malloc(sizeof(double) * (size_t)D * D);
    int rc = 0;
    for (int i = 0; i < D * D; ++i) { A[i] = C[i]; if (!(C[i] == C[i]) || C[i] - C[i] != 0.0) rc = 1; }
    for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) V[i * D + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60 && rc == 0; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < D - 1; ++p)
            for (int q = p + 1; q < D; ++q) {
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
                for (int i = 0; i < D; ++i) {
                    const double ap = A[i * D + p], aq = A[i * D + q];
                    alpha += ap * ap; beta += aq * aq; gamma += ap * aq;
                }
                if (!(fabs(gamma) > 1e-15 * sqrt(alpha * beta))) continue;
                rotated = 1;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int i = 0; i < D; ++i) {
                    const double ap = A[i * D + p], aq = A[i * D + q];
                    A[i * D + p] = c * ap - sn * aq; A[i * D + q] = sn * ap + c * aq;
                    const double vp = V[i * D + p], vq = V[i * D + q];
                    V[i * D + p] = c * vp - sn * vq; V[i * D + q] = sn * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    double smax = 0.0;
    if (rc == 0) {
        for (int j = 0; j < D; ++j) {
            double s2 = 0.0;
            for (int i = 0; i < D; ++i) s2 += A[i * D + j] * A[i * D + j];
            const double sj = sqrt(s2);
            if (!(sj == sj) || sj - sj != 0.0) { rc = 1; break; }
            if (sj > smax) smax = sj;
        }
    }
    if (rc == 0 && !(smax > 0.0)) rc = 1;   /* C = 0: nothing to align */
    if (rc == 0) {
        /* U: the columns with a length are normalised; a rank-deficient C (fewer distinct reconstructions than dimensions) leaves
         * columns without one -- any orthonormal completion maximises tr(R C) equally, so they are completed deterministically:
         * the unit vector e_k with the least of its length inside the span of the columns already final (first minimum),
         * two rounds of Gram-Schmidt against them, normalised */
        unsigned char *fin = (unsigned char *)calloc((size_t)D, 1);
        for (int j = 0; j < D; ++j) {
            double s2 = 0.0;
            for (int i = 0; i < D; ++i) s2 += A[i * D + j] * A[i * D + j];
            const double sj = sqrt(s2);
            if (sj > 1e-12 * smax) {
                for (int i = 0; i < D; ++i) A[i * D + j] = A[i * D + j] / sj;
                fin[j] = 1;
            }
        }
        for (int j = 0; j < D && rc == 0; ++j) {
            if (fin[j]) continue;
            int kbest = 0;               /* the unit vector with the least of its length inside the span of the final columns */
            double ebest = 0.0;
            for (int k = 0; k < D; ++k) {
                double e2 = 0.0;
                for (int c = 0; c < D; ++c)
                    if (fin[c]) e2 += A[k * D + c] * A[k * D + c];
                if (k == 0 || e2 < ebest) { ebest = e2; kbest = k; }
            }
            for (int i = 0; i < D; ++i) A[i * D + j] = i == kbest ? 1.0 : 0.0;
            for (int round = 0; round < 2; ++round)
                for (int c = 0; c < D; ++c) {
                    if (!fin[c]) continue;
                    double dot = 0.0;
                    for (int i = 0; i < D; ++i) dot += A[i * D + j] * A[i * D + c];
                    for (int i = 0; i < D; ++i) A[i * D + j] = A[i * D + j] - dot * A[i * D + c];
                }
            double s2 = 0.0;
            for (int i = 0; i < D; ++i) s2 += A[i * D + j] * A[i * D + j];
            const double sj = sqrt(s2);
            if (!(sj > 1e-8)) { rc = 1; break; }
            for (int i = 0; i < D; ++i) A[i * D + j] = A[i * D + j] / sj;
            fin[j] = 1;
        }
        free(fin);
    }
    if (rc == 0)
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) {   /* R[i][j] = sum_k V[i][k] U[j][k] */
                double acc = 0.0;
                for (int k = 0; k < D; ++k) acc += V[i * D + k] * A[j * D + k];
                R[i * D + j] = (float)acc;
            }
    free(A); free(V);
    return rc;
}

ORC_API int orc_opq_learn_rotation(const float *x, int64_t n, int D, int M, int K, int outer, int niter, uint64_t seed,
                                   float *R /* [D][D] out */, float *books /* [M][K][D/M] out */)
{
    if (n < K || D < 1 || M < 1 || D % M != 0 || outer < 0) return -1;
    const int step = D / M;
    float *xr = (float *)malloc(sizeof(float) * (size_t)n * D), *y = (float *)malloc(sizeof(float) * (size_t)n * D);
    int32_t *assign = (int32_t *)malloc(sizeof(int32_t) * (size_t)n * M);
    double *C = (double *)malloc(sizeof(double) * (size_t)D * D);
    int rc = 0;
    for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) R[i * D + j] = i == j ? 1.0f : 0.0f;
    for (int t = 0; t <= outer && rc == 0; ++t) {
        orc_rotate_fma(R, D, x, n, xr);
        for (int m = 0; m < M && rc == 0; ++m)
            rc = orc_kmeans(xr + m * step, D, n, step, K, niter, seed, books + (int64_t)m * K * step, assign + (int64_t)m * n, NULL);
        if (rc != 0 || t == outer) break;
        for (int64_t r = 0; r < n; ++r)
            for (int m = 0; m < M; ++m) {
                const int32_t a = assign[(int64_t)m * n + r];
                for (int j = 0; j < step; ++j) y[r * D + m * step + j] = a < 0 ? 0.0f : books[((int64_t)m * K + a) * step + j];
            }
        orc_xty(x, y, n, D, C);
        (void)orc_procrustes(C, D, R);   /* a degenerate step keeps R */
    }
    free(xr); free(y); free(assign); free(C);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * HNSW search over a graph saved by the reference: HierarchicalNSW::searchKnn
 * (hnsw_sifts_retrieval/hnswlib/hnswalg.h:688-729) = greedy descent through the upper levels (:692-712),
 * then searchBaseLayerST (:217-280) with ef = max(ef_, k), then the k best.
 * File layout = saveIndex (:491-519): 96-byte header (size_t offsetLevel0, max_elements, cur_element_count,
 * size_data_per_element, label_offset, offsetData; int maxlevel; unsigned enterpoint; size_t maxM, maxM0, M;
 * double mult; size_t ef_construction), level-0 block [max_elements][size_data_per_element] =
 * {unsigned count; unsigned link[maxM0]; vector bytes; size_t label}, then per element
 * {unsigned linkListSize; level 1.. link lists of (maxM + 1) unsigned each}.
 *
 * Both queues are std::priority_queue with CompareByFirst (:78-83): ties between equal distances are
 * decided by the binary-heap mechanics, so libstdc++'s push_heap / pop_heap are restated literally
 * (bits/stl_heap.h: __push_heap, __adjust_heap) -- a different but valid heap would pick different
 * duplicates.
 * ---------------------------------------------------------------------------------------- */
typedef struct { float d; uint32_t id; } hn_ent;
typedef struct { hn_ent *a; int64_t n, cap; } hn_heap;  /* max-heap on d (CompareByFirst) */

static void hn_push_heap(hn_ent *first, int64_t hole, int64_t top, hn_ent v)
{
    int64_t parent = (hole - 1) / 2;
    while (hole > top && first[parent].d < v.d) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = v;
}
static void hn_push(hn_heap *h, float d, uint32_t id)
{
    if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 64; h->a = (hn_ent *)realloc(h->a, sizeof(hn_ent) * (size_t)h->cap); }
    hn_ent v; v.d = d; v.id = id;
    h->a[h->n] = v;
    h->n++;
    hn_push_heap(h->a, h->n - 1, 0, v);
}
static void hn_pop(hn_heap *h)
{
    if (h->n > 1) {
        hn_ent *first = h->a;
        const int64_t len = h->n - 1;
        hn_ent value = first[len];
        first[len] = first[0];
        int64_t hole = 0, second = 0;
        while (second < (len - 1) / 2) {
            second = 2 * (second + 1);
            if (first[second].d < first[second - 1].d) second--;
            first[hole] = first[second];
            hole = second;
        }
        if ((len & 1) == 0 && second == (len - 2) / 2) {
            second = 2 * (second + 1);
            first[hole] = first[second - 1];
            hole = second - 1;
        }
        hn_push_heap(first, hole, 0, value);
    }
    h->n--;
}

typedef struct {
    uint64_t offsetLevel0, max_elements, cur_count, size_per_elem, label_offset, offsetData;
    int32_t maxlevel; uint32_t enterpoint;
    uint64_t maxM, maxM0, M; double mult; uint64_t efc;
    const uint8_t *level0;
    const uint8_t **lists; /* per element: pointer to its upper-level link block or NULL */
} hn_index;

static int hn_parse(const uint8_t *f, int64_t bytes, hn_index *ix)
{
    if (bytes < 96) return -1;
    const uint8_t *p = f;
    memcpy(&ix->offsetLevel0, p, 8); p += 8; memcpy(&ix->max_elements, p, 8); p += 8;
    memcpy(&ix->cur_count, p, 8); p += 8; memcpy(&ix->size_per_elem, p, 8); p += 8;
    memcpy(&ix->label_offset, p, 8); p += 8; memcpy(&ix->offsetData, p, 8); p += 8;
    memcpy(&ix->maxlevel, p, 4); p += 4; memcpy(&ix->enterpoint, p, 4); p += 4;
    memcpy(&ix->maxM, p, 8); p += 8; memcpy(&ix->maxM0, p, 8); p += 8; memcpy(&ix->M, p, 8); p += 8;
    memcpy(&ix->mult, p, 8); p += 8; memcpy(&ix->efc, p, 8); p += 8;
    ix->level0 = p;
    p += ix->max_elements * ix->size_per_elem;
    ix->lists = (const uint8_t **)calloc((size_t)ix->max_elements, sizeof(uint8_t *));
    for (uint64_t i = 0; i < ix->max_elements; ++i) {
        if (p + 4 > f + bytes) return -1;
        uint32_t sz; memcpy(&sz, p, 4); p += 4;
        if (sz) { ix->lists[i] = p; p += sz; }
    }
    return p <= f + bytes ? 0 : -1;
}

typedef float (*hn_dist_fn)(void *ctx, const float *q, uint32_t id);

/* the traversal of searchKnn with a pluggable node distance; output ascending (dist, label), padded (0, -1) */
static void hn_search_core(const hn_index *ixp, hn_dist_fn dist, void *ctx, int qdim, const float *queries, int64_t nq,
                           int64_t k, int64_t ef_param, float *out_d, int64_t *out_label)
{
    const hn_index ix = *ixp;
    const uint64_t links_per = ix.maxM * 4 + 4;
    uint8_t *visited = (uint8_t *)malloc((size_t)ix.max_elements);
    hn_heap top = { NULL, 0, 0 }, cand = { NULL, 0, 0 };
    orc_pair *res = (orc_pair *)malloc(sizeof(orc_pair) * (size_t)(k > 0 ? k : 1));
    for (int64_t qi = 0; qi < nq; ++qi) {
        const float *q = queries + qi * qdim;
        for (int64_t i = 0; i < k; ++i) { out_d[qi * k + i] = 0.0f; out_label[qi * k + i] = -1; }
        if (ix.cur_count == 0) continue;
        uint32_t cur = ix.enterpoint;
        float curdist = dist(ctx, q, cur);
        for (int level = ix.maxlevel; level > 0; --level) {
            int changed = 1;
            while (changed) {
                changed = 0;
                const uint8_t *ll = ix.lists[cur] + (uint64_t)(level - 1) * links_per;
                uint32_t size; memcpy(&size, ll, 4);
                for (uint32_t i = 0; i < size; ++i) {
                    uint32_t c; memcpy(&c, ll + 4 + 4 * (uint64_t)i, 4);
                    const float d = dist(ctx, q, c);
                    if (d < curdist) { curdist = d; cur = c; changed = 1; }
                }
            }
        }
        const int64_t ef = ef_param > k ? ef_param : k;
        memset(visited, 0, (size_t)ix.max_elements);
        top.n = 0; cand.n = 0;
        float d0 = dist(ctx, q, cur);
        hn_push(&top, d0, cur);
        hn_push(&cand, -d0, cur);
        visited[cur] = 1;
        float lower = d0;
        while (cand.n) {
            const hn_ent c = cand.a[0];
            if (-c.d > lower) break;
            hn_pop(&cand);
            const uint8_t *ll = ix.level0 + (uint64_t)c.id * ix.size_per_elem + ix.offsetLevel0;
            uint32_t size; memcpy(&size, ll, 4);
            for (uint32_t j = 1; j <= size; ++j) {
                uint32_t nb; memcpy(&nb, ll + 4 * (uint64_t)j, 4);
                if (visited[nb]) continue;
                visited[nb] = 1;
                const float d = dist(ctx, q, nb);
                if (top.a[0].d > d || top.n < ef) {
                    hn_push(&cand, -d, nb);
                    hn_push(&top, d, nb);
                    if (top.n > ef) hn_pop(&top);
                    lower = top.a[0].d;
                }
            }
        }
        while (top.n > k) hn_pop(&top);
        int64_t m = 0;
        while (top.n > 0) {
            uint64_t lab; memcpy(&lab, ix.level0 + (uint64_t)top.a[0].id * ix.size_per_elem + ix.label_offset, 8);
            res[m].d = top.a[0].d; res[m].id = (int64_t)lab; ++m;
            hn_pop(&top);
        }
        qsort(res, (size_t)m, sizeof(orc_pair), pair_cmp_qsort);  /* the (dist, label) order of `results` (:719-726) */
        for (int64_t i = 0; i < m; ++i) { out_d[qi * k + i] = res[i].d; out_label[qi * k + i] = res[i].id; }
    }
    free(visited); free(top.a); free(cand.a); free(res);
}

typedef struct { const hn_index *ix; int metric, D; } hn_vec_ctx;
static float hn_vec_dist(void *c, const float *q, uint32_t id)
{
    const hn_vec_ctx *x = (const hn_vec_ctx *)c;
    return orc_dist(x->metric, 4, q, x->ix->level0 + (uint64_t)id * x->ix->size_per_elem + x->ix->offsetData, x->D);
}

/* metric: ORC_IP / ORC_L2F with the distance functions the reference's Space classes select for D
 * (InnerProductSpace space_ip.h:213-222: SSE 4-lane order for D % 4 == 0 -- the AVX branches are '#if 0' in
 * this tree --; L2Space space_l2.h:159-164). */
ORC_API int orc_hnsw_search(const uint8_t *file, int64_t bytes, int metric, int D, const float *queries, int64_t nq,
                            int64_t k, int64_t ef_param, float *out_d, int64_t *out_label)
{
    hn_index ix;
    if (hn_parse(file, bytes, &ix) != 0) return -1;
    hn_vec_ctx c = { &ix, metric, D };
    hn_search_core(&ix, hn_vec_dist, &c, D, queries, nq, k, ef_param, out_d, out_label);
    free((void *)ix.lists);
    return 0;
}

/* "HNSW over OPQ-compressed vectors" (BASELINE config 5; not in the reference): the same traversal, node
 * distance = ADC sum of the query's table over the node's PQ code, tables and sum as in IVFOPQ::Query
 * (IVFOPQ.cpp:273-291, :302-306) with a zero coarse centroid.  q_rot: rotated queries [nq][D];
 * codes [n][M] in internal-id order. */
typedef struct { const float *books; const uint8_t *codes; int D, M, K; float *lut; const float *lut_for; } hn_adc_ctx;
static float hn_adc_dist(void *c, const float *q, uint32_t id)
{
    hn_adc_ctx *x = (hn_adc_ctx *)c;
    if (x->lut_for != q) { orc_lut(q, x->D, NULL, x->books, x->M, x->K, x->lut); x->lut_for = q; }
    float s = 0.0f;
    for (int m = 0; m < x->M; ++m) s += x->lut[m * x->K + x->codes[(int64_t)id * x->M + m]];
    return s;
}
ORC_API int orc_hnsw_search_adc(const uint8_t *file, int64_t bytes, int D, const float *books, int M, int K,
                                const uint8_t *codes, const float *q_rot, int64_t nq, int64_t k, int64_t ef_param,
                                float *out_d, int64_t *out_label)
{
    hn_index ix;
    if (hn_parse(file, bytes, &ix) != 0) return -1;
    hn_adc_ctx c = { books, codes, D, M, K, (float *)malloc(sizeof(float) * (size_t)M * K), NULL };
    hn_search_core(&ix, hn_adc_dist, &c, D, q_rot, nq, k, ef_param, out_d, out_label);
    free(c.lut); free((void *)ix.lists);
    return 0;
}

/* OpenMP-free multi-thread helper is deliberately absent: bench.py's cpu_baseline times the
 * single-thread loop (cores = 1) exactly as the reference runs it. */
