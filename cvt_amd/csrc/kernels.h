// kernels.h -- internal launch interface between the C-ABI glue (api.hip) and the HIP kernels.
// All pointers are device pointers; every launcher is asynchronous on `st`.
#pragma once
#include <vector>

#include "common.h"

namespace cvtmi {

// k of a search: up to 128 through every kernel; 129 .. kBigK through the exact kernels only, one query per workgroup with a
// 4096-entry selection buffer (block_topk.h is generic in k: the buffer has to hold k entries below its compaction trigger)
constexpr int kBigK = CVTMI_K_MAX, kBigCap = 4096, kBigTrig = 3072;
static_assert(kBigK <= kBigTrig && 2 * kBigK <= kBigCap, "selection buffer too small for kBigK");

struct OpqModelDev {
    int D, coarseK, M, K, step;
    const float *coarse;  // [coarseK][D]
    const float *books;   // [M][K][step]
    const float *R;       // [D][D] or null
    const int32_t *perm;  // [D] or null
};

// ---- rotate.hip ----
int launch_permute(const int32_t *perm, int D, const float *x, int64_t n, float *y, hipStream_t st);
int launch_rotate_gemm(const float *R, int D, const float *x, int64_t n, float *y, hipStream_t st);

// ---- opq_encode.hip ----
int launch_coarse_assign(const OpqModelDev &m, const float *x_rot, int64_t n, int32_t *list_id, hipStream_t st);
// list_id may be null (=> list 0 for every row)
// variant: 0 = choose, 1 = VALU kernel, 2 = matrix-core filter + exact resolution
// single_list_out (coarseK == 1 only, and only when pq_encode_fuses_lists() says so): the encode kernel writes the list
// assignment (0, or -1 for rows no centroid can claim) itself, sparing the separate pass over the rows
int launch_pq_encode(const OpqModelDev &m, const float *x_rot, int64_t n, const int32_t *list_id, uint8_t *codes,
                     hipStream_t st, int variant = 0, int32_t *single_list_out = nullptr, const int32_t *perm = nullptr);
// perm != nullptr: x_rot holds the rows BEFORE the model's permutation and the kernel gathers through it (needs pq_encode_takes_perm)
bool pq_encode_takes_perm(const OpqModelDev &m, const float *x, int64_t n, int variant);
bool pq_encode_fuses_lists(const OpqModelDev &m, const float *x_rot, int64_t n, int variant);
// ld: entries per (query, m) row of the output, 0 = K; ld > K pads with +inf
// tables: tables per query in the output, 0 = M; tables > M appends all-zero tables (the M < 16 scans: a row padded to 16 code bytes looks
// its zero bytes up there, and a zero changes no sum)
int launch_lut(const OpqModelDev &m, const float *q_rot, int64_t nq, const int32_t *list_id, float *lut,
               hipStream_t st, int ld = 0, int tables = 0);

// ---- adc_scan.hip ----
struct ScanPlan {
    int qtile;    // 1,2,4,8
    int splits;   // row splits per query group
    int variant;  // 0 = one row per lane, same sub-quantiser across the wave; 1 = skewed conflict-free (M = 16)
    // two-region plan (adc_scan16q only): the first groups_a query groups use `splits` row splits, the rest use
    // splits_b (> splits) so that the last, partly filled round of workgroups is made of shorter ones.
    // splits_b == 0: one region.  Partial results are laid out [nq][stride()][k].
    int groups_a = 0, splits_b = 0;
    int stride() const { return splits_b > splits ? splits_b : splits; }
    // M < 16 served by the M = 16 kernels (api.hip opq_plan): the scan reads a copy of the rows padded to 16 bytes with zeros and
    // per-query tables padded with all-zero tables; real_M = the model's M (0: not padded)
    int real_M = 0;
    // round 6: M = 8 / M = 4 NATIVE (adc_scan_p.hip: adc_scan16p): the rows as they lie in memory, 16 / M of them per 16-byte load, from
    // a packed pre-rotated copy (launch_rotate_codes_packed); the tables as for real_M (the model's first, zeros behind).  Set with real_M.
    bool packed = false;
};
ScanPlan plan_scan(const OpqModelDev &m, int64_t n_rows, int64_t nq, int k, int want_qtile, int want_splits,
                   int want_variant);
// adc_scan16q under a two-region plan whose first region scans every group in ONE piece: the lists of that region's queries are final
// when the scan kernel ends, so it writes them to the result arrays itself (launch_adc_scan's final_d / final_id) and the merge starts
// behind them.  Returns the number of leading queries this holds for (0: none; a multiple of the query tile).
// A plan without row splits at all (stride 1) has every query in place: nq.
inline int64_t scan_in_place_queries(const ScanPlan &plan, int M, int64_t nq)
{
    const int64_t groups = (nq + plan.qtile - 1) / plan.qtile;
    if (plan.variant < 3 || plan.variant > 4 || M != 16 || plan.qtile != 8 || plan.splits != 1) return 0;
    if (!(plan.splits_b > plan.splits && plan.groups_a > 0 && plan.groups_a < groups)) return plan.stride() == 1 ? nq : 0;
    return (int64_t)plan.groups_a * plan.qtile;
}
// part_d / part_id: [nq][plan.stride()][k]
// lut_scratch: nq * M * K floats, needed when plan.variant >= 3 (see scan_lut_floats)
// gthr: nq words of scratch (adc_scan16q with more than one row split: shared filter thresholds) or null;
// lazy: adc_scan16q selects on its integer lower bounds between checkpoints and re-sums exactly once, at the end
int launch_adc_scan(const OpqModelDev &m, const uint8_t *codes, int64_t n_rows, int64_t id_base, const float *q_rot,
                    int64_t nq, int k, const ScanPlan &plan, float *part_d, int64_t *part_id, float *lut_scratch,
                    const uint8_t *codes_rot, hipStream_t st, uint32_t *gthr = nullptr, int lazy = 1, float *final_d = nullptr,
                    int64_t *final_id = nullptr, const uint32_t *only = nullptr);
// only ([nq] words or null; k > 128 with one query per workgroup and ONE row split): answer query g only when only[g] != 0
// final_d / final_id ([nq][k], or null): where the first scan_in_place_queries() queries' lists go instead of part_* (with stride 1
// part_* ARE the final arrays: pass them again)
// M = 16 only: codes_rot rows [row0, n) = the code rows rotated left by (row & 15) bytes, the layout adc_scan16q
// (plan.variant >= 3) streams when codes_rot is given
int launch_rotate_codes(const uint8_t *codes, uint8_t *codes_rot, int64_t row0, int64_t n, hipStream_t st);
// rows [row0, n) of M code bytes -> 16 bytes each, zeros behind the M
int launch_pad_codes(const uint8_t *codes, int M, uint8_t *codes16, int64_t row0, int64_t n, hipStream_t st);
// ---- adc_scan_p.hip ----  M = 8 / 4: rows [row0, n) of M code bytes -> the packed pre-rotated copy adc_scan16p streams (same size as
// the rows, rounded up to 16 bytes: 16 / M rows per 16-byte group, each rotated left by (group & (M - 1)) bytes within its M bytes)
int launch_rotate_codes_packed(const uint8_t *codes, int M, uint8_t *codes_rot, int64_t row0, int64_t n, hipStream_t st);
struct ScanArgs;
int launch_adc_scan16p(const ScanArgs &a, int M, int64_t blocks, hipStream_t st);

// ---- adc_scan_h.hip: adc_scan16h (plan.variant == 6), a persistent grid walking a host-built item table ----
// one item = one row segment of one query group: rows [64 * row0_64, 64 * row0_64 + rows) scanned for the group's 8 queries;
// its k best go to partial list `sidx` of the group's `nseg` (ascending rows); nseg == 0 marks an unused table entry.  The 64-row
// chunks of the segment are walked from chunk `chunk0` on, wrapping around: the planner picks it so that the workgroups that are
// busy at the same time are at the same rows (they share them through their XCD's L2)
struct ScanItem {
    int32_t group;
    uint32_t row0_64, rows, chunk0;
    uint16_t sidx, nseg;
    uint32_t pad;
};
struct ScanHPlan {
    std::vector<ScanItem> items;  // [rounds][grid]
    std::vector<uint32_t> multi;  // [nq]: 1 = the query's group was scanned in several segments (its partial lists need the merge)
    int grid = 0, rounds = 0, stride = 1;  // workgroups, items per workgroup, partial lists per query
};
void scanh_plan(int64_t n_rows, int64_t nq, int splits, ScanHPlan &p, int cus = 0);   // cus = 0: the device's CU count
size_t scanh_spill_bytes(int grid);
// k = 129 .. 2048 through the filter scan (adc_scan_h.hip, round 6): sampled histogram bound -> candidate lists -> one selection workgroup
// per query; *flags_out ([nq] words inside `scratch`): 1 = not answered, the caller runs the exact kernel for those (launch_adc_scan's `only`)
bool scank_applies(const OpqModelDev &m, int64_t n_rows, int64_t nq, int k);
size_t scank_scratch_bytes(int64_t n_rows, int64_t nq, int k);
int launch_adc_scan_bigk(const OpqModelDev &m, const uint8_t *codes, const uint8_t *codes_rot, int64_t n_rows, int64_t id_base, const float *q_rot,
                         int64_t nq, int k, float *dist, int64_t *ids, float *lut_g, void *qlut, void *qp_g, void *scratch, int lazy,
                         uint32_t **flags_out, hipStream_t st);
size_t scanh_qlut_bytes(int64_t nq);
size_t scanh_qp_bytes(int64_t nq);
void set_scanh_balance(int v);       // 0 = choose, 1 = equal shares of the flat (group x row) space, 2 = (group, split) blocks
void set_scanh_min_rows(int64_t v);  // smallest share of a workgroup in the balanced plan
void set_scanh_tail(int v);
void set_scanh_share_hist(int v);
void set_scanh_fix(double v);
size_t scanh_gthr_bytes(int64_t nq);   // adc_scan16h: [nq] shared bounds + [nq][256] shared histograms          // 1 (default) = the groups of the last, partly filled round of blocks may be cut finer
// part_d / part_id: [nq][plan.stride][k]; out_d / out_id: the final [nq][k] lists (groups scanned in one piece write there); lut_g: nq * 16 * 256 floats; qlut / qp_g / spill: scanh_*_bytes; gthr: nq words or null
int launch_adc_scan_h(const OpqModelDev &m, const uint8_t *codes, const uint8_t *codes_rot, int64_t n_rows, int64_t id_base,
                      const float *q_rot, int64_t nq, int k, const ScanHPlan &plan, const ScanItem *items_dev, float *part_d,
                      int64_t *part_id, float *out_d, int64_t *out_id, float *lut_g, void *qlut, void *qp_g, void *spill, uint32_t *gthr, int lazy, int seed,
                      hipStream_t st);
int scan_seed_enabled();
// adc_scan_h.hip: 1 .. 8 queries -- global bound from a histogram pass, candidate lists, selection by the last workgroup (3 launches,
// the rotation of RAW queries folded into the first when rotate != 0)
bool scans_applies(const OpqModelDev &m, int64_t n_rows, int64_t nq, int k);
bool scans_fuses_rotation(const OpqModelDev &m);
void set_scans_dbg(int v);   // timing experiments, results wrong when non-zero   // else the caller rotates and passes rotate = 0
size_t scans_scratch_bytes();
int launch_adc_scan_small(const OpqModelDev &m, const uint8_t *codes, const uint8_t *codes_rot, int64_t n_rows, int64_t id_base, const float *q,
                          int rotate, int64_t nq, int k, float *dist, int64_t *ids, float *lut_g, void *qlut, void *qp_g, void *scratch, int lazy,
                          hipStream_t st);

// ---- topk_merge.hip ----
int launch_topk_merge(const float *in_d, const int64_t *in_id, int64_t nq, int L, int k, float *out_d, int64_t *out_id,
                      hipStream_t st, const uint32_t *only_if = nullptr);

int launch_topk_select(const float *in_d, const int64_t *in_id, int64_t nq, int64_t n_cand, int k, float *out_d,
                       int64_t *out_id, hipStream_t st, const uint32_t *only_if = nullptr);
// L per-rank lists of k per query in the all-gather layout: list r of query q at in_d[r * stride_d + q * k ..] / in_id[r * stride_id + ..]
int launch_topk_merge_gathered(const float *in_d, const int64_t *in_id, int64_t stride_d, int64_t stride_id, int64_t nq, int L, int k,
                               float *out_d, int64_t *out_id, hipStream_t st);

// ---- query_video.hip ----
// probe[nq][nprobe] list ids in visiting order; list_off[coarseK+1]; codes/video_id in list order.
int launch_coarse_probe(const OpqModelDev &m, const float *q_rot, int64_t nq, int nprobe, int32_t *probe,
                        hipStream_t st, void *scratch = nullptr);
size_t coarse_probe_scratch_bytes(int64_t nq, int nprobe);   // the few-queries form's partial lists (placed behind the probe array)
// coarse top-nprobe through the matrix-core filter (assign_mfma.hip): same probes as the exact kernels
void set_probe_variant(int v);
void set_scan_seed(int v);  // adc_scan16q / 16a: 1 (default) = first thresholds from a histogram of the split's first 2048 rows
bool coarse_probe_filter_applies(const float *q, int64_t nq, int d, const float *cent, int k, int nprobe);
int launch_coarse_probe_filtered(const float *q, int64_t nq, int d, const float *cent, int k, int nprobe, int32_t *probe, hipStream_t st);
int launch_query_video(const OpqModelDev &m, const float *q_rot, int64_t nq, int nprobe, const int32_t *probe,
                       const int64_t *list_off, const uint8_t *codes, const int32_t *video_id, int img_num,
                       float *match_score, int64_t longest_list, hipStream_t st);
// list-ordered (CSR) copy of the entries by a stable counting sort on the device; scratch of csr_scratch_bytes();
// stats_out (16 bytes, device): int64 longest list, int32 min / max video id
size_t csr_scratch_bytes(int64_t n, int L, int *nb_out);
int launch_csr_build(const int32_t *lists, const int32_t *videos, const uint8_t *codes, int64_t n, int L, int M, void *scratch,
                     int64_t *list_off, uint8_t *out_codes, int32_t *out_videos, void *stats_out, hipStream_t st);

// ---- flat.hip ----
int flat_plan_splits(int64_t n, int64_t nq, int qtile);
int flat_qtile(int64_t nq);
// part_d / part_id: [nq][splits][k]; for CVTMI_METRIC_L2U8 part_d carries the int32 distance BITS
// (non-negative ints order like their float bit patterns, nothing does float arithmetic on them).
// only_if != nullptr: [nq] predicate words, a workgroup whose queries are all 0 exits at once (the conditional re-run of
// queries a filter gave up on)
int launch_flat_search(int metric, int D, const void *data, int64_t n, const void *q, int64_t nq, int k, int qtile,
                       int splits, float *part_d, int64_t *part_id, hipStream_t st, const uint32_t *only_if = nullptr);
// uint8 L2 on the i8 matrix cores (D % 32 == 0, D <= 512, nq >= 8); norms[n] = sum (x-128)^2 per row
int launch_flat_u8_norms(const uint8_t *x, int64_t n, int D, int32_t *norms, hipStream_t st);
// fp32 metrics with D % 4 == 0 keep their rows in a blocked layout (float4 c of 64 consecutive rows contiguous)
// so that a wave's row reads are coalesced; launch_flat_search expects that layout for those shapes
inline bool flat_blocked(int metric, int D) { return metric != CVTMI_METRIC_L2U8 && (D % 4) == 0; }
// flat_mfma.hip: fp32 IP / L2 search through the bf16 matrix-core filter (32 <= D <= 128, D % 16 == 0, nq >= 64)
bool flat_filter_applies(int metric, int D, int64_t n, int64_t nq, int k);
size_t flat_pack_bytes(int nch, int64_t n);   // nch K steps of 16 dimensions per row (>= D / 16: zeros beyond D)
int launch_flat_pack(const float *X, int64_t n, int D, int nch, int metric, uint4 *pack, uint32_t *bias, uint32_t *stats, hipStream_t st, int64_t row0 = 0);
int launch_flat_thr(const float *q, int64_t nq, int D, int metric, const float *sample_d, int k, uint32_t *stats, float *thr,
                    float *margin, hipStream_t st);
int launch_flat_filter(const float *q, int64_t nq, int D, const uint4 *pack, const uint32_t *bias, const float *thr, int64_t row_begin,
                       int64_t n, uint32_t pair_cap, uint32_t *pair_cnt, uint4 *pairs, hipStream_t st);
int launch_flat_finish(int metric, const float *X, int64_t n, int D, const float *q, int64_t nq, const uint32_t *pair_cnt, uint32_t pair_cap,
                       const uint4 *pairs, int cap, int k, const float *margin, const float *sample_d, const int64_t *sample_i,
                       uint32_t *cand_cnt, float *cand_t, int32_t *cand_row, float *out_d, int64_t *out_i, uint32_t *overflow,
                       hipStream_t st);
// uint8 L2 through the same pipeline (exact integer distances on the i8 matrix cores: no bound, no second cut)
bool flat_u8_filter_applies(int D, int64_t n, int64_t nq, int k);
size_t flat_u8_pack_bytes(int D, int64_t n);
int launch_flat_u8_pack(const uint8_t *X, int64_t n, int D, uint4 *pack, hipStream_t st, int64_t row0 = 0);   // rows [row0, n), row0 % 32 == 0
int launch_flat_u8_filter(const uint8_t *q, int64_t nq, int D, const uint4 *pack, const int32_t *norms, const float *sample_d, int k,
                          int64_t row_begin, int64_t n, uint32_t pair_cap, uint32_t *pair_cnt, uint4 *pairs, hipStream_t st);
int launch_flat_u8_finish(int64_t nq, const uint32_t *pair_cnt, uint32_t pair_cap, const uint4 *pairs, int cap, int k, const float *sample_d,
                          const int64_t *sample_i, uint32_t *cand_cnt, float *cand_d, int32_t *cand_row, float *out_d, int64_t *out_i,
                          uint32_t *overflow, hipStream_t st);
int launch_flat_unblock(const float *blocked, int64_t row0, int64_t n, int D, float *dst, hipStream_t st);   // rows [row0 / 64 * 64, n) of the blocked layout -> row-major dst
int launch_flat_block(const float *src, int64_t n, int D, int64_t row0, float *dst, hipStream_t st);
// flat_f32_stream.hip: fp32 IP / L2 search as one stream over the blocked rows (bf16 matrix-core scores, group best / second best,
// exact distances of the candidates); D % 16 == 0, 16 <= D <= 256, up to flat_f32_stream_qmax(D) queries per pass
int flat_f32_stream_qmax(int D);
int flat_f32_stream_private_max(int D);   // queries per pass of the private-ring kernel
int get_flat_f32_dbg();
void set_flat_f32_dbg(int v);     // timing experiments, results wrong when non-zero
void set_flat_f32_share(int v);   // shared-ring kernel: 0 choose, 1 four waves x 32 QB queries, 2 eight waves x 32 queries
void set_flat_f32_nt(int v);     // 0 = never, 1 = choose (default), 2 = always: non-temporal hint on the stream kernels' row loads
bool flat_f32_stream_applies(int metric, int D, int64_t n, int k);
// round 6 (flat_f32_tfilter.hip): batches (any width that is a multiple of 4 up to 2048-d, >= 262 144 rows, k <= 128) as a threshold filter: queries in LDS, the rows'
// bf16 operand copy (launch_flat_pack) in registers, per-query thresholds from sample maxima, candidate lists, exact finish;
// redo[nq] (zeroed inside): 1 = the exact kernels must answer the query
int flat_f32_tfilter_nch(int D);   // K steps of the kernel that takes D-dimensional rows (0: none)
bool flat_f32_tfilter_width(int D);
bool flat_f32_tfilter_applies(int metric, int D, int64_t n, int64_t nq, int k);
size_t flat_f32_tfilter_scratch(int64_t nq, int k);
// Xrows: row-major copy of the rows (launch_flat_unblock) for the exact finish, or null (it gathers from the blocked layout: eight times the bytes)
int launch_flat_f32_tfilter(int metric, int D, const float *X, const float *Xrows, const void *pack, const uint32_t *pstats, const float *bias, const uint32_t *stats, int64_t n,
                            const float *q, int64_t nq, int k, void *scratch, float *out_d, int64_t *out_i, uint32_t *redo, hipStream_t st);
void set_flat_f32_tfilter(int v);
void set_flat_f32_tfilter_min(int v);
void set_flat_f32_tfilter_one(int v);
void set_flat_f32_tfilter_retry(int v);
void set_flat_f32_tfilter_wide_band(int v);
void set_flat_f32_tfilter_bigk(int v);
void set_flat_f32_tfilter_sample(int v);
void set_flat_f32_tfilter_min_rows(int v);
int64_t flat_f32_tfilter_min_rows();
size_t flat_f32_stream_scratch(int D, int64_t n, int64_t nq_pass);
// round 6 (flat_u8_tfilter.hip): uint8 L2 batches (from 128 queries; any batch when k = 129 .. CVTMI_K_MAX; 32 .. 512-d in steps of 32,
// >= 262 144 rows) as a threshold filter over the int8 operand copy
// (launch_flat_u8_pack): exact integer scores, so no margins; flags[nq + 1] (device, zeroed inside): [0] != 0 afterwards = the other paths must answer the
// call, [1 + q] != 0 = query q
bool flat_u8_tfilter_width(int D);
bool flat_u8_tfilter_applies(int D, int64_t n, int64_t nq, int k);
size_t flat_u8_tfilter_scratch(int D, int64_t n, int64_t nq, int k);
int launch_flat_u8_tfilter(int D, const void *pack, const int32_t *norms, int64_t n, const uint8_t *q, int64_t nq, int k, void *scratch, float *out_d,
                           int64_t *out_i, uint32_t *flags, hipStream_t st);
void set_flat_u8_tfilter(int v);
void set_flat_u8_tfilter_min_k(int v);
void set_flat_u8_tfilter_min_rows(int64_t v);
void set_flat_u8_tfilter_small_min_nq(int v);
void set_flat_u8_tfilter_min_nq(int v);
void set_flat_u8_tfilter_min_nq_k65(int v);
void set_flat_u8_tfilter_sample(int v);
void set_flat_u8_tfilter_chunks(int v);
// bias[r] (and zeroed padding rows) for rows [row0, row1); stats[0] = max |x|^2 bits, stats[1] = non-finite rows (both accumulate)
int launch_flat_f32_bias(float *X, int D, int metric, int64_t row0, int64_t row1, float *bias, uint32_t *stats, hipStream_t st);
// redo[nq], cnt[nq] (zeroed inside): redo is set to 1 for queries the exact kernels must answer
int launch_flat_f32_stream(int metric, int D, const float *X, const float *bias, const uint32_t *stats, int64_t n, const float *q, int64_t nq,
                           int k, void *scratch, float *out_d, int64_t *out_i, uint32_t *redo, uint32_t *cnt, hipStream_t st, const void *pack = nullptr,
                           const uint32_t *pstats = nullptr, const float *Xrows = nullptr);   // Xrows: row-major copy of the rows for the exact distances, or null
void set_flat_f32_packed(int v);   // 1 (default): up to 32 queries stream the bf16 operand copy (launch_flat_pack) when pack / pstats are given
int flat_u8_mfma_qtile(int D, int k, int64_t nq);  // queries per workgroup, 0 = shape not covered
int flat_u8_mfma_splits(int64_t n, int64_t nq, int qt);
// gthr: nq uint32 scratch (set to 0xff.. inside) through which the row splits of a query share their k-th best
int launch_flat_u8_mfma(int D, const uint8_t *data, const int32_t *norms, int64_t n, const uint8_t *q, int64_t nq, int k,
                        int splits, float *part_d, int64_t *part_id, uint32_t *gthr, hipStream_t st);
// ids[i] = ids[i] >= 0 ? labels[ids[i]] : -1
void set_flat_u8_opt(int v);
int set_flat_u8_dbg(int v);   // -DCVTMI_GF_DBG builds only
// uint8 L2, 1..128 queries: matrix-core stream over the raw rows keeping tile / wave minima (flat_mfma.hip) + selection (flat.hip)
bool flat_u8_mstream_applies(int D, int64_t n, int64_t nq, int k);
size_t flat_u8_mstream_scratch(int64_t n, int64_t nq, int *nqp, int *waves);
int64_t flat_u8_mstream_groups(int64_t n);   // entries of the tile-group minima array
int flat_u8_mstream_group();                 // tiles per group
int launch_flat_u8_mstream(int D, const uint8_t *data, const int32_t *norms, int64_t n, const uint8_t *q, int64_t nq, int32_t *tmin,
                           int32_t *wmin, hipStream_t st);
int launch_flat_u8_mstream_finish(int D, const uint8_t *data, int64_t n, const uint8_t *q, int64_t nq, int k, const int32_t *wmin, int G,
                                  const int32_t *tmin, int nqp, int tile_group, float *part_d, int64_t *part_id, float *out_d, int64_t *out_rows, hipStream_t st);
int flat_u8_stream_slices();
void set_flat_u8_mstream_min(int v);
void set_flat_u8_mstream_min_rows(int64_t v);
void set_sq8_wave_blocks(int v);
void set_sq8_encode_wave(int v);
void set_sq8_filter(int v);
void set_scan_tail_splits(int v);
void set_sq8_flags(int v);
size_t flat_u8_stream_scratch(int64_t n, int64_t nq, int *slices, int64_t *ld);
int launch_flat_u8_stream(int D, const uint8_t *data, const int32_t *norms, int64_t n, const uint8_t *q, int64_t nq, int k, float *scratch,
                          float *part_d, int64_t *part_id, float *out_d, int64_t *out_rows, hipStream_t st);
void set_flat_u8_gfilter(int v);   // flat_mfma.hip: the software-pipelined (LDS-DMA) uint8 filter kernel on / off
bool flat_u8_gfilter_shape(int D);
int launch_gather_labels(int64_t *ids, int64_t count, const int64_t *labels, hipStream_t st);
// ids[i] = ids[i] >= 0 ? ids[i] + base : -1
int launch_offset_labels(int64_t *ids, int64_t count, int64_t base, hipStream_t st);

// ---- sq8.hip ----
// den[n] = float(max(1e-12, sqrt(sum_i double(x_i * x_i))))   (int8_quan.cc:46-52)
int launch_sq8_rownorm(const float *x, int64_t n, int d, float *den, hipStream_t st);
// Int8Encode over n rows (int8_quan.cc:72-94): optional L2 normalisation (written back over x when write_back),
// then the bytes.  den_scratch: n floats, used only for row widths the single-pass kernel does not take.
int launch_sq8_encode_rows(const float *vmin, const float *vdiff, int d, float *x, int64_t n, int l2norm, int write_back,
                           uint8_t *codes, float *den_scratch, hipStream_t st);
int launch_sq8_decode(const float *vmin, const float *vdiff, int d, const uint8_t *codes, int64_t n, float *x,
                      hipStream_t st, int mode = 0);
// kmin/kmax: [d] ordered-uint32 scratch (initialised inside); results in vmin / vdiff; den_scratch as above
int launch_sq8_train(const float *x, int64_t n, int d, int l2norm, float *den_scratch, uint32_t *kmin, uint32_t *kmax,
                     float *vmin, float *vdiff, hipStream_t st);
// true when (d, pointers) take the single-pass kernel, i.e. no den_scratch is needed
bool sq8_single_pass(int d, const void *x, const void *codes, const void *vmin, const void *vdiff, int64_t n = 0);   // n: rows of the call (the wave kernels take >= 4096)

// ---- kmeans.hip (codebook training) ----
// assign[r] = nearest of cent[k][d] (first minimum; -1 when no distance is below float(UINT_MAX)); *changed +=
// number of rows whose assignment moved.  x rows are ld floats apart.
// pca.hip: y = normalise((x - mean) * E^T), E [dout][din] (pca_utils.cc:25-35)
int launch_pca_project(const float *mean, const float *E, int din, int dout, const float *x, int64_t n, int l2norm, float *y,
                       hipStream_t st);
// assign_mfma.hip: nearest centroid through the bf16 matrix-core filter (32 <= d <= 128), exact kernels for undecided rows
bool assign_filter_applies(const float *x, int64_t ld, int64_t n, int d, const float *cent, int k);
int launch_assign_filtered(const float *x, int64_t ld, int64_t n, int d, const float *cent, int k, int32_t *assign,
                           unsigned long long *changed, hipStream_t st);
int launch_kmeans_assign_exact(const float *x, int64_t ld, int64_t n, int d, const float *cent, int k, int32_t *assign,
                               unsigned long long *changed, hipStream_t st);
// exact assignment of few rows: centroid range cut into `splits` workgroup columns, folded in ascending order; part_* hold
// splits * n entries each
int launch_kmeans_assign_split(const float *x, int64_t ld, int64_t n, int d, const float *cent, int k, int32_t *assign, int splits,
                               float *part_d, int32_t *part_i, hipStream_t st);
void set_assign_variant(int v);  // 0 = choose, 1 = exact kernels, 2 = filter wherever it applies
int launch_kmeans_assign(const float *x, int64_t ld, int64_t n, int d, const float *cent, int k, int32_t *assign,
                         unsigned long long *changed, hipStream_t st);
// cent[c] = float(double sum of the rows assigned to c, ascending row order / count); empty clusters untouched
int launch_kmeans_update(const float *x, int64_t ld, int64_t n, int d, const int32_t *assign, int k, float *cent,
                         hipStream_t st);
int launch_kmeans_gather(const float *x, int64_t ld, int d, const int64_t *rows, int k, float *cent, hipStream_t st);
int launch_kmeans_fill(int32_t *p, int64_t n, int32_t v, hipStream_t st);
int launch_kmeans_residual(const float *x, int64_t n, int d, const float *cent, const int32_t *assign, float *res,
                           hipStream_t st);

// ---- hnsw.hip ----
struct HnswDevGraph {
    const float *vec;          // [n][D] vectors, internal-id order
    const uint32_t *links0;    // [n][maxM0 + 1]: count, neighbours
    const int64_t *labels;     // [n] external labels
    const int64_t *upper_off;  // [n] first word of the element's upper-level block, -1 if it lives on level 0 only
    const uint32_t *upper;     // levels x (maxM + 1) words per element that has upper levels
    int64_t n;
    int D, maxM, maxM0, maxlevel;
    uint32_t enterpoint;
};
// one wave per slot; visited: slots x words uint32; cand_scratch: slots x gcap 8-byte entries; *err set on overflow
int launch_hnsw_search(const HnswDevGraph &g, int metric, const float *q, int64_t nq, int k, int ef, float *out_d,
                       int64_t *out_label, uint32_t *visited, void *cand_scratch, int slots, int64_t words, int64_t gcap,
                       int *err, hipStream_t st);
int launch_hnsw_search_adc(const HnswDevGraph &g, const float *lut, const uint8_t *codes, int M, int K, int64_t nq, int k, int ef,
                           float *out_d, int64_t *out_label, uint32_t *visited, void *cand_scratch, int slots, int64_t words,
                           int64_t gcap, int *err, hipStream_t st, int raw_ids = 0, int state_floats = 0);
// exact fp32 distances (reference summation order) of the raw queries to the nodes ids [nq][R] (-1 = padding -> +inf)
int launch_hnsw_rerank(const HnswDevGraph &g, int metric, const float *q, int64_t nq, int R, const int64_t *ids, float *out_d,
                       hipStream_t st);
int hnsw_lds_bytes(int state_floats, int ef);
int hnsw_ef_max();
int hnsw_lcap();
int hnsw_top_lds(int ef);
void set_hnsw_top_lds(int v);
void set_hnsw_adc_tables(int v);
int hnsw_adc_state_floats(int MK);

}  // namespace cvtmi
