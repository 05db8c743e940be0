/*
 * cvtmi.h -- C ABI of the MI355X-native OPQ-encode / ADC-search hot path of willard-yuan/cvt.
 *
 * This is the drop-in boundary: a C++ (or cgo / ctypes / JNI) caller that today runs the loops
 * of the reference's L1 layer on the CPU binds these entry points instead.  Each entry names the
 * reference interface it replaces (paths relative to the reference tree).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, opaque handles, no C++ / torch types.
 *   - Every function returns 0 (CVTMI_OK) or a negative cvtmi_status; the message of the last
 *     failure on the calling thread is cvtmi_last_error().  No exception crosses the boundary.
 *   - Functions without a suffix take HOST pointers (caller-owned buffers, like the reference's
 *     float** / float* arguments) and return when the result is in the caller's buffer.
 *     Functions ending in `_dev` take DEVICE pointers (HBM-resident) plus a `stream`
 *     (a hipStream_t passed as void*, NULL = the default stream) and are asynchronous.
 *   - Handles may be shared between threads and streams.  SEARCHES on one handle (cvtmi_opq_search*, cvtmi_opq_query_video*,
 *     cvtmi_flat_search*, cvtmi_hnsw_search*) are reads, as QueryThrehold / searchKnn are in the reference
 *     (opq/src/IVFOPQ.cpp:322-422, brutoforce.hpp:73-93, hnswalg.h:688-728): each call leases a scratch set from the
 *     handle's pool (tables, partial top-k lists, visited bitmaps, staging buffers; the pool grows to the number of calls
 *     in flight) and runs on its caller's stream (host-pointer entries: the set's own stream), so searches from several
 *     threads or streams proceed side by side and every one returns what a search alone would.  Calls that CHANGE a
 *     handle (add, add_codes, set_param, load) take it exclusively: they wait for the searches in flight, and searches
 *     issued later wait for them; the result is that of some serial order.  cvtmi_hnsw_search_adc* holds its OPQ handle
 *     the way a search of that handle would.  A sharded search additionally serialises on its communicator (below).  The
 *     small accessors (ntotal, set_id_base) take no lock: do not race them with calls that change the handle.
 *   - One handle lives on the HIP device that was current when it was created.
 *   - The library has no CPU fallback: without a usable HIP device every compute entry fails
 *     with CVTMI_EHIP.
 *
 * Numerics contract (tests/ enforce it against oracle/ and the golden vectors):
 *   uint8 PQ codes, list ids, SQ8 codes, uint8-L2 distances: bit-exact.
 *   LUT entries and ADC distances: bit-exact (same fp32 operation order as the reference,
 *   separate multiply and add), which implies the 1e-4 relative bound of the north star.
 *   top-k: the k smallest (distance, id) pairs in lexicographic order, ascending.
 */
#ifndef CVTMI_H
#define CVTMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CVTMI_VERSION 200 /* 0.2.0 */
/* Largest k of the search entries (cvtmi_opq_search*, cvtmi_flat_search*, cvtmi_topk_*): searchKnn(query, k) and
 * get_sort_results(score, num_show) of the reference take any k (brutoforce.hpp:73-93, opq/src/common.h:25-37).  k <= 128 runs on
 * every kernel; 129 .. CVTMI_K_MAX on the exact kernels (one query per workgroup, 4096-entry selection buffer) -- slower per query. */
#define CVTMI_K_MAX 2048

typedef enum cvtmi_status {
    CVTMI_OK = 0,
    CVTMI_EINVAL = -1,       /* bad argument */
    CVTMI_ENOMEM = -2,       /* host or device allocation failed */
    CVTMI_EHIP = -3,         /* HIP runtime error / no device */
    CVTMI_ESTATE = -4,       /* handle not in a state that allows the call */
    CVTMI_EUNSUPPORTED = -5, /* shape outside what the kernels are built for */
    CVTMI_ECOMM = -6         /* RCCL not loadable / a collective failed */
} cvtmi_status;

typedef enum cvtmi_metric {
    CVTMI_METRIC_IP = 0,   /* 1 - <q,x>, fp32   : brute_force_search/src/space_ip.hpp:211-239 */
    CVTMI_METRIC_L2F = 1,  /* sum (q-x)^2, fp32  : hnsw_sifts_retrieval/hnswlib/space_l2.h:153-184 */
    CVTMI_METRIC_L2U8 = 2  /* sum (q-x)^2, uint8 -> int32 : space_l2.h:186-245 (L2SqrI / L2SpaceI) */
} cvtmi_metric;

typedef struct cvtmi_opq_s *cvtmi_opq_t;
typedef struct cvtmi_flat_s *cvtmi_flat_t;
typedef struct cvtmi_hnsw_s *cvtmi_hnsw_t;
typedef struct cvtmi_comm_s *cvtmi_comm_t;

/* ---------------------------------------------------------------- library / device ---------- */
int cvtmi_version(void);
const char *cvtmi_last_error(void);
int cvtmi_device_count(int *count);
int cvtmi_set_device(int device);
/* Library-wide tuning / measurement hooks (no effect on results).
 *   "assign_variant"  nearest-centroid assignment (coarse argmin of cvtmi_opq_encode, cvtmi_kmeans): 0 = choose (default);
 *                     1 = the reference's chain for every centroid on the VALU; 2 = bf16 matrix-core filter with exact
 *                     resolution of undecided rows wherever it applies (32 <= d <= 128, d % 16 == 0, k >= 64)
 *   "flat_variant"    exhaustive search: 0 = choose (default: large batches through the filter pipelines -- fp32: bf16 matrix-core
 *                     filter with a proven bound; uint8: exact sample, software-pipelined i8 matrix-core threshold filter, sort,
 *                     from 256 queries on >= 1 M rows; smaller uint8 batches are one stream over the rows, 128 queries per pass); 1 = exact / row-tile
 *                     kernels only; 2 = the filter pipelines wherever they apply
 *   "flat_u8_gfilter" the uint8 filter stage: 1 (default) / 3 = LDS-DMA pipelined kernel, one 8-wave workgroup per CU; 2 = two 4-wave
 *                     workgroups per CU; 4 = one wave per SIMD holding 96-128 queries (GEMM-shaped; measured equal / slower);
 *                     0 = the round-1 filter kernel (register-staged tiles)
 *   "flat_u8_dbg"     timing experiments of the filter kernel (-DCVTMI_GF_DBG builds only; EUNSUPPORTED otherwise: results are wrong)
 *   "flat_u8_mstream_min"  smallest uint8 batch that takes the streaming matrix-core kernel (default 1; 129 = never: row-per-lane / row-tile kernels)
 *   "sq8_encode_wave" 0 = SQ8 encode through the 64-row tile kernel for every width (default 1: wave-per-row kernel at d = 256 / 512)
 *   "sq8_wave_blocks" workgroups per CU of the wave-per-row SQ8 training kernel (default 3; powers of two lose 8 % to HBM channel conflicts)
 *   "flat_u8_opt"     measurement variants of the uint8 row-tile kernel (0 = shipped; 1..3 spill registers and are slower)
 *   "probe_variant"   coarse top-nk of cvtmi_opq_query_video: 0 = choose (matrix-core filter + exact distances of the candidates
 *                     from 256 query frames, 32 <= D <= 128, coarseK >= 256); 1 = exact kernels only; 2 = filter wherever it applies
 *   "flat_f32_nt"     non-temporal hint on the row loads of the fp32 stream kernels (rows a CU reads once per launch need no place in
 *                     L2 / Infinity Cache): 0 = never, 1 = choose (default: everywhere except three query blocks per wave), 2 = always
 *   "scan_seed"       1 (default) = scan variants 3 / 4 / 5 take their first filter thresholds from a histogram of the first 2048 rows of
 *                     a row split instead of starting with "every row passes" (1-4 % on 1 M rows, more on short splits); 0 = off
 *   "scan_tail_splits" adc_scan16q, query groups past the last full round of workgroups: 0 (default) = cut them into 2 / 4 / 8 row
 *                     splits while that still leaves at most one workgroup per CU (a last round is only expensive while it leaves CUs
 *                     empty: 4256 queries over 1 M rows 2.11 -> 2.79 M queries/s, 10 000 queries unchanged); -1 = never; S > 0 = S splits
 *   "scan_pad_m"      1 (default) = an OPQ index with M < 16 is searched by the M = 16 scan kernels over rows padded to 16 code bytes
 *                     with zeros (derived copies: 32 bytes per row) and per-query tables padded with all-zero tables -- M = 8 at
 *                     10 000 queries x 1 M rows: 9.8 -> 3.2 ms, and every M from 1 to 15 is searchable; 0 = the row-per-lane
 *                     kernels (M = 4 / 8 only)
 *   "scan_bigk"       1 (default) = an OPQ search (M = 16, >= 65 536 rows) with k = 129 .. 2048 runs through the filter scan: a sampled
 *                     histogram bound per query, candidate lists, one selection workgroup per query, the exact kernel behind it for the
 *                     queries it could not answer (round 6: k = 1000 at 10 000 queries x 1 M rows within 2x of k = 100); 0 = the exact
 *                     kernel for every query (one query per workgroup: 11x slower at that size)
 *   "scan_packed_m"   1 (default) = an OPQ index with M = 8 or M = 4 is scanned as it lies in memory (adc_scan16p, round 6): 16 / M rows
 *                     per 16-byte load from a pre-rotated copy of the rows (+M bytes per row, no padded copy), 16 / M copies of the M
 *                     tables in LDS so that the look-ups stay conflict-free -- no look-up is spent on a zero table; 0 = the padded
 *                     rows of "scan_pad_m" for these M as well
 *   "sq8_host_small" 1 (default) = SQ8 host-pointer calls of up to 1 MB (the reference's one vector per call) run out of a page-locked
 *                     scratch area the kernels read and write directly (512-d: 80 -> 26 us per call); 0 = allocate, copy, free
 *   "scans_max_work"  OPQ search: the small-batch form (<= 128 queries) answers while rows x query groups stays at or under this
 *                     (default 48 << 20; beyond, its per-group passes over the table lose to the persistent grid: 100 M rows,
 *                     128 queries 8.3 against 3.7 ms)
 *   "flat_u8_filter_min_nq" / "flat_u8_filter_min_rows" / "flat_u8_filter_min_work"  uint8 search: the sample + matrix-core filter pipeline
 *                     answers from this many queries (default 129), rows (524 288) and rows x width x queries in units of 1e9
 *                     (130) on; below, passes of up to 128 queries through the streaming kernel (round 5: the fitted crossover)
 *   "flat_u8_tfilter" 1 (default) = uint8 batches over >= 65 536 rows of 32 … 512 bytes in steps of 32 go through the threshold
 *                     filter of round 6 (flat_u8_tfilter.hip: exact integer scores on the i8 matrix cores, 256 queries in LDS per pass, thresholds
 *                     from 4096 sample maxima, no margins): every batch when k = 129 .. 2048 (2 M x 512-d, 1000 queries, k = 129: 139 -> 1.5 ms),
 *                     k <= 128 from "flat_u8_tfilter_min_nq" (129) queries on, k = 65 .. 128 and tables of a GB or more from "flat_u8_tfilter_min_nq_k65" (97) on and only
 *                     for k >= "flat_u8_tfilter_min_k" (1); 0 = the streaming passes / the sample + filter pipeline / the exact kernels as in
 *                     round 5.  "flat_u8_tfilter_sample": the sample pass takes one tile group in this many (0 = sqrt(8000 x GB of rows / k) within 2 .. 32);
 *                     "flat_u8_tfilter_chunks": query chunks (1 / 2 / 4, default 4) that share one pass over the rows;
 *                     "flat_u8_tfilter_min_rows" (262 144) / "flat_u8_tfilter_small_min_nq" (129): tables under _min_rows come here from that many queries on;
 *                     widths without a streaming kernel (all but 128 / 256 / 512 bytes) from two queries on
 *   "flat_u8_mstream_min_rows" smallest table the uint8 streaming kernel takes (default 4096 = its structural bound; it was 262 144 until
 *                     round 5: 65 536 x 512-d, 100 queries 0.84 -> 0.06 ms)
 *   "flat_u8_sample_passes" the uint8 filter pipeline searches its leading sample exactly through the streaming kernel while that takes at
 *                     most this many 128-query passes (default 10: batches up to 1280 queries), through the row-tile kernels beyond
 *   "flat_small_zero_copy" 1 (default) = a small host-pointer flat search (queries <= 64 KB, lists <= 768 KB: the brute_force CLI's one
 *                     searchKnn per query) sends its queries up from a page-locked staging area and lets the last kernel write the
 *                     lists into that area (1 M x 128-d, one query: 137 -> 125 us per call); 0 = three copy-engine copies
 *   "opq_host_zero_copy" 1 (default) = cvtmi_opq_search with page-locked result arrays lets the kernels write them (one launch chain);
 *                     0 = pipelined pieces through device buffers and the copy engines, as for pageable arrays
 *   "sq8_filter"      1 (default) = the wave-per-row SQ8 kernels (d = 256 / 512) decide code bytes / column extremes from a bounded
 *                     approximation and run the reference's chain (two correctly rounded divisions + the byte) only where it cannot
 *                     decide; 0 = the chain for every element.  Same codes, rows and ranges, bit for bit
 *   "sq8_flags"       bit 0 (default set) = the row-norm sums of those kernels run on DPP moves instead of the ds_bpermute butterfly
 *   "hnsw_adc_tables" HNSW over OPQ codes: 0 (default) = a query's fp32 distance tables are read from the table scratch (L2 / Infinity
 *                     Cache), 32 traversals per CU; 1 = copied into LDS first (16 KB per query: 8 traversals per CU, round 2 - 4)
 *   "comm_force_rccl" 1 = cvtmi_comm_create goes through RCCL (ncclCommInitRank, ncclAllGather) for world == 1 too, which
 *                     otherwise needs no transport (test hook for 1-GPU boxes)
 *   "comm_inject_failure" r >= 0: the local search of rank r of every sharded search fails (tests of the failure path); -1 = off
 *   "opq_host_chunk"  queries per piece of cvtmi_opq_search's pipelined form for pageable arrays (default 4096 = one full round of
 *                     workgroups; 0 = one piece)
 *   "opq_small_zero_copy" 1 (default) = host-pointer OPQ searches that take the small-batch form read their queries from and write their
 *                     lists to the handle's pinned staging area from the kernels (no copy engine in the chain); 0 = copies
 *   "host_spin_us"    microseconds a host-pointer entry polls its stream before it blocks (default 200: waking from a blocking wait costs
 *                     as much again as a small search)
 *   "scanh_balance" / "scanh_min_rows" / "scanh_tail" / "scanh_fix" / "scanh_share_hist"  planner of the persistent-grid scan (variant
 *                     6): 0 choose / 1 equal row-time shares / 2 row blocks; smallest row segment; two-region tail on / off; an item's
 *                     fixed cost in row-equivalents (160 000); one candidate histogram per query shared by its segments (1) or not
 *   "flat_f32_tfilter" fp32 searches (any width that is a multiple of 4 up to 2048-d, >= 262 144 rows, k <= 128; k up to 2048: "flat_f32_tfilter_bigk") of "flat_f32_tfilter_min" queries or more run as a
 *                     threshold filter (round 6, flat_f32_tfilter.hip): sample maxima -> per-query threshold -> queries in LDS, the rows'
 *                     bf16 operand copy in registers, no barrier, hits recorded -> per-query lists -> exact distances.  1 .. 3 = the
 *                     bf16 products per term: 1 (x1.q1), 2 ((x1 + x2).q1; both with margins from each query's own rounding residues),
 *                     3 (x1.q1 + x2.q1 + x1.q2, the stream kernels' margin); 4 (default) = one product up to "flat_f32_tfilter_one"
 *                     (default: no limit -- measured ahead at every batch size) queries, two beyond; 0 = the stream kernels for every
 *                     batch.  1 M x 128-d, top-100: 1000 queries 0.97 -> 0.69 ms, 128 queries 0.25 -> 0.16 ms, 16 queries 0.119 -> 0.107 ms
 *   "flat_f32_tfilter_min"  smallest batch that takes that pipeline; 0 (default) = choose: 65 ... 97 at the widths the stream kernels take (smaller batches
 *                     stream the operand copy, "flat_f32_packed"), 16 at the others
 *   "flat_f32_tfilter_one"  largest batch that multiplies one product under "flat_f32_tfilter" 4
 *   "flat_f32_packed" 1 (default) = fp32 searches of up to 32 queries (the stream kernels' private rings) read the bf16 operand copy the threshold
 *                     filter keeps -- one ready-made term per value, half the bytes of the fp32 rows, one product, margins from the query's
 *                     own rounding residues -- on tables that have one (>= "flat_f32_tfilter_min_rows" rows); 0 = the fp32 rows, split on the fly
 *   "flat_f32_tfilter_min_rows"  smallest table that takes that pipeline (default 262 144; >= 32 768.  Measured on 128-d: below ~130 K rows the stream
 *                     kernels are ahead for fewer than 128 queries and level beyond; 200 K rows, 1000 queries 0.42 -> 0.32 ms)
 *   "flat_f32_tfilter_sample"  the sample that sets the thresholds is about 1 / this of the row-tile groups, spread evenly over the
 *                     rows and rounded to a whole number of groups per wave (1 M x 128-d, 1000 queries, k = 100: 1/8 0.63 ms, 1/5 0.505, 1/3 0.52);
 *                     0 (default) = sqrt(1280 x GB / k) within 3 .. 32 (: 4 M x 128-d, k = 10: 1.31 -> 1.04 ms)
 *   "flat_f32_rows_copy"  narrowest fp32 row (default 4 = every width the threshold filter takes; 0 = never) that gets a row-major copy beside the
 *                     blocked rows once the filter answers on the handle: + 4 D bytes per row, the exact finish reads whole cache lines instead of 16 of
 *                     every 128 bytes (262 144 x 1024-d, 1000 queries: 1.89 -> 1.14 ms; 1 M x 128-d 0.53 -> 0.47); skipped where it does not fit
 *   "flat_f32_tfilter_wide_band"  1 (default) = rows of 256 dimensions or more: a query with more than 1024 rows inside the margin band of its k-th
 *                     score (tight, non-negative features) gets a second finish that ranks up to 4096 (262 144 x 1024-d RootSIFT-shaped rows,
 *                     k = 128: 18.6 -> 6.2 ms per 1000 queries); 0 = the exact kernels answer such a query
 *   "flat_f32_tfilter_bigk"  1 (default) = fp32 searches with k = 129 .. 2048 (every batch size, tables of "flat_f32_tfilter_min_rows" rows and more)
 *                     run through the threshold filter with 4096 sample maxima, candidate lists of 32 768 and a workgroup-wide selection
 *                     (1 M x 128-d, 1000 queries: k = 129 46.6 -> 1.3 ms, k = 1000 47.8 -> 2.4, k = 2048 53.5 -> 3.2); 0 = the exact kernels
 *   "flat_f32_tfilter_retry"  1 = a query whose candidate list ran over takes ONE second filter pass under the threshold its own stored
 *                     candidates give (it helps when the rows above the sample's threshold are many, not when the rows inside the
 *                     margin band are: measured no gain on clustered 300-d rows, three empty launches = ~10 us on every search);
 *                     0 (default) = the exact kernels at once
 *   "flat_f32_share"  fp32 stream, batches beyond one wave's queries: 0 choose (round 6: the four-wave shared ring -- the eight- and
 *                     twelve-wave forms answered one query of ~10^5 wrongly in a sweep against the exact kernels and are only run when
 *                     asked for), 1 the four-wave form, 2 the twelve-wave shared-ring kernel,
 *                     3 = the shared ring with eight waves of 64 queries (512 queries per pass over the rows instead of 384; round 6,
 *                     measured: 385 .. 512 queries 0.74 -> 0.51 ms on 1 M x 128-d, 1000 queries unchanged -- its registers spill)
 *   "hnsw_top_lds"    entries of an HNSW traversal's top queue kept in LDS (default 256; 0 = all)
 *   "hnsw_slots"      cap on HNSW traversals per CU (0 = what LDS allows, at most 32)
 *   "scans_dbg" / "flat_f32_dbg"  measurement switches of the small-batch scan and of the fp32 stream (phases skipped: results are WRONG
 *                     when non-zero; development only) */
int cvtmi_set_tuning(const char *name, int64_t value);

/* ---------------------------------------------------------------- OPQ model + code index ---- */
/*
 * Replaces IVFOPQ::LoadModel's in-memory tables (opq/src/IVFOPQ.cpp:64-102):
 *   coarse [coarseK][D] fp32, books [M][K][D/M] fp32 (each sub-codebook contiguous), and the
 *   "rotation": either perm[D] (the reference's reorder_, y[i] = x[perm[i]], IVFOPQ.cpp:424-439)
 *   or a dense row-major R[D][D] (y = R x, the general OPQ rotation run as an fp32 MFMA GEMM);
 *   both NULL = identity.  K <= 256, M <= 16 (IVFelem::PQindex[16], IVFOPQ.h:28), D % M == 0.
 */
int cvtmi_opq_create(int D, int coarseK, int M, int K, const float *coarse, const float *books,
                     const float *R, const int32_t *perm, cvtmi_opq_t *out);
int cvtmi_opq_destroy(cvtmi_opq_t h);

/* IVFOPQ::reorder over n rows (IVFOPQ.cpp:424-439, :459-461).  x and y may not alias. */
int cvtmi_opq_rotate(cvtmi_opq_t h, const float *x, int64_t n, float *y);
int cvtmi_opq_rotate_dev(cvtmi_opq_t h, const float *x, int64_t n, float *y, void *stream);

/* The encode loop of IVFOPQ::Add (IVFOPQ.cpp:107-163) on already-rotated rows:
 * list_id[n] = coarse argmin (first minimum, -1 if none), codes[n][M] = per-sub-quantiser argmin
 * (255 if none).  list_id may be NULL. */
int cvtmi_opq_encode(cvtmi_opq_t h, const float *x_rot, int64_t n, int32_t *list_id, uint8_t *codes);
int cvtmi_opq_encode_dev(cvtmi_opq_t h, const float *x_rot, int64_t n, int32_t *list_id, uint8_t *codes,
                         void *stream);

/* LoadSingleFeatFile's reorder + Add's encode (IVFOPQ.cpp:459-461 + :107-163) on RAW rows in one call: what cvtmi_opq_rotate followed
 * by cvtmi_opq_encode computes, without a caller-side buffer of rotated rows (they pass through a cache-resident scratch of the
 * handle).  Same codes and list ids. */
int cvtmi_opq_rotate_encode(cvtmi_opq_t h, const float *x, int64_t n, int32_t *list_id, uint8_t *codes);
int cvtmi_opq_rotate_encode_dev(cvtmi_opq_t h, const float *x, int64_t n, int32_t *list_id, uint8_t *codes,
                                void *stream);

/* m_ivfList[vw].push_back(elem) of Add (IVFOPQ.cpp:167): append n entries to the resident index.
 * list_id NULL = list 0 (only valid when coarseK == 1); video_id NULL = one "video" per entry,
 * numbered by insertion order.  Entry ids are id_base + insertion index. */
int cvtmi_opq_add_codes(cvtmi_opq_t h, const uint8_t *codes, const int32_t *list_id,
                        const int32_t *video_id, int64_t n);
int cvtmi_opq_add_codes_dev(cvtmi_opq_t h, const uint8_t *codes, const int32_t *list_id,
                            const int32_t *video_id, int64_t n, void *stream);
int cvtmi_opq_reserve(cvtmi_opq_t h, int64_t n_total);
int cvtmi_opq_ntotal(cvtmi_opq_t h, int64_t *n);
int cvtmi_opq_reset(cvtmi_opq_t h);                      /* drop all entries, keep the model */
int cvtmi_opq_set_id_base(cvtmi_opq_t h, int64_t base);  /* first id of this row shard */
/* Copy the resident entries back in list order (the SaveIndex order, IVFOPQ.cpp:557-575).
 * list_off[coarseK+1], video_id[ntotal], codes[ntotal][M]; any may be NULL. */
int cvtmi_opq_get_entries(cvtmi_opq_t h, int64_t *list_off, int32_t *video_id, uint8_t *codes);

/* PQ_table of Query (IVFOPQ.cpp:273-291): lut[nq][M][K] for rotated queries; list_id[nq] selects
 * the coarse centroid each residual is taken against (NULL = list 0). */
int cvtmi_opq_lut(cvtmi_opq_t h, const float *q_rot, int64_t nq, const int32_t *list_id, float *lut);
int cvtmi_opq_lut_dev(cvtmi_opq_t h, const float *q_rot, int64_t nq, const int32_t *list_id, float *lut,
                      void *stream);

/* The north-star search: (optional rotation) + LUT + exhaustive ADC scan (IVFOPQ.cpp:300-306)
 * + k smallest (distance, id) (opq/src/common.h:25-37) per query, over every resident entry.
 * Requires coarseK == 1.  dist[nq][k] / ids[nq][k] ascending; rows short of k entries are padded
 * with (+inf, -1).  rotate != 0 applies cvtmi_opq_rotate to the queries first.  1 <= k <= CVTMI_K_MAX. */
int cvtmi_opq_search(cvtmi_opq_t h, const float *q, int64_t nq, int rotate, int k, float *dist,
                     int64_t *ids);
int cvtmi_opq_search_dev(cvtmi_opq_t h, const float *q, int64_t nq, int rotate, int k, float *dist,
                         int64_t *ids, void *stream);

/* IVFOPQ::Query / QueryThrehold (IVFOPQ.cpp:213-320 / :322-422): per query frame probe the nprobe
 * nearest coarse lists and keep, per video, the minimum ADC score clamped at 1.0.
 * match_score[nq][img_num].  rotate as above.  The list-ordered copy of the entries is (re)built on the device by the
 * first query after an append (a stable counting sort; ~2.5 ms per million entries); entries whose list id is outside
 * [0, coarseK) are skipped; a video id outside [0, img_num) fails the call. */
int cvtmi_opq_query_video(cvtmi_opq_t h, const float *q, int64_t nq, int rotate, int nprobe,
                          int img_num, float *match_score);
int cvtmi_opq_query_video_dev(cvtmi_opq_t h, const float *q, int64_t nq, int rotate, int nprobe,
                              int img_num, float *match_score, void *stream);

/* Tuning / measurement hooks (no effect on results).
 *   "splits"   row splits per query group of the scan (0 = automatic)
 *   "qtile"    queries sharing one pass over the codes: 1, 2, 4 or 8 (0 = automatic)
 *   "profile"  1 = bracket the scan kernel with HIP events on its stream
 *   "encode_variant"  PQ encode kernel: 0 = choose (default); 1 = every centroid through the reference's sub / mul / add
 *                 chain on the VALU; 2 = fp32 matrix-core filter, exact chain only for pairs it cannot separate
 *                 (K = 256, step 8 or 16, D <= 128; chosen automatically from 8192 rows up).  Same codes either way.
 *   "scan_variant"  M = 16 kernel choice: 0 row-per-lane; 1 / 2 skewed fp32 tables (512 / 1024 threads);
 *                   3 / 4 skewed 15-bit lower-bound tables, 8 queries per pass (1024 / 512 threads; 3 = default);
 *                   5 = variant 3 without checkpoints: one wave of the workgroup compacts beside the 15 that scan
 *                   (adc_scan16a; measured 7-10 % slower than 3 on 1 M rows, kept for comparison)
 *                   6 = adc_scan16h: the tables of variant 3, quantised once per query group by a preparation kernel; a
 *                   persistent grid (two workgroups per CU) walks a host-built item table (cvtmi_opq_scan_plan); candidates go
 *                   to per-workgroup areas in HBM and are selected once per (row segment, query), the filter bounds come from
 *                   a histogram of the candidates' integer sums -- ONE histogram per query for all the row segments its group
 *                   is cut into, so a segment's bound comes from the rows all of them have seen; takes any number of queries
 *                   7 = (default) the library's choice: 6 where it measured ahead (100 ... 3200 queries on a cache-resident
 *                   index: wherever the planner cuts a query group into row segments), 3 elsewhere
 *   "prerotate"   1 (default) = variants 3 / 4 stream a copy of the code rows in which row r is rotated by r & 15
 *                 bytes (the lane skew of the conflict-free table reads), kept next to the rows: +16 bytes of HBM
 *                 per row, 12 VALU instructions fewer per row in the VALU-bound scan loop; 0 = rotate in registers
 *   "tail_split"  1 (default) = with automatic splits, the query groups of the last, partly filled round of
 *                 workgroups may be split finer than the others (variants 3 / 4); 0 = one split count for all
 *   "scan_lazy"   1 (default) = variants 3 / 4 select on their integer lower bounds between checkpoints and compute exact
 *                 reference-order sums once, for the rows still held at the end; 0 = exact sums at every checkpoint
 *   "scan_share"  1 (default) = the row splits of a query publish their filter thresholds to each other (variants 3 / 4)
 *   "scan_small"  1 (default) = batches of 1 .. 128 queries (M = 16, >= 65 536 rows, k <= 128, "scan_variant" 7) take the small-batch
 *                 path: per group of 8 queries a histogram pass over a quarter of the rows fixes one global bound per query, a second
 *                 pass collects the rows below it into per-workgroup lists, one workgroup per query selects (exact fall-back inside the
 *                 kernel when a list overflows) -- four launches, rotation included, no partial lists and no merge
 *   "groups_a", "splits_b"  force that two-region shape: the first groups_a query groups use "splits" row
 *                 splits, the others splits_b (> splits); 0 = planner's choice */
int cvtmi_opq_set_param(cvtmi_opq_t h, const char *name, int64_t value);
/* With "profile" on: MEAN duration of the scan-kernel launches recorded since the previous call
 * (the 64 most recent are kept) and the algorithmic code bytes ONE launch reads
 * (passes x rows x M, passes = ceil(nq / qtile)).  Synchronises on the profiling events only. */
int cvtmi_opq_last_scan(cvtmi_opq_t h, float *ms, int64_t *code_bytes, int *qtile, int *splits);
/* The item table "scan_variant" 6 would walk for n_rows code rows and nq queries (pure host logic: no device needed; the
 * number of workgroup slots is taken as 2 x `cus`, 0 = the current device's CU count or 256 without one).  splits > 0 forces
 * (query group, row split) blocks, 0 lets the planner choose (equal shares of the flat group x row space when the code matrix
 * is cache-resident and there are at least as many query groups as slots).  items: up to cap entries of 6 ints {query group, first
 * row / 64, rows, partial-list index inside the group, partial lists of the group (0 = unused entry), first 64-row chunk of the
 * wrap-around walk over the segment}, item i of workgroup w at [i * grid + w].  Returns the number of
 * entries (rounds * grid; nothing is written past cap) or a negative status. */
int64_t cvtmi_opq_scan_plan(int64_t n_rows, int64_t nq, int splits, int cus, int64_t *items, int64_t cap, int *grid, int *rounds,
                            int *stride);

/* Which scan form a search of nq queries (top k) over n_rows code rows of a D-dimensional model with M sub-quantisers of K codewords
 * would take under the default settings (pure host logic: no device needed; 256 CUs are assumed without one).  out[0] = 1: the
 * small-batch form (up to 128 queries); otherwise out[1] = scan variant (3 / 4 = adc_scan16q, 5 = adc_scan16a, 6 = persistent grid,
 * 0 .. 2 = the row-per-lane / fp32-table kernels), out[2] = queries per pass, out[3] = row splits, out[4] / out[5] = groups of the first
 * region and row splits of the second of a two-region plan (0 0: one region), out[6] = M when the M = 16 kernels run over padded rows
 * (M < 16), else 0.  What tests/test_scan_plan.py pins the dispatch rules of round 5 with. */
int cvtmi_opq_describe_dispatch(int D, int M, int K, int64_t n_rows, int64_t nq, int k, int out[7]);

/* The same for a flat search (metric: CVTMI_METRIC_*; pure host logic): out[0] = 1 the fp32 one-stream kernels / 2 the fp32 threshold filter
 * (round 6: any width that is a multiple of 4 up to 2048-d, >= 262 144 rows, batches from "flat_f32_tfilter_min" queries on), out[1] = the fp32 sample + matrix-core filter
 * pipeline is eligible behind them, out[2] = the uint8 sample + filter pipeline, out[3] = uint8 streaming passes of up
 * to 128 queries; all zero: the exact / row-tile kernels. */
int cvtmi_flat_describe_dispatch(int metric, int D, int64_t n_rows, int64_t nq, int k, int out[4]);

/* Page-locked host memory.  The host-pointer entries move their arrays through pinned staging areas (one extra host copy each way);
 * arrays that already ARE page-locked -- from here, or the caller's own hipHostMalloc / hipHostRegister -- need none: page-locked
 * queries go to the copy engine as they are, and page-locked RESULT arrays of cvtmi_opq_search are written by the kernels themselves
 * (device-visible host memory: no device copy of the lists, no copy back, the batch is not cut into pieces; tuning key
 * "opq_host_zero_copy", default 1).  10 000 queries x top-100 over 1 M rows: pageable arrays in pipelined pieces 2.7 M queries/s,
 * page-locked arrays 3.1 M, device pointers 3.3 M (round 5, profiles/r05_host_api_sweep.txt). */
int cvtmi_host_alloc(size_t bytes, void **p);
int cvtmi_host_free(void *p);

/* ---------------------------------------------------------------- top-k merge ---------------- */
/* Exchange step of a row-sharded search: merge L sorted (distance, id) lists per query
 * (in_dist / in_ids [nq][L][k], id < 0 = padding, lists ordered by ascending id range) into the
 * k smallest pairs.  Same shape as FLANN-MPI's ResultsMerger
 * (retrieval/vlindex/lib/FLANN/mpi/index.h:74-108). */
int cvtmi_topk_merge(const float *in_dist, const int64_t *in_ids, int64_t nq, int L, int k,
                     float *dist, int64_t *ids);
int cvtmi_topk_merge_dev(const float *in_dist, const int64_t *in_ids, int64_t nq, int L, int k,
                         float *dist, int64_t *ids, void *stream);

/* ---------------------------------------------------------------- row-sharded search --------- */
/* The north-star multi-GPU layout: the code rows are split into contiguous blocks, one per GPU, one
 * process per GPU; a search scans the local block, then ONE RCCL all-gather (xGMI) of the per-shard top-k lists and a
 * k-way merge on every rank.  The reference tree's only distributed search has this shape: FLANN-MPI's
 * mpi::Index::knnSearch (retrieval/vlindex/lib/FLANN/mpi/index.h:196-226: local knnSearch, indices += offset_,
 * boost::mpi reduce with ResultsMerger :74-108).
 *
 * A communicator spans `world` ranks, this process being `rank` on the HIP device current at creation:
 *   cvtmi_comm_unique_id   rank 0 draws the job's id (ncclGetUniqueId) and hands its CVTMI_COMM_ID_BYTES bytes to the
 *                          other ranks by any means (env, file, MPI, torch.distributed ...);
 *   cvtmi_comm_create      every rank calls it at the same time (ncclCommInitRank).  world == 1 needs no id and no RCCL;
 *   cvtmi_comm_create_custom  the same exchange over a caller-supplied all-gather instead of RCCL (an MPI job like the
 *                          reference's, or several ranks sharing one GPU in tests): fn must gather `bytes` bytes from
 *                          every rank into recv_dev in rank order, IN PLACE (send_dev == recv_dev + rank * bytes), ordered
 *                          after the work already enqueued on `stream` and complete, or enqueued on `stream`, on return;
 *                          0 = success.  All pointers are device pointers.
 * Collectives must be entered by all ranks in the same order.  A rank whose LOCAL search fails still enters the
 * all-gather, with its error code in its slot's status word, so nobody is left waiting.  What the other ranks do with the
 * status words is cvtmi_set_tuning("comm_check_status", v):
 *   2 (default)  checked on the device, behind the merge, without synchronising the stream: if any rank failed, the results
 *                of that search are overwritten with the padding pattern (+inf, -1) on EVERY rank, and every rank gets
 *                CVTMI_ECOMM from its next call on the communicator, from cvtmi_comm_status, or -- host-pointer entries,
 *                which synchronise for their copy anyway -- from the failing call itself;
 *   1            read back inside the call (one stream synchronisation per sharded search): every rank returns
 *                CVTMI_ECOMM from the failing call itself, `_dev` entries included;
 *   0            ignored: only the rank that failed returns an error.
 * Calls on one communicator are serialised like calls that change a handle (the communicator's lock is taken before the
 * handle's): drive one communicator from one thread, or issue the searches that share it in the same order on every rank. */
#define CVTMI_COMM_ID_BYTES 128
typedef int (*cvtmi_allgather_fn)(void *ctx, const void *send_dev, void *recv_dev, size_t bytes, void *stream);
int cvtmi_comm_unique_id(void *id /* [CVTMI_COMM_ID_BYTES] */);
int cvtmi_comm_create(const void *id, int rank, int world, cvtmi_comm_t *out);
/* CVTMI_ECOMM (once) if a search since the last report failed on some rank under the deferred status check, else CVTMI_OK.
 * Does not synchronise: call it after the stream of the searches in question has been synchronised. */
int cvtmi_comm_status(cvtmi_comm_t c);
int cvtmi_comm_create_custom(cvtmi_allgather_fn fn, void *ctx, int rank, int world, cvtmi_comm_t *out);
/* ONE process driving every GPU (the reference's callers are single processes: opq/src/multi_frame_index_test.cpp:32-91):
 * ncclCommInitAll over `ndev` devices (devices == NULL: 0 .. ndev - 1); comms[d] is rank d of ndev on devices[d].  Use with
 * the *_sharded_all searches below (their all-gathers leave as one ncclGroupStart / ncclGroupEnd group); each communicator is
 * destroyed with cvtmi_comm_destroy. */
int cvtmi_comm_create_all(int ndev, const int *devices, cvtmi_comm_t *comms /* [ndev] */);
int cvtmi_comm_destroy(cvtmi_comm_t c);
/* rank / world; transport: 0 none (world == 1), 1 RCCL, 2 caller-supplied; all-gathers issued so far and the bytes
 * each rank contributed to the last one.  Any pointer may be NULL. */
int cvtmi_comm_info(cvtmi_comm_t c, int *rank, int *world, int *transport, int64_t *collectives, int64_t *bytes_per_rank);
/* Bytes one rank contributes to the all-gather of an [nq][k] result (status word + fp32 + int64 fields, 16-byte aligned): the
 * size a caller-supplied transport has to stage per rank. */
int cvtmi_comm_slot_bytes(int64_t nq, int k, size_t *bytes);
/* Allocates the communicator's gather buffer (world x slot bytes) for results of up to [nq][k] ahead of the searches.  A sharded
 * search whose LOCAL part fails still enters the all-gather and fails on every rank (the status word travels in the slot); the one
 * failure that cannot be signalled that way is running out of device memory for this buffer inside a search -- the rank returns
 * CVTMI_ENOMEM before the collective and its peers wait in theirs.  Reserving up front moves that failure to a point where the
 * caller can still tell the other ranks. */
int cvtmi_comm_reserve(cvtmi_comm_t c, int64_t nq, int k);
/* Row block of `rank`: [begin, end) of n_total rows, the first n_total % world ranks own one row more. */
int cvtmi_shard_range(int64_t n_total, int rank, int world, int64_t *begin, int64_t *end);
/* cvtmi_opq_search on a row shard: `h` holds this rank's block of rows (cvtmi_opq_set_id_base = its first row), every
 * rank passes the same queries; dist / ids [nq][k] = the global result, identical on every rank and identical to a
 * single handle holding all rows. */
int cvtmi_opq_search_sharded(cvtmi_opq_t h, cvtmi_comm_t c, const float *q, int64_t nq, int rotate, int k, float *dist,
                             int64_t *ids);
int cvtmi_opq_search_sharded_dev(cvtmi_opq_t h, cvtmi_comm_t c, const float *q, int64_t nq, int rotate, int k,
                                 float *dist, int64_t *ids, void *stream);
/* The same on one process: handles[d] = the row block of device d, comms from cvtmi_comm_create_all; host pointers. */
int cvtmi_opq_search_sharded_all(cvtmi_opq_t *handles, cvtmi_comm_t *comms, int ndev, const float *q, int64_t nq, int rotate,
                                 int k, float *dist, int64_t *ids);
/* The exchange step alone, for per-shard lists produced by any search: local_dist / local_ids [nq][k] ascending
 * (distance, id) with GLOBAL ids (id < 0 = padding), ranks holding ascending id ranges.  Distances are 4-byte fields: fp32,
 * or non-negative int32 (the uint8 metric) passed as their bit patterns -- they order the same way. */
int cvtmi_shard_merge_topk_dev(cvtmi_comm_t c, const float *local_dist, const int64_t *local_ids, int64_t nq, int k,
                               float *dist, int64_t *ids, void *stream);

/* get_sort_results (opq/src/common.h:25-37): the k smallest (score, index) pairs of scores[nq][n],
 * ascending; rows short of k entries are padded with (+inf, -1).  1 <= k <= CVTMI_K_MAX. */
int cvtmi_topk_select(const float *scores, int64_t nq, int64_t n, int k, float *dist, int64_t *ids);
int cvtmi_topk_select_dev(const float *scores, int64_t nq, int64_t n, int k, float *dist, int64_t *ids,
                          void *stream);

/* ---------------------------------------------------------------- exhaustive (flat) search -- */
/* hnswlib::BruteforceSearch<dist_t> + SpaceInterface (brute_force_search/src/brutoforce.hpp:9-136,
 * hnswlib.hpp:34-58).  Rows are D fp32 (IP, L2F) or D uint8 (L2U8). */
int cvtmi_flat_create(int metric, int D, cvtmi_flat_t *out);
int cvtmi_flat_destroy(cvtmi_flat_t h);
/* addPoint (brutoforce.hpp:43-56) for n rows; labels NULL = row numbers continuing from ntotal.
 * Labels must be unique and ascending in insertion order for the (distance, label) tie rule to
 * hold on device; the C++ BruteforceSearch mirror re-orders rows by label when they are not. */
int cvtmi_flat_add(cvtmi_flat_t h, const void *x, const int64_t *labels, int64_t n);
int cvtmi_flat_add_dev(cvtmi_flat_t h, const void *x, const int64_t *labels, int64_t n, void *stream);
int cvtmi_flat_ntotal(cvtmi_flat_t h, int64_t *n);
int cvtmi_flat_reset(cvtmi_flat_t h);
/* searchKnn (brutoforce.hpp:73-93) for nq queries: k smallest (distance, label), ascending.
 * dist is float[nq][k] for IP / L2F and int32_t[nq][k] for L2U8.  1 <= k <= CVTMI_K_MAX.  An index with fewer than k rows pads:
 * label -1, distance field 0x7f800000 -- +inf as a float, and for L2U8 the same bit pattern read as int32 (2139095040: larger than
 * any real distance, which is what lets the padded lists of row shards merge like all others). */
int cvtmi_flat_search(cvtmi_flat_t h, const void *q, int64_t nq, int k, void *dist, int64_t *labels);
int cvtmi_flat_search_dev(cvtmi_flat_t h, const void *q, int64_t nq, int k, void *dist, int64_t *labels,
                          void *stream);
/* Measurement hook: how the last search was answered -- *filtered = 0 the exact kernels; 1 the sample + matrix-core filter
 * pipeline (fp32 metrics, 32 <= D <= 128, D % 16 == 0, >= 131072 rows: exact search of a leading sample, bf16 matrix-core
 * products decide which other rows need an exact distance); 2 the fp32 stream (D in {32, 64, 96, 128, 192, 256}, >= 32768 rows:
 * one pass over the rows scores them on the bf16 matrix cores, the k-and-a-few candidates get exact distances; queries its
 * error bound does not cover are re-run by the exact kernels inside the same call) -- same results every way -- and the largest
 * candidate list of the pipeline.  cvtmi_set_tuning("flat_variant", 1) allows the exact kernels only, 2 prefers the pipeline;
 * cvtmi_set_tuning("flat_f32_stream", 0 / 1 / 2) = never / choose / wherever it applies. */
int cvtmi_flat_last_search(cvtmi_flat_t h, int *filtered, int64_t *max_candidates);
/* Row shards of an exhaustive index (BruteforceSearch<dist_t>::searchKnn over rows split across GPUs: config 3's 5 GB of
 * uint8 rows shard like config 4's codes).  Rows added without labels report label = id_base + row; the sharded searches
 * return the global k best -- float distances, or the int32 distances of the uint8 metric (same 4-byte fields) -- identical
 * on every rank and to one handle holding all rows.  See cvtmi_opq_search_sharded* for the communicator rules. */
int cvtmi_flat_set_id_base(cvtmi_flat_t h, int64_t base);
int cvtmi_flat_search_sharded(cvtmi_flat_t h, cvtmi_comm_t c, const void *q, int64_t nq, int k, void *dist, int64_t *labels);
int cvtmi_flat_search_sharded_dev(cvtmi_flat_t h, cvtmi_comm_t c, const void *q, int64_t nq, int k, void *dist, int64_t *labels,
                                  void *stream);
int cvtmi_flat_search_sharded_all(cvtmi_flat_t *handles, cvtmi_comm_t *comms, int ndev, const void *q, int64_t nq, int k,
                                  void *dist, int64_t *labels);

/* ---------------------------------------------------------------- int8 scalar quantisation -- */
/* Per-dimension min / (max - min) over (optionally L2-normalised) rows: what
 * faiss::IndexScalarQuantizer(d, QT_8bit).train leaves in sq.trained
 * (scalar_quantization/train/src/sq_train.cpp:84-103).  x is not modified. */
int cvtmi_sq8_train(const float *x, int64_t n, int d, int l2norm, float *vmin, float *vdiff);
int cvtmi_sq8_train_dev(const float *x, int64_t n, int d, int l2norm, float *vmin, float *vdiff,
                        void *stream);
/* Int8Quan::Int8Encode arithmetic (scalar_quantization/scalar_quantization/int8_quan.cc:72-94)
 * over n rows; with l2norm == 1 each row of x is L2-normalised IN PLACE first (:46-56), as the
 * reference does to its caller's buffer; l2norm == 2 encodes the normalised rows but leaves x as it
 * was (same codes, 4 d bytes of write traffic per row less: for callers that do not read x again). */
int cvtmi_sq8_encode(const float *vmin, const float *vdiff, int d, float *x, int64_t n, int l2norm,
                     uint8_t *codes);
int cvtmi_sq8_encode_dev(const float *vmin, const float *vdiff, int d, float *x, int64_t n, int l2norm,
                         uint8_t *codes, void *stream);
/* Int8Quan::Int8Decode(std::string&) arithmetic (int8_quan.cc:117-132) over n rows. */
int cvtmi_sq8_decode(const float *vmin, const float *vdiff, int d, const uint8_t *codes, int64_t n,
                     float *x);
int cvtmi_sq8_decode_dev(const float *vmin, const float *vdiff, int d, const uint8_t *codes, int64_t n,
                         float *x, void *stream);
/* Int8Quan::Int8Decode(uint8_t*) and Int8DecodeFaiss (int8_quan.cc:96-115) do NOT use the formula above: they call
 * faiss::ScalarQuantizer::decode.  faiss is a dependency that is not vendored in the reference (pinned 1.5.3 by its build notes);
 * its published QT_8bit codec is fp32 throughout: xi = (code + 0.5f) / 255.0f, x = vmin + xi * vdiff, product and sum rounded
 * separately (a faiss built with FMA contraction could fuse the two; this is the ISO evaluation).  About a third of the floats
 * differ from cvtmi_sq8_decode's by one ulp. */
int cvtmi_sq8_decode_faiss(const float *vmin, const float *vdiff, int d, const uint8_t *codes, int64_t n,
                           float *x);
int cvtmi_sq8_decode_faiss_dev(const float *vmin, const float *vdiff, int d, const uint8_t *codes, int64_t n,
                               float *x, void *stream);

/* ---------------------------------------------------------------- PCA projection ------------- */
/* cvtk::PCAUtils::reduceDim (pca_train_project/pca_online/pca_utils.cc:25-35; same arithmetic in
 * project/pca_dimension.h:47-58): y = cv::PCA::project(x) = (x - mean) * vectors^T, then with l2norm != 0 every
 * row divided by float(max(1e-12, sqrt(y . y))).  mean [din], vectors [dout][din] row-major (the "mean" and
 * "vectors" matrices of the model file), x [n][din], y [n][dout].  din % 4 == 0, dout <= 256.
 * cv::PCA::project is OpenCV (3.2 / 3.3 per the reference's comments), absent here: PARITY UNPINNED.  The
 * projection is evaluated as the k-ascending fp32 fused multiply-add chain of (x[k] - mean[k]) * vectors[j][k]
 * (OpenCV's own gemm accumulates in double or calls a BLAS, depending on its build); tests hold it within
 * 2e-6 absolute of the double-accumulated value on unit-norm outputs. */
int cvtmi_pca_project(const float *mean, const float *vectors, int din, int dout, const float *x, int64_t n,
                      int l2norm, float *y);
int cvtmi_pca_project_dev(const float *mean, const float *vectors, int din, int dout, const float *x,
                          int64_t n, int l2norm, float *y, void *stream);

/* ---------------------------------------------------------------- codebook training ---------- */
/* TrainPQ::CoarseQuan / ProdQuan (opq/train_codebook/train_PQ_codebook.cpp:150-244).  The reference calls
 * yael's kmeans(d, n, k, niter = 0, v, nt, seed = 1, redo = 1, ...), which is not vendored: PARITY UNPINNED.
 * What runs here is a fully specified Lloyd iteration (the tests hold it bit-exact against the CPU checker):
 * k distinct seed rows drawn with splitmix64(seed); nearest-centroid assignment with the arithmetic of
 * IVFOPQ::Add (sequential fp32 distance, strict '<'); centroid = float(double sum in ascending row order /
 * count), empty clusters keep their centroid; stops when a pass changes no assignment or after niter updates
 * (niter = 0: until convergence, at most 100).  x: [n] rows, ld floats apart, d <= 512 columns used.
 * centroids [k][d]; assign [n] may be NULL; iters_done may be NULL. */
int cvtmi_kmeans(const float *x, int64_t n, int d, int k, int niter, uint64_t seed, float *centroids,
                 int32_t *assign, int *iters_done);
int cvtmi_kmeans_dev(const float *x, int64_t ld, int64_t n, int d, int k, int niter, uint64_t seed,
                     float *centroids, int32_t *assign, int *iters_done, void *stream);
/* TrainPQ::IFVPQ (:144-148) on already permuted rows (LoadFeatureSample :77-82): coarse k-means on the whole
 * vectors, residuals x - coarse[assign] (:190-197), one k-means per sub-space on the residual columns
 * (:214-233), every k-means with the same seed as in the reference.  coarse [coarseK][D], books [M][K][D/M]
 * = the SaveCodebook / LoadModel layout (:283-287, IVFOPQ.cpp:75-95). */
int cvtmi_opq_train(const float *x, int64_t n, int D, int coarseK, int M, int K, int niter, uint64_t seed,
                    float *coarse, float *books);
int cvtmi_opq_train_dev(const float *x, int64_t n, int D, int coarseK, int M, int K, int niter, uint64_t seed,
                        float *coarse, float *books, void *stream);
/* OPQ rotation learning (optional): the dense D x D rotation the a-R GEMM applies, learned from a sample instead of
 * handed in.  NOT in the reference (opq/ only permutes dimensions, reorder_ IVFOPQ.cpp:424-439, and reads the permutation from a
 * file): self-specified, held bit for bit to the oracle's orc_opq_learn_rotation.  Non-parametric alternation (Ge et al. 2013):
 * R = I; `outer` times { Xr = X R^T (the fp32 MFMA GEMM); per sub-space k-means of Xr (cvtmi_kmeans: K centroids, niter, seed);
 * Y = the rows' reconstructions; C = X^T Y in double (rows in blocks of 1024, ascending); R = V U^T for C = U S V^T (orthogonal
 * Procrustes, one-sided Jacobi on the host) }; books = the k-means of the final Xr.  outer = 0 returns the identity and plain PQ
 * codebooks.  R [D][D] row-major and books [M][K][D/M] feed cvtmi_opq_create (coarseK = 1, a zero centroid: the exhaustive
 * configuration).  D in {32, 64, 96, 128}. */
int cvtmi_opq_learn_rotation(const float *x, int64_t n, int D, int M, int K, int outer, int niter, uint64_t seed, float *R,
                             float *books);
int cvtmi_opq_learn_rotation_dev(const float *x, int64_t n, int D, int M, int K, int outer, int niter, uint64_t seed, float *R,
                                 float *books, void *stream);

/* ---------------------------------------------------------------- HNSW search ---------------- */
/* hnswlib::HierarchicalNSW<float> (hnsw_sifts_retrieval/hnswlib/hnswalg.h), search side.
 * cvtmi_hnsw_load  = loadIndex (:522-581): takes the bytes of a file written by the reference's saveIndex
 *   (:491-519) for D-dimensional fp32 vectors and moves vectors, level-0 links, upper-level links and labels
 *   to HBM.  metric: CVTMI_METRIC_IP (InnerProductSpace) or CVTMI_METRIC_L2F (L2Space).
 * cvtmi_hnsw_search = setEf(ef) + searchKnn(query, k) (:688-729) for nq queries at once, one wave per query:
 *   dist / labels [nq][k] in ascending (distance, label) order -- the order the reference's result queue
 *   yields back to front --, padded with (0, -1) when the graph returns fewer than k.  Distances use the
 *   summation order of the reference's distance functions and both priority queues replay libstdc++'s
 *   push_heap / pop_heap (the reference compares by distance only, so heap mechanics decide ties): labels
 *   and distances are bit-identical to the reference's on the same graph.  k, ef <= 1024.
 * Graph construction (addPoint) is not offered here: graphs are built with the reference's tools. */
int cvtmi_hnsw_load(const void *file, int64_t bytes, int metric, int D, cvtmi_hnsw_t *out);
int cvtmi_hnsw_destroy(cvtmi_hnsw_t h);
int64_t cvtmi_hnsw_ntotal(cvtmi_hnsw_t h);
int cvtmi_hnsw_search(cvtmi_hnsw_t h, const float *q, int64_t nq, int k, int ef, float *dist, int64_t *labels);
int cvtmi_hnsw_search_dev(cvtmi_hnsw_t h, const float *q, int64_t nq, int k, int ef, float *dist, int64_t *labels,
                          void *stream);
/* HNSW over OPQ-compressed vectors (BASELINE config 5; not in the reference, whose HNSW holds fp32 vectors):
 * the same traversal over the same graph, but a node's distance is the ADC sum of the query's tables over the
 * node's PQ code (IVFOPQ.cpp:273-291 tables, :302-306 sum) -- M bytes gathered per neighbour instead of 4 D.
 * `opq` must hold exactly one code row per graph node, appended in the graph's internal-id order (the order
 * the vectors were added to the graph), with a coarseK == 1 model of the graph's dimension; rotate as in
 * cvtmi_opq_search.  Output as cvtmi_hnsw_search, distances = ADC distances. */
int cvtmi_hnsw_search_adc(cvtmi_hnsw_t h, cvtmi_opq_t opq, const float *q, int64_t nq, int rotate, int k, int ef,
                          float *dist, int64_t *labels);
int cvtmi_hnsw_search_adc_dev(cvtmi_hnsw_t h, cvtmi_opq_t opq, const float *q, int64_t nq, int rotate, int k, int ef,
                              float *dist, int64_t *labels, void *stream);
/* cvtmi_hnsw_search_adc followed by an exact re-rank: the ADC traversal returns its `rerank` best nodes (k <= rerank <= 1024), their
 * fp32 distances to the RAW query are computed from the graph's own vectors in the summation order of the reference's distance
 * functions, and the k smallest come back (k <= CVTMI_K_MAX); nodes with equal exact distances keep their ADC order.  16-byte codes steer
 * the traversal, full vectors are only touched for `rerank` nodes per query: the recall of the fp32 graph at a fraction of its
 * gather traffic. */
int cvtmi_hnsw_search_adc_rerank(cvtmi_hnsw_t h, cvtmi_opq_t opq, const float *q, int64_t nq, int rotate, int k, int ef,
                                 int rerank, float *dist, int64_t *labels);
int cvtmi_hnsw_search_adc_rerank_dev(cvtmi_hnsw_t h, cvtmi_opq_t opq, const float *q, int64_t nq, int rotate, int k, int ef,
                                     int rerank, float *dist, int64_t *labels, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CVTMI_H */
