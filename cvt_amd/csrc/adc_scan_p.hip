// adc_scan_p.hip -- adc_scan16p: the 15-bit filter scan (adc_scan16q, adc_scan.hip) for models with M = 8 or M = 4, NATIVE:
// no padded copy of the rows, no look-ups spent on zero tables.  (opq/src/IVFOPQ.h:24-29: PQindex[16], any M <= 16; the
// reference's own test model is M = 8, opq/src/multi_frame_index_test.cpp.  Scores: opq/src/IVFOPQ.cpp:300-309.)
//
// Round 5 served M < 16 through the M = 16 kernel over rows padded to 16 code bytes: an M = 8 row cost sixteen look-ups, eight of
// them into all-zero tables (3.1-3.4 M queries/s at 10 000 x 1 M rows -- the M = 16 rate for half the work).  Here a lane's 16-byte
// load holds RPL = 16 / M consecutive rows AS THEY LIE IN MEMORY.  The table in LDS keeps its sixteen 16-byte slots per code value
// (64 KB, as for M = 16), filled with 16 / M COPIES of the M sub-spaces' entries: lane c = lane & 15 reads copy c / M, and within it
// walks the sub-spaces in its own rotated order m = (t + c) & (M - 1), t = 0 .. M - 1, so the sixteen lanes of a ds_read_b128 group
// always hit sixteen different slots -- conflict-free, like the M = 16 walk.  The rows are streamed from a pre-rotated copy (row r
// rotated left by ((r / RPL) & (M - 1)) bytes within its M bytes: byte t of the stored row is the code of sub-space (t + c) & (M - 1)),
// so a look-up's LDS address is still one v_perm_b32.  The M look-ups of a row are summed into the four packed accumulators, tested,
// and the same accumulators then take the next row of the load: 16 look-ups, 32 v_add3 and RPL tests per load -- RPL rows for the
// price of one M = 16 row.  Everything else is adc_scan16q's: 15-bit lower-bound tables (per-query scale over the M real tables),
// lazy selection on the integer keys between checkpoints, candidates re-summed exactly from the fp32 tables in the reference's
// m order (ExactFromLutBatchP) before they are ranked -- thresholds, kept sets and distances bit-identical to the reference's.
#include "adc_scan16.h"

namespace cvtmi {

// exact reference-order sums (IVFOPQ.cpp:302-306) of up to four buffer entries per lane: rows of MP code bytes, tables [nq][16][256]
// fp32 of which the first MP belong to the model (launch_lut with tables = 16)
template <int MP>
struct ExactFromLutBatchP {
    const uint8_t *rows;
    const float *lut_g;
    int K, nq, group;
    __device__ __forceinline__ void operator()(int q, unsigned long long (&e)[4], const bool (&need)[4]) const
    {
        int qi = group * SQ_QT + q;
        qi = qi < nq ? qi : nq - 1;
        const float *t = lut_g + (int64_t)qi * 16 * 256;
        uint32_t c[4][MP / 4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t row = need[r] ? (uint32_t)e[r] : 0u;
            const uint32_t *p = reinterpret_cast<const uint32_t *>(rows + (size_t)row * MP);
#pragma unroll
            for (int w = 0; w < MP / 4; ++w) c[r][w] = p[w];
        }
#pragma unroll
        for (int h = 0; h < 4; h += 2) {
            if (__ballot(need[h] || need[h + 1]) == 0) continue;  // wave-uniform
            float v[2][MP];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
#pragma unroll
                for (int m = 0; m < MP; ++m) {
                    const int j = (int)((c[h + r][m >> 2] >> (8 * (m & 3))) & 0xffu);
                    v[r][m] = t[m * 256 + j];  // padded to 256 entries (+inf past K): no predicate
                }
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                float s = 0.0f;
#pragma unroll
                for (int m = 0; m < MP; ++m) s = __fadd_rn(s, v[r][m]);
                if (need[h + r]) e[h + r] = ((unsigned long long)__float_as_uint(s) << 32) | (uint32_t)e[h + r];
            }
        }
    }
};

// The QT queries' tables of `group`, quantised into LDS: lut[code j][slot][q] u16, slot = m + MP * copy.  As scan16q_build_tables
// (adc_scan.hip) with the range / bias over the MP real tables only.  Whole workgroup; ends with a barrier.
template <int NT, int MP>
__device__ __forceinline__ void scan16p_build_tables(const ScanArgs &a, int group, uint32_t *lut, QuantParams &qp, uint32_t (*mx_bits)[16],
                                                     int *nonfinite, int *lazy)
{
    constexpr int QT = SQ_QT;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < QT * 16) {
        qp.mn_bits[tid >> 4][tid & 15] = 0x7f7fffffu;  // FLT_MAX
        mx_bits[tid >> 4][tid & 15] = 0u;
    }
    if (tid < QT) nonfinite[tid] = 0;
    __syncthreads();
    int qis[QT];
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        const int qi = group * QT + q;
        qis[q] = qi < a.nq ? qi : a.nq - 1;
    }
    auto entry = [&](int m, int j, float (&acc)[QT]) {
#pragma unroll
        for (int q = 0; q < QT; ++q) acc[q] = a.lut_g[((int64_t)qis[q] * 16 + m) * 256 + j];  // padded with +inf past K
    };
    for (int e = tid; e < MP * 256; e += NT) {  // pass A: range of the finite entries per (query, real sub-space)
        const int m = e >> 8, j = e & 255;
        float acc[QT];
        entry(m, j, acc);
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            const uint32_t bits = __float_as_uint(acc[q]);
            uint32_t lo = bits < 0x7f800000u ? bits : 0x7f7fffffu;
            uint32_t hi = bits < 0x7f800000u ? bits : 0u;
            if (__ballot(bits >= 0x7f800000u) != 0 && lane == 0) nonfinite[q] = 1;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const uint32_t l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
                lo = l2 < lo ? l2 : lo;
                hi = h2 > hi ? h2 : hi;
            }
            if (lane == 0) {
                atomicMin(&qp.mn_bits[q][m], lo);
                atomicMax(&mx_bits[q][m], hi);
            }
        }
    }
    __syncthreads();
    if (tid < QT) {
        const int q = tid;
        float range = 0.0f;
        double bias = 0.0;
        for (int m = 0; m < MP; ++m) {
            uint32_t lo = qp.mn_bits[q][m], hi = mx_bits[q][m];
            if (lo > hi) { lo = 0u; hi = 0u; }
            const float fl = __uint_as_float(lo), fh = __uint_as_float(hi);
            qp.mn[q][m] = fl;
            range += fh - fl;
            bias += (double)fl;
        }
        float scale = range > 0.0f ? range / (float)SQ_MAXSUM * 1.001f : 1.0f;
        if (!(scale > 1e-37f)) scale = 1e-37f;
        const float inv = 1.0f / scale;
        qp.inv_scale[q] = inv;
        qp.scale_eff[q] = 1.0 / (double)inv;
        qp.bias[q] = bias;
        // (the band of scan_compact_lazy_q is derived for sixteen entries of two units each; MP entries need less: valid as it is)
        const double sl = 34.0 + ceil(4e-6 * (32767.0 + bias * (double)inv));
        const bool lazy_ok = a.lazy && !nonfinite[q] && sl < 1024.0 && bias >= 0.0;
        qp.slack[q] = lazy_ok ? (uint32_t)sl : 0u;
        lazy[q] = lazy_ok ? 1 : 0;
    }
    __syncthreads();
    for (int e = tid; e < 16 * 256; e += NT) {  // pass B: every slot (copy c of sub-space m at slot m + MP c)
        const int slot = e >> 8, j = e & 255, m = slot & (MP - 1);
        float acc[QT];
        entry(m, j, acc);
        uint32_t qv[QT];
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            const float v = acc[q];
            int iv = 0;
            if (__float_as_uint(v) < 0x7f800000u) {
                const float f = __fmul_rn(__fsub_rn(v, qp.mn[q][m]), qp.inv_scale[q]);
                iv = (int)floorf(f) - 1;
                iv = iv < 0 ? 0 : (iv > SQ_MAXSUM ? SQ_MAXSUM : iv);
            }
            qv[q] = (uint32_t)iv;
        }
        *reinterpret_cast<uint4 *>(&lut[(j * 16 + slot) * (QT / 2)]) =
            make_uint4(qv[0] | (qv[1] << 16), qv[2] | (qv[3] << 16), qv[4] | (qv[5] << 16), qv[6] | (qv[7] << 16));
    }
    __syncthreads();
}

// the sixteen table reads of one 16-byte load (RPL rows of MP code bytes, pre-rotated): v[s * MP + t] = look-up t of row s
template <int MP>
__device__ __forceinline__ void scan16p_reads(const uint4 &ld, const uint32_t (&moffp)[MP / 4], const char *lut_b, uint4 (&v)[16])
{
    const uint32_t w[4] = { ld.x, ld.y, ld.z, ld.w };
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int t = i & (MP - 1);   // look-up within the row
        const uint32_t sel = 0x0c0c0000u | ((4u + (t & 3)) << 8) | (uint32_t)(t & 3);
        const uint32_t addr = __builtin_amdgcn_perm(w[i >> 2], moffp[t >> 2], sel);  // code*256 + slot*16
        v[i] = *reinterpret_cast<const uint4 *>(lut_b + addr);
    }
}
// the packed sums of row s of the load, on top of the start values (0, or 0x8000 - T per 16-bit field: see adc_scan16q)
template <int MP>
__device__ __forceinline__ void scan16p_sum(const uint4 (&v)[16], int s, uint32_t &s0, uint32_t &s1, uint32_t &s2, uint32_t &s3)
{
#pragma unroll
    for (int t = 0; t < MP; t += 2) {
        const uint4 &a = v[s * MP + t], &b = v[s * MP + t + 1];
        s0 = s0 + a.x + b.x; s1 = s1 + a.y + b.y; s2 = s2 + a.z + b.z; s3 = s3 + a.w + b.w;
    }
}

// the M look-ups of row s of a load: issue / accumulate separately (software pipelining across the rows of a load)
template <int MP>
__device__ __forceinline__ void scan16p_issue(const uint4 &ld, int s, const uint32_t (&moffp)[MP / 4], const char *lut_b, uint4 (&v)[MP])
{
    const uint32_t w[4] = { ld.x, ld.y, ld.z, ld.w };
#pragma unroll
    for (int t = 0; t < MP; ++t) {
        const uint32_t sel = 0x0c0c0000u | ((4u + (t & 3)) << 8) | (uint32_t)(t & 3);
        const uint32_t addr = __builtin_amdgcn_perm(w[(s * MP + t) >> 2], moffp[t >> 2], sel);  // code*256 + slot*16
        v[t] = *reinterpret_cast<const uint4 *>(lut_b + addr);
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <int MP>
__device__ __forceinline__ void scan16p_accum(const uint4 (&v)[MP], uint32_t &s0, uint32_t &s1, uint32_t &s2, uint32_t &s3)
{
#pragma unroll
    for (int t = 0; t < MP; t += 2) {
        s0 = s0 + v[t].x + v[t + 1].x; s1 = s1 + v[t].y + v[t + 1].y;
        s2 = s2 + v[t].z + v[t + 1].z; s3 = s3 + v[t].w + v[t + 1].w;
    }
}

// the M look-ups of row s of a load, summed on top of the start values
template <int MP>
__device__ __forceinline__ void scan16p_row(const uint4 &ld, int s, const uint32_t (&moffp)[MP / 4], const char *lut_b,
                                            uint32_t &s0, uint32_t &s1, uint32_t &s2, uint32_t &s3)
{
    const uint32_t w[4] = { ld.x, ld.y, ld.z, ld.w };
    uint4 v[MP];
#pragma unroll
    for (int t = 0; t < MP; ++t) {
        const uint32_t sel = 0x0c0c0000u | ((4u + (t & 3)) << 8) | (uint32_t)(t & 3);
        const uint32_t addr = __builtin_amdgcn_perm(w[(s * MP + t) >> 2], moffp[t >> 2], sel);  // code*256 + slot*16
        v[t] = *reinterpret_cast<const uint4 *>(lut_b + addr);
    }
#pragma unroll
    for (int t = 0; t < MP; t += 2) {
        s0 = s0 + v[t].x + v[t + 1].x; s1 = s1 + v[t].y + v[t + 1].y;
        s2 = s2 + v[t].z + v[t + 1].z; s3 = s3 + v[t].w + v[t + 1].w;
    }
}

// first thresholds from a histogram of the split's first SQ_SEED_CHUNKS loads (64 RPL rows each): scan16q_seed for packed rows
template <int NT, int MP, class Load>
__device__ __forceinline__ void scan16p_seed(int k, const Load &load1k, const uint32_t (&moffp)[MP / 4], const char *lut_b, uint32_t *hist,
                                             const QuantParams &qp, const int *lazy, uint32_t *thr_x, uint32_t *thr_pk)
{
    constexpr int QT = SQ_QT, NW = NT / 64, RPL = 16 / MP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < QT * 256; i += NT) hist[i] = 0;
    __syncthreads();
    for (uint32_t ch = wave; ch < SQ_SEED_CHUNKS; ch += NW) {
        const uint4 ld = load1k(ch);
        uint4 v[16];
        scan16p_reads<MP>(ld, moffp, lut_b, v);
#pragma unroll
        for (int s = 0; s < RPL; ++s) {
            uint32_t sm[4] = { 0, 0, 0, 0 };
            scan16p_sum<MP>(v, s, sm[0], sm[1], sm[2], sm[3]);
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                const uint32_t sq = (sm[q >> 1] >> (16 * (q & 1))) & 0xffffu;
                atomicAdd(&hist[q * 256 + (sq >> 7)], 1u);
            }
        }
    }
    __syncthreads();
    if (wave < QT && lazy[wave]) {
        const int q = wave;
        uint32_t c4[4], mine = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { c4[j] = hist[q * 256 + lane * 4 + j]; mine += c4[j]; }
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        const unsigned long long reach = __ballot(incl >= (uint32_t)k);
        if (reach) {  // wave-uniform
            const int l0 = __ffsll((long long)reach) - 1;
            if (lane == l0) {
                uint32_t cum = incl - mine;
                int b = lane * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    cum += c4[j];
                    if (cum >= (uint32_t)k) { b = lane * 4 + j; break; }
                }
                uint32_t t = ((uint32_t)(b + 1) << 7) + qp.slack[q];
                t = t < 32767u ? t : 32767u;
                const uint32_t have = thr_x[q];
                t = have < t ? have : t;
                thr_x[q] = t;
                reinterpret_cast<uint16_t *>(thr_pk)[q] = (uint16_t)t;
            }
        }
    }
    __syncthreads();
}

template <int NT, int MP>
__global__ __launch_bounds__(NT, NT / 128) void adc_scan16p_kernel(const ScanArgs a)
{
    constexpr int QT = SQ_QT, RPL = 16 / MP;
    static_assert(MP == 8 || MP == 4, "packed rows: M = 8 (two per load) or M = 4 (four)");
    __shared__ __attribute__((aligned(16))) uint32_t lut[256 * 16 * QT / 2];  // u16 [code j][slot][q]: 16 B per (j, slot)
    __shared__ TopKShared<QT, SQ_CAP> tk;
    __shared__ QuantParams qp;
    __shared__ struct { int stop, done_waves; uint32_t next_chunk; uint32_t thr_pk[QT / 2]; int lazy[QT]; int nonfinite[QT]; } ck;

    int group, split, my_splits = a.splits;
    int64_t my_rps = a.rows_per_split;
    {
        int b = blockIdx.x, g0 = 0;
        if (b >= a.groups_a * a.splits) {  // region B: the tail groups, split finer (dispatched last)
            b -= a.groups_a * a.splits;
            g0 = a.groups_a;
            my_splits = a.splits_b;
            my_rps = a.rows_per_split_b;
        }
        if ((my_splits & 7) == 0) {
            const int s8 = my_splits >> 3;
            const int xcd = b & 7, i = b >> 3;
            split = xcd + 8 * (i % s8);
            group = g0 + i / s8;
        } else {
            split = b % my_splits;
            group = g0 + b / my_splits;
        }
    }
    const int tid = threadIdx.x, lane = tid & 63;
    topk_init(tk);
    scan16p_build_tables<NT, MP>(a, group, lut, qp, reinterpret_cast<uint32_t(*)[16]>(&tk.buf[0][0]), ck.nonfinite, ck.lazy);

    const ExactFromLutBatchP<MP> fixb{ a.codes, a.lut_g, a.K, a.nq, group };
    const QuantThr thrx{ &qp };
    if (tid < QT / 2) {  // pass-all until k rows are known -- or what the other row splits of these queries have already established
        uint32_t t2[2] = { 32767u, 32767u };
        if (a.gthr) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int qi = group * QT + 2 * tid + h;
                if (qi < a.nq) {
                    const uint32_t g = __hip_atomic_load(&a.gthr[qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    t2[h] = g < t2[h] ? g : t2[h];
                }
            }
        }
        tk.thr_x[2 * tid] = t2[0]; tk.thr_x[2 * tid + 1] = t2[1];
        ck.thr_pk[tid] = t2[0] | (t2[1] << 16);
    }
    if (tid == 0) { ck.stop = 0; ck.done_waves = 0; ck.next_chunk = 0; }
    __syncthreads();

    const int64_t row_begin = (int64_t)split * my_rps;   // a multiple of 2048 rows: whole 1 KB loads, lane <-> (row / RPL) & 63
    int64_t row_end = row_begin + my_rps;
    row_end = row_end < a.n_rows ? row_end : a.n_rows;
    const uint32_t n_local = (uint32_t)(row_end > row_begin ? row_end - row_begin : 0);
    const char *rows_b = reinterpret_cast<const char *>(a.codes_rot) + row_begin * MP;
    // one 16-byte load per lane = RPL rows; a wave's load = 1 KB = 64 RPL rows.  The last load may read up to 1 KB past the split (the
    // next split's rows, or the kDevSlack bytes every device buffer carries); what is read there is never used (lrow < n_local)
    constexpr uint32_t WROWS = 64 * RPL;
    const uint32_t n_chunks = (n_local + WROWS - 1) / WROWS;
    const uint32_t last_chunk = n_chunks ? n_chunks - 1 : 0;
    const uint32_t lane16 = (uint32_t)lane * 16u;
    auto load_rows = [&](uint32_t chunk) -> uint4 {
        const uint32_t cc = chunk < last_chunk ? chunk : last_chunk;  // wave-uniform
        return *reinterpret_cast<const uint4 *>(rows_b + (size_t)cc * 1024u + lane16);
    };
    auto load1k = [&](uint32_t c1k) -> uint4 { return *reinterpret_cast<const uint4 *>(rows_b + (size_t)c1k * 1024u + lane16); };
    const uint32_t c = tid & 15;
    uint32_t moffp[MP / 4];
#pragma unroll
    for (int w = 0; w < MP / 4; ++w) {
        moffp[w] = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) moffp[w] |= (((((4 * w + b + c) & (MP - 1)) | (c & ~(uint32_t)(MP - 1)))) * 16u) << (8 * b);
    }
    const char *lut_b = reinterpret_cast<const char *>(lut);
    if (a.seed && n_chunks >= 4 * SQ_SEED_CHUNKS)  // workgroup-uniform
        scan16p_seed<NT, MP>(a.k, load1k, moffp, lut_b, reinterpret_cast<uint32_t *>(&tk.buf[0][0]), qp, ck.lazy, tk.thr_x, ck.thr_pk);

    // ---- main loop: adc_scan16q's protocol (waves run on their own between checkpoints; a full buffer stops everybody) ----
    constexpr int NW = NT / 64;
    const uint32_t next_chunk_addr = (uint32_t)(uintptr_t)&ck.next_chunk;
    auto grab_pair = [&]() -> uint32_t {
        uint32_t v;
        asm volatile("" : "=v"(v));
        if (lane == 0) asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(next_chunk_addr), "v"(1u) : "memory");
        return 2u * (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    };
    auto after = [&](uint32_t chunk) -> uint32_t { return (chunk & 1u) ? grab_pair() : chunk + 1u; };
    uint32_t it = grab_pair(), it_next = it + 1u, done = 0, done_for = 0xffffffffu;
    bool counted = false;
    uint4 cur, nxt;
    if (n_local) cur = load_rows(it);
    for (;;) {
        uint32_t tpk[QT / 2], bpk[QT / 2];
#pragma unroll
        for (int i = 0; i < QT / 2; ++i) { tpk[i] = ck.thr_pk[i]; bpk[i] = 0x80008000u - tpk[i]; }
        while (it < n_chunks) {
            const int stop_seen = ck.stop;
            const uint32_t base = it * WROWS;
            nxt = load_rows(it_next);
            bool failed_any = false;
            // row by row: the M look-ups of a row, its test, (rarely) its candidates -- then the same accumulators take the next row of the
            // load.  (All RPL rows' sums kept until one common test spilled registers inside this loop at M = 4: 64 registers per lane.)
            // (M = 4: the four reads of row s + 1 are issued before row s is summed and tested -- a row's own four reads are too few to
            //  cover the LDS latency; at M = 8 two rows in flight would take every register the kernel has)
            constexpr bool PIPE = MP == 4;
            uint4 vr[PIPE ? 2 : 1][MP];
            if constexpr (PIPE) scan16p_issue<MP>(cur, 0, moffp, lut_b, vr[0]);
#pragma unroll
            for (int s = 0; s < RPL; ++s) {
                uint32_t s0 = bpk[0], s1 = bpk[1], s2 = bpk[2], s3 = bpk[3];
                if constexpr (PIPE) {
                    if (s + 1 < RPL) scan16p_issue<MP>(cur, s + 1, moffp, lut_b, vr[(s + 1) & 1]);
                    scan16p_accum<MP>(vr[s & 1], s0, s1, s2, s3);
                } else {
                    scan16p_row<MP>(cur, s, moffp, lut_b, s0, s1, s2, s3);
                }
                const uint32_t sg = (~((s0 & s1) & (s2 & s3))) & 0x80008000u;   // a clear bit 15 = sum < T for that query
                if (__ballot(sg != 0)) {  // rare once the thresholds have tightened
                    if (done_for != it) { done = 0; done_for = it; }
                    bool failed = false;
                    uint32_t lrow = base + (uint32_t)lane * RPL + s;
                    asm volatile("" : "+v"(lrow));
                    if (sg != 0 && lrow < n_local) {
                        const uint32_t sums[4] = { s0 - bpk[0], s1 - bpk[1], s2 - bpk[2], s3 - bpk[3] };   // (field-wise: no borrow)
#pragma unroll
                        for (int q = 0; q < QT; ++q) {
                            const uint32_t bit = 1u << (s * QT + q);
                            const uint32_t sq = (sums[q >> 1] >> (16 * (q & 1))) & 0xffffu;
                            const uint32_t tq = (tpk[q >> 1] >> (16 * (q & 1))) & 0xffffu;
                            if (sq < tq && !(done & bit)) {
                                bool dummy = false;
                                if (topk_push<QT, SQ_CAP, SQ_TRIG>(tk, q, sq, (uint32_t)(row_begin + lrow), dummy)) done |= bit;
                                else failed = true;
                            }
                        }
                    }
                    failed_any |= __ballot(failed) != 0;
                }
            }
            if (failed_any) {  // some buffer is full: stop everyone, come back to this load after the compaction
                if (lane == 0) ck.stop = 1;
                break;
            }
            cur = nxt;
            it = it_next;
            it_next = after(it);
            if (__builtin_amdgcn_readfirstlane(stop_seen)) break;
        }
        if (it >= n_chunks && !counted) {
            counted = true;
            if (lane == 0) atomicAdd(&ck.done_waves, 1);
        }
        __syncthreads();  // checkpoint (A)
        bool need = ck.stop != 0;
#pragma unroll
        for (int q = 0; q < QT; ++q) need |= tk.cnt[q] >= SQ_TRIG;
        const bool all_done = ck.done_waves == NW;
        if (need) {  // workgroup-uniform
            const int wv = tid >> 6;
            const int keep_max = a.k + 48 < SQ_TRIG - 24 ? a.k + 48 : SQ_TRIG - 24;
            for (int q = wv; q < QT; q += NW) {  // one wave per query, in registers
                const int keep = ck.lazy[q] ? scan_compact_lazy_q<QT, SQ_CAP>(tk, q, a.k, fixb, thrx, qp.slack[q], keep_max, &ck.lazy[q])
                                            : topk_compact_wave_q<QT, SQ_CAP, false>(tk, q, a.k, fixb, thrx);
                if (lane == 0) {
                    tk.cnt[q] = keep;
                    if (a.gthr) {  // the row splits of a query tighten each other's filter (same tables -> same units)
                        const int qi = group * QT + q;
                        if (qi < a.nq) {
                            const uint32_t mine = tk.thr_x[q];
                            const uint32_t seen = atomicMin(&a.gthr[qi], mine);
                            tk.thr_x[q] = seen < mine ? seen : mine;
                        }
                    }
                }
            }
            __syncthreads();
            if (tid < QT / 2) ck.thr_pk[tid] = tk.thr_x[2 * tid] | (tk.thr_x[2 * tid + 1] << 16);
            if (tid == 0) ck.stop = 0;
        }
        __syncthreads();  // checkpoint (B)
        if (all_done && !need) break;
    }

    __syncthreads();
    topk_compact_wave<QT, SQ_CAP, NT, true>(tk, a.k, fixb, thrx);
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        const int qi = group * QT + q;
        if (qi >= a.nq) break;
        const int cnt = tk.cnt[q];
        const bool in_place = a.out_d != nullptr && group < a.groups_a;   // (out_d is only set when that region has one split)
        float *const pd = in_place ? a.out_d : a.part_d;
        int64_t *const pi = in_place ? a.out_id : a.part_id;
        const int64_t o = in_place ? (int64_t)qi * a.k : ((int64_t)qi * a.stride + split) * a.k;
        for (int i = tid; i < a.k; i += NT) {
            if (i < cnt) {
                const unsigned long long e = tk.buf[q][i];
                pd[o + i] = __uint_as_float((uint32_t)(e >> 32));
                pi[o + i] = a.id_base + (int64_t)(uint32_t)e;
            } else {
                pd[o + i] = __uint_as_float(0x7f800000u);
                pi[o + i] = -1;
            }
        }
        if (!in_place && split == 0 && my_splits < a.stride) {
            const int64_t o2 = ((int64_t)qi * a.stride + my_splits) * a.k;
            for (int i = tid; i < (a.stride - my_splits) * a.k; i += NT) {
                a.part_d[o2 + i] = __uint_as_float(0x7f800000u);
                a.part_id[o2 + i] = -1;
            }
        }
    }
}

int launch_adc_scan16p(const ScanArgs &a, int M, int64_t blocks, hipStream_t st)
{
    if (blocks <= 0) return CVTMI_OK;
    if (blocks > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "adc_scan16p: grid too large (%lld)", (long long)blocks);
    if (!a.codes_rot || !a.lut_g) return fail(CVTMI_EINVAL, "adc_scan16p: pre-rotated rows / table scratch missing");
    const dim3 g((unsigned)blocks);
    if (M == 8) hipLaunchKernelGGL((adc_scan16p_kernel<1024, 8>), g, dim3(1024), 0, st, a);
    else if (M == 4) hipLaunchKernelGGL((adc_scan16p_kernel<1024, 4>), g, dim3(1024), 0, st, a);
    else return fail(CVTMI_EUNSUPPORTED, "adc_scan16p: M=%d (8 or 4)", M);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

// packed pre-rotation: the 16-byte group g (= 16 / M rows) keeps its rows where they are, each rotated left by (g & (M - 1)) bytes
// within its M bytes: stored byte t of a row = its code byte (t + g) & (M - 1).  Groups [g0, g1).
template <int MP>
__global__ __launch_bounds__(kBlock) void rotate_codes_packed_kernel(const uint4 *__restrict__ codes, uint4 *__restrict__ out, int64_t g0, int64_t g1)
{
    const int64_t g = g0 + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (g >= g1) return;
    const uint4 v = codes[g];
    const uint32_t c = (uint32_t)g & (MP - 1);
    uint4 o;
    if constexpr (MP == 8) {
        const auto rot8 = [&](uint32_t lo, uint32_t hi, uint32_t &olo, uint32_t &ohi) {   // 64-bit value rotated right by 8 c bits
            const unsigned long long x = ((unsigned long long)hi << 32) | lo;
            const unsigned long long y = c ? ((x >> (8 * c)) | (x << (64 - 8 * c))) : x;
            olo = (uint32_t)y; ohi = (uint32_t)(y >> 32);
        };
        rot8(v.x, v.y, o.x, o.y);
        rot8(v.z, v.w, o.z, o.w);
    } else {
        const auto rot4 = [&](uint32_t x) -> uint32_t { return c ? ((x >> (8 * c)) | (x << (32 - 8 * c))) : x; };
        o.x = rot4(v.x); o.y = rot4(v.y); o.z = rot4(v.z); o.w = rot4(v.w);
    }
    out[g] = o;
}

// rows [row0, n) of an M = 8 / M = 4 index: the groups that hold them (a partly filled last group reads the slack behind the rows)
int launch_rotate_codes_packed(const uint8_t *codes, int M, uint8_t *codes_rot, int64_t row0, int64_t n, hipStream_t st)
{
    if (n <= row0) return CVTMI_OK;
    if (M != 8 && M != 4) return fail(CVTMI_EINVAL, "rotate_codes_packed: M=%d", M);
    const int rpl = 16 / M;
    const int64_t g0 = row0 / rpl, g1 = (n + rpl - 1) / rpl;
    const int64_t blocks = (g1 - g0 + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "rotate_codes_packed: too many rows");
    if (M == 8) hipLaunchKernelGGL(rotate_codes_packed_kernel<8>, dim3((unsigned)blocks), dim3(kBlock), 0, st, reinterpret_cast<const uint4 *>(codes),
                                   reinterpret_cast<uint4 *>(codes_rot), g0, g1);
    else hipLaunchKernelGGL(rotate_codes_packed_kernel<4>, dim3((unsigned)blocks), dim3(kBlock), 0, st, reinterpret_cast<const uint4 *>(codes),
                            reinterpret_cast<uint4 *>(codes_rot), g0, g1);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

}  // namespace cvtmi
