// sq8.hip -- int8 scalar quantisation kernels (cvtk::quant::Int8Quan + sq_train).
//
// Reference arithmetic (scalar_quantization/scalar_quantization/int8_quan.cc):
//   L2NormalizeVector :46-56   accum(double) += float(v*v) sequentially; sqrt in double;
//                              den = float(max(1e-12, norm)); v = v / den (fp32 divide)
//   Int8Encode        :79-92   xi = (x - vmin)/vdiff (0 when vdiff == 0); clamp to [0,1];
//                              byte = (int)(255 * xi)
//   Int8Decode(string):126-130 x = vmin + vdiff * (b + 0.5) / 255.0, evaluated in double
// sq_train.cpp:84-103: rows L2-normalised, then per-dimension min and max - min (faiss
// ScalarQuantizer default RS_minmax; that library is not vendored: parity unpinned).
//
// All of it is streaming work whose roofline is HBM (encode: 4d bytes in + d out per row), so the kernels
// are built to read every row exactly once and to keep the VALU cost per element below the HBM time:
//
//  * sq8_tile_kernel (d % 4 == 0, d <= 512): a 1024-thread workgroup owns 64-row tiles.  A tile is loaded
//    ONCE (16-byte coalesced loads into registers, the next tile already in flight) and parked in LDS,
//    where wave 0 folds the double-precision norms -- one lane per row, in index order, because the order
//    of the additions is part of bit-exactness -- before all waves encode it (or min/max-reduce it, for
//    training).
//  * IEEE division is ~11 VALU instructions and there are two per element.  Both divisors are shared (one
//    per row, one per column), so their correctly rounded reciprocals y = RN(1/b) are computed once and
//    each quotient costs three instructions: q = RN(a*y), e = fma(-b, q, a), RN(q + e*y).  That is RN(a/b)
//    whenever b's significand is not all ones and nothing leaves the normal range (Markstein's
//    theorem; tools/ubench/div_check.c brute-forces 1.1e9 quotients incl. every significand of b);
//    operands outside the guarded range take __fdiv_rn.
//  * decode divides by the constant 255.0 in double the same way (3 full-rate fp64 ops instead of a ddiv).
//  * rows of other widths take the two-pass kernels at the bottom (norms, then an element-wise pass).
#include <algorithm>
#include <atomic>

#include "div_rn.h"
#include "kernels.h"

namespace cvtmi {

// row streams in, code / row streams out: non-temporal (common.h).  4 M x 512: encode 5.12 -> 5.33 TB/s, decode 4.96 -> 5.19,
// normalising encode 5.30 -> 5.54 of traffic; training unchanged (5.84 without normalisation; with it the norm arithmetic binds)
#define SQ8_LD(p) ld_nt(p)
#define SQ8_ST(p, v) st_nt(p, v)

__device__ __forceinline__ uint32_t sq8_byte(float v, float lo, const DivBy &df)
{
    float xi = 0.0f;
    if (df.b != 0.0f) xi = div_rn(__fsub_rn(v, lo), df);
    if (xi < 0.0f) xi = 0.0f;
    if (xi > 1.0f) xi = 1.0f;
    return (uint32_t)(uint8_t)(int)__fmul_rn(255.0f, xi);
}

// ---- the single-pass tile kernel -------------------------------------------------------------------
constexpr int SQ_NT = 1024;     // threads per workgroup
constexpr int SQ_ROWS = 64;     // rows per tile = lanes of the folding wave
constexpr int SQ_PF = 8;        // float4 registers per thread and tile (d = 512: 64 rows x 128 float4 / 1024)

struct Sq8Args {
    const float *vmin, *vdiff;  // encode
    float *x;                   // rows (written back normalised when write_back)
    uint8_t *codes;
    uint32_t *kmin, *kmax;      // train: per-column ordered-uint keys
    int64_t n;
    int d, l2norm, write_back;
};

template <bool TRAIN>
__global__ __launch_bounds__(SQ_NT) void sq8_tile_kernel(const Sq8Args a)
{
    extern __shared__ float4 sq_tile[];  // [SQ_ROWS][CG + 1]
    __shared__ float den_s[SQ_ROWS], rcp_s[SQ_ROWS];
    __shared__ int ok_s[SQ_ROWS], seq_s[SQ_ROWS];
    const int CG = a.d >> 2;            // float4 per row
    const int TY = SQ_NT / CG;          // rows covered by one sweep of the workgroup
    const int tid = threadIdx.x;
    const int ty = tid / CG, tx = tid - ty * CG;
    const bool active = ty < TY;
    const int64_t n_tiles = (a.n + SQ_ROWS - 1) / SQ_ROWS;
    const float4 *x4 = reinterpret_cast<const float4 *>(a.x);

    float4 lo4 = make_float4(0, 0, 0, 0);
    DivBy df[4];
    float4 mn = make_float4(__uint_as_float(0x7f800000u), __uint_as_float(0x7f800000u), __uint_as_float(0x7f800000u),
                            __uint_as_float(0x7f800000u));
    float4 mx = make_float4(__uint_as_float(0xff800000u), __uint_as_float(0xff800000u), __uint_as_float(0xff800000u),
                            __uint_as_float(0xff800000u));
    if constexpr (!TRAIN) {
        if (active) {
            lo4 = reinterpret_cast<const float4 *>(a.vmin)[tx];
            const float4 d4 = reinterpret_cast<const float4 *>(a.vdiff)[tx];
            df[0] = div_by(d4.x); df[1] = div_by(d4.y); df[2] = div_by(d4.z); df[3] = div_by(d4.w);
        }
    }

    auto load_tile = [&](int64_t tile, float4 (&v)[SQ_PF]) {
#pragma unroll
        for (int i = 0; i < SQ_PF; ++i) {
            const int r = ty + i * TY;
            const int64_t row = tile * SQ_ROWS + r;
            v[i] = make_float4(0, 0, 0, 0);
            if (active && r < SQ_ROWS && row < a.n) v[i] = x4[row * CG + tx];
        }
    };

    // One register set: the tile in hand goes to LDS, then the registers take the next tile's loads, which stay
    // in flight during the fold and the encode (which reads the tile back from LDS).
    float4 pf[SQ_PF];
    int64_t tile = blockIdx.x;
    if (tile < n_tiles) load_tile(tile, pf);
    for (; tile < n_tiles; tile += gridDim.x) {
        lds_barrier();  // previous tile's readers are done
#pragma unroll
        for (int i = 0; i < SQ_PF; ++i) {
            const int r = ty + i * TY;
            if (active && r < SQ_ROWS) sq_tile[r * (CG + 1) + tx] = pf[i];
        }
        const int64_t tile_next = tile + gridDim.x;
        if (tile_next < n_tiles) load_tile(tile_next, pf);
        lds_barrier();
        if (a.l2norm) {
            // The reference adds the squares into one double in index order (int8_quan.cc:48-51).  That order only
            // matters through float(sqrt(sum)): sixteen lanes per row add their share in any order (each sum is within
            // 2^-43 of the other, 512 roundings of 2^-53 on either side), and when the float roots of sum (1 -+ 2^-42)
            // coincide the reference's root is that float, proven.  Otherwise (~4 rows in a million, and every non-finite
            // row) the row is summed again in index order below.
            {
                const int row = tid >> 4, part = tid & 15;
                const float4 *rowp = sq_tile + row * (CG + 1);
                double s0 = 0.0, s1 = 0.0;
                for (int j = part; j < CG; j += 32) {
                    const float4 t0 = rowp[j];
                    s0 += (double)__fmul_rn(t0.x, t0.x); s0 += (double)__fmul_rn(t0.y, t0.y);
                    s0 += (double)__fmul_rn(t0.z, t0.z); s0 += (double)__fmul_rn(t0.w, t0.w);
                    if (j + 16 < CG) {
                        const float4 t1 = rowp[j + 16];
                        s1 += (double)__fmul_rn(t1.x, t1.x); s1 += (double)__fmul_rn(t1.y, t1.y);
                        s1 += (double)__fmul_rn(t1.z, t1.z); s1 += (double)__fmul_rn(t1.w, t1.w);
                    }
                }
                double sum = s0 + s1;
#pragma unroll
                for (int o = 8; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 64);
                if (part == 0 && row < SQ_ROWS) {
                    const double rlo = __dsqrt_rn(sum * (1.0 - 0x1p-42)), rhi = __dsqrt_rn(sum * (1.0 + 0x1p-42));
                    const float flo = (float)(rlo > 1e-12 ? rlo : 1e-12), fhi = (float)(rhi > 1e-12 ? rhi : 1e-12);
                    const bool proven = flo == fhi;  // false for NaN
                    seq_s[row] = proven ? 0 : 1;
                    if (proven) {
                        const DivBy dd = div_by(flo);
                        den_s[row] = dd.b;
                        rcp_s[row] = dd.y;
                        ok_s[row] = dd.ok;
                    }
                }
            }
            lds_barrier();
            if (tid < SQ_ROWS && seq_s[tid]) {  // index order, one lane per row
                const float4 *rowp = sq_tile + tid * (CG + 1);
                double accum = 0.0;
                // The additions form one dependent fp64 chain per row; everything else (LDS reads, squares,
                // conversions) is done one group of 8 elements ahead so that the chain never waits for it.
                auto squares = [&](int j, double (&o)[8]) {
                    const float4 t0 = rowp[j], t1 = rowp[j + 1];
                    o[0] = (double)__fmul_rn(t0.x, t0.x); o[1] = (double)__fmul_rn(t0.y, t0.y);
                    o[2] = (double)__fmul_rn(t0.z, t0.z); o[3] = (double)__fmul_rn(t0.w, t0.w);
                    o[4] = (double)__fmul_rn(t1.x, t1.x); o[5] = (double)__fmul_rn(t1.y, t1.y);
                    o[6] = (double)__fmul_rn(t1.z, t1.z); o[7] = (double)__fmul_rn(t1.w, t1.w);
                };
                int j = 0;
                if (CG >= 2) {
                    double dn[8];
                    squares(0, dn);
                    for (j = 2; j + 2 <= CG; j += 2) {
                        double dc[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) dc[u] = dn[u];
                        squares(j, dn);
#pragma unroll
                        for (int u = 0; u < 8; ++u) accum += dc[u];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) accum += dn[u];
                }
                for (; j < CG; ++j) {
                    const float4 t = rowp[j];
                    accum += (double)__fmul_rn(t.x, t.x);
                    accum += (double)__fmul_rn(t.y, t.y);
                    accum += (double)__fmul_rn(t.z, t.z);
                    accum += (double)__fmul_rn(t.w, t.w);
                }
                const double nrm = __dsqrt_rn(accum);
                const DivBy dd = div_by((float)(nrm > 1e-12 ? nrm : 1e-12));
                den_s[tid] = dd.b;
                rcp_s[tid] = dd.y;
                ok_s[tid] = dd.ok;
            }
            lds_barrier();
        }
#pragma unroll 2
        for (int i = 0; i < SQ_PF; ++i) {
            const int r = ty + i * TY;
            const int64_t row = tile * SQ_ROWS + r;
            if (!(active && r < SQ_ROWS && row < a.n)) continue;
            float4 v = sq_tile[r * (CG + 1) + tx];
            if (a.l2norm) {
                DivBy dd;
                dd.b = den_s[r]; dd.y = rcp_s[r]; dd.ok = ok_s[r] != 0;
                v.x = div_rn(v.x, dd); v.y = div_rn(v.y, dd); v.z = div_rn(v.z, dd); v.w = div_rn(v.w, dd);
                if (!TRAIN && a.write_back) reinterpret_cast<float4 *>(a.x)[row * CG + tx] = v;
            }
            if constexpr (TRAIN) {
                mn.x = v.x < mn.x ? v.x : mn.x; mn.y = v.y < mn.y ? v.y : mn.y;
                mn.z = v.z < mn.z ? v.z : mn.z; mn.w = v.w < mn.w ? v.w : mn.w;
                mx.x = v.x > mx.x ? v.x : mx.x; mx.y = v.y > mx.y ? v.y : mx.y;
                mx.z = v.z > mx.z ? v.z : mx.z; mx.w = v.w > mx.w ? v.w : mx.w;
            } else {
                const uint32_t w = sq8_byte(v.x, lo4.x, df[0]) | (sq8_byte(v.y, lo4.y, df[1]) << 8) |
                                   (sq8_byte(v.z, lo4.z, df[2]) << 16) | (sq8_byte(v.w, lo4.w, df[3]) << 24);
                reinterpret_cast<uint32_t *>(a.codes)[row * CG + tx] = w;
            }
        }
    }
    if constexpr (TRAIN) {
        if (active) {  // min / max are order-independent: one atomic per thread and column
            const float lo[4] = { mn.x, mn.y, mn.z, mn.w }, hi[4] = { mx.x, mx.y, mx.z, mx.w };
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                atomicMin(&a.kmin[4 * tx + c], f32_key(lo[c]));
                atomicMax(&a.kmax[4 * tx + c], f32_key(hi[c]));
            }
        }
    }
}

static bool sq8_tile_ok(int d, const void *x, const void *codes, const void *vmin, const void *vdiff);
static bool sq8_wave_takes(int d, int64_t n, const void *x, const void *codes, const void *vmin, const void *vdiff);
// true: the launchers below need no per-row norm scratch for this call (the tile kernel, or the wave-per-row kernels)
bool sq8_single_pass(int d, const void *x, const void *codes, const void *vmin, const void *vdiff, int64_t n)
{
    return sq8_tile_ok(d, x, codes, vmin, vdiff) || sq8_wave_takes(d, n, x, codes, vmin, vdiff);
}
static bool sq8_tile_ok(int d, const void *x, const void *codes, const void *vmin, const void *vdiff)
{
    return d >= 4 && d <= 512 && (d & 3) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)codes & 3) == 0 &&
           ((uintptr_t)vmin & 15) == 0 && ((uintptr_t)vdiff & 15) == 0;
}

template <bool TRAIN>
static int launch_sq8_tile(const Sq8Args &a, hipStream_t st)
{
    const size_t lds = (size_t)SQ_ROWS * ((a.d >> 2) + 1) * sizeof(float4);
    // set on every call: the attribute belongs to the (function, device) pair, and a process may hold handles on several devices
    CVTMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sq8_tile_kernel<TRAIN>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 129 * 16));
    const int64_t n_tiles = (a.n + SQ_ROWS - 1) / SQ_ROWS;
    int64_t blocks = lds > 72 * 1024 ? 256 : 512;  // one or two workgroups per CU
    if (blocks > n_tiles) blocks = n_tiles;
    hipLaunchKernelGGL(sq8_tile_kernel<TRAIN>, dim3((unsigned)blocks), dim3(SQ_NT), lds, st, a);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

// ---- two-pass kernels for the other row widths ----------------------------------------------------
constexpr int NORM_ROWS = 64;   // rows per workgroup
constexpr int NORM_COLS = 256;  // columns staged per step

__global__ __launch_bounds__(kBlock) void sq8_rownorm_kernel(const float *__restrict__ x, int64_t n, int d,
                                                             float *__restrict__ den)
{
    __shared__ float tile[NORM_ROWS][NORM_COLS + 1];
    const int64_t row0 = (int64_t)blockIdx.x * NORM_ROWS;
    const int tid = threadIdx.x;
    double accum = 0.0;
    for (int c0 = 0; c0 < d; c0 += NORM_COLS) {
        const int cl = (d - c0) < NORM_COLS ? (d - c0) : NORM_COLS;
        __syncthreads();
        for (int i = tid; i < NORM_ROWS * NORM_COLS; i += kBlock) {
            const int r = i / NORM_COLS, c = i - r * NORM_COLS;
            float v = 0.0f;
            if (c < cl && row0 + r < n) v = x[(row0 + r) * d + c0 + c];
            tile[r][c] = v;
        }
        __syncthreads();
        if (tid < NORM_ROWS) {
            for (int c = 0; c < cl; ++c) {
                const float v = tile[tid][c];
                accum += (double)__fmul_rn(v, v);
            }
        }
    }
    if (tid < NORM_ROWS && row0 + tid < n) {
        const double nrm = __dsqrt_rn(accum);
        den[row0 + tid] = (float)(nrm > 1e-12 ? nrm : 1e-12);
    }
}

int launch_sq8_rownorm(const float *x, int64_t n, int d, float *den, hipStream_t st)
{
    if (n <= 0) return CVTMI_OK;
    const int64_t blocks = (n + NORM_ROWS - 1) / NORM_ROWS;
    if (blocks > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "sq8: n too large");
    hipLaunchKernelGGL(sq8_rownorm_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, x, n, d, den);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

constexpr int EW_ROWS = 16;  // rows per workgroup of the element-wise kernels

__global__ __launch_bounds__(kBlock) void sq8_encode_kernel(const float *__restrict__ vmin,
                                                            const float *__restrict__ vdiff, int d, float *x, int64_t n,
                                                            const float *__restrict__ den, int write_back,
                                                            uint8_t *__restrict__ codes)
{
    for (int64_t r0 = (int64_t)blockIdx.x * EW_ROWS; r0 < n; r0 += (int64_t)gridDim.x * EW_ROWS) {
        for (int c = threadIdx.x; c < d; c += kBlock) {  // lanes walk adjacent columns: coalesced
            const float lo = vmin[c];
            const DivBy df = div_by(vdiff[c]);
            for (int r = 0; r < EW_ROWS && r0 + r < n; ++r) {
                const int64_t e = (r0 + r) * d + c;
                float v = x[e];
                if (den) {
                    v = __fdiv_rn(v, den[r0 + r]);
                    if (write_back) x[e] = v;
                }
                codes[e] = (uint8_t)sq8_byte(v, lo, df);
            }
        }
    }
}

int launch_sq8_encode(const float *vmin, const float *vdiff, int d, float *x, int64_t n, const float *den,
                      int write_back, uint8_t *codes, hipStream_t st)
{
    if (n <= 0) return CVTMI_OK;
    int64_t blocks = (n + EW_ROWS - 1) / EW_ROWS;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(sq8_encode_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, vmin, vdiff, d, x, n, den,
                       write_back, codes);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

static int launch_sq8_encode_wave(const float *vmin, const float *vdiff, int d, float *x, int64_t n, int l2norm, int write_back,
                                  uint8_t *codes, hipStream_t st);
// row widths the wave-per-row filter kernels take: 256 NF floats with NF in {1, 2, 3, 4, 6, 8} -- 256 ... 2048-d (round 5: beyond 512-d
// the rows used to go through a row-norm pass + the generic kernels: 0.8 TB/s with normalisation, 3.3 without)
static bool sq8_wave_width(int d) { return d == 256 || d == 512 || d == 768 || d == 1024 || d == 1536 || d == 2048; }
static bool sq8_wave_aligned(const void *x, const void *codes, const void *vmin, const void *vdiff)
{
    return ((uintptr_t)x & 15) == 0 && ((uintptr_t)codes & 3) == 0 && ((uintptr_t)vmin & 15) == 0 && ((uintptr_t)vdiff & 15) == 0;
}
static bool sq8_filter_on();
static int g_sq8_encode_wave = 1;
// (conservative: both the training and the encode dispatch take the wave kernels under these conditions)
static bool sq8_wave_takes(int d, int64_t n, const void *x, const void *codes, const void *vmin, const void *vdiff)
{
    return g_sq8_encode_wave && sq8_filter_on() && sq8_wave_width(d) && n >= (d > 512 ? 1 : 4096) && sq8_wave_aligned(x, codes, vmin, vdiff);
}   // cvtmi_set_tuning("sq8_encode_wave"): 0 = the tile kernel for every width
void set_sq8_encode_wave(int v) { g_sq8_encode_wave = v; }

static int launch_sq8_encode_group(const float *vmin, const float *vdiff, int d, float *x, int64_t n, int l2norm, int write_back, uint8_t *codes, hipStream_t st);
int launch_sq8_encode_rows(const float *vmin, const float *vdiff, int d, float *x, int64_t n, int l2norm, int write_back,
                           uint8_t *codes, float *den_scratch, hipStream_t st)
{
    if (n <= 0) return CVTMI_OK;
    // 64-d / 128-d rows (round 6): several rows per wave through the same decision filter (sq8_encode_group_f_kernel)
    if (g_sq8_encode_wave && sq8_filter_on() && (d == 64 || d == 128) && n >= 4096 && sq8_wave_aligned(x, codes, vmin, vdiff))
        return launch_sq8_encode_group(vmin, vdiff, d, x, n, l2norm, write_back, codes, st);
    // (rows wider than the tile kernel's 512 floats: at every row count -- the two-pass kernels' norm pass alone costs 100 us for ONE row)
    if (g_sq8_encode_wave && sq8_wave_width(d) && n >= (d > 512 ? 1 : 4096) && sq8_wave_aligned(x, codes, vmin, vdiff) && (d <= 512 || sq8_filter_on()))
        return launch_sq8_encode_wave(vmin, vdiff, d, x, n, l2norm, write_back, codes, st);
    if (sq8_tile_ok(d, x, codes, vmin, vdiff)) {
        Sq8Args a{};
        a.vmin = vmin; a.vdiff = vdiff; a.x = x; a.codes = codes; a.n = n; a.d = d; a.l2norm = l2norm;
        a.write_back = write_back;
        return launch_sq8_tile<false>(a, st);
    }
    if (l2norm) {
        if (!den_scratch) return fail(CVTMI_EINVAL, "sq8_encode: norm scratch missing");
        CVTMI_TRY(launch_sq8_rownorm(x, n, d, den_scratch, st));
    }
    return launch_sq8_encode(vmin, vdiff, d, x, n, l2norm ? den_scratch : nullptr, write_back, codes, st);
}

// x = vmin + vdiff * (b + 0.5) / 255.0 in double (int8_quan.cc:126-130).  vdiff * (b + 0.5) is exact in
// double (24 + 9 bits); the division by the constant uses RN(1/255) and two fmas (255's significand is
// not all ones), guarded to the range where no intermediate can underflow.
// mode 1: the 8-bit codec of faiss 1.5.3 that Int8Decode(uint8_t*) / Int8DecodeFaiss delegate to (int8_quan.cc:96-115 -> sq.decode):
// fp32 throughout, xi = (code + 0.5f) / 255.0f, x = vmin + xi * vdiff, product and sum rounded separately
__device__ __forceinline__ float sq8_decode_one(float lo, float df, bool fast, uint32_t b, int mode = 0)
{
    if (mode == 1) return __fadd_rn(lo, __fmul_rn(__fdiv_rn(__fadd_rn((float)b, 0.5f), 255.0f), df));
    const double t0 = __dmul_rn((double)df, __dadd_rn((double)b, 0.5));
    double t;
    if (fast) {
        const double y = 1.0 / 255.0;  // constant-folded, correctly rounded
        const double q = __dmul_rn(t0, y);
        const double e = __fma_rn(-255.0, q, t0);
        t = __fma_rn(e, y, q);
    } else {
        t = __ddiv_rn(t0, 255.0);
    }
    return (float)__dadd_rn((double)lo, t);
}

__global__ __launch_bounds__(kBlock) void sq8_decode_kernel(const float *__restrict__ vmin,
                                                            const float *__restrict__ vdiff, int d,
                                                            const uint8_t *__restrict__ codes, int64_t n,
                                                            float *__restrict__ x, int mode)
{
    // d % 4 == 0 and aligned: a thread owns 4 adjacent columns (one dword of codes, one float4 out)
    const int CG = d >> 2;
    for (int64_t r0 = (int64_t)blockIdx.x * EW_ROWS; r0 < n; r0 += (int64_t)gridDim.x * EW_ROWS) {
        for (int c = threadIdx.x; c < CG; c += kBlock) {
            const float4 lo = reinterpret_cast<const float4 *>(vmin)[c];
            const float4 df = reinterpret_cast<const float4 *>(vdiff)[c];
            const float dfa[4] = { df.x, df.y, df.z, df.w }, loa[4] = { lo.x, lo.y, lo.z, lo.w };
            bool fast[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float m = fabsf(dfa[u]);
                fast[u] = (m >= 0x1p-100f && m <= 0x1p100f) || m == 0.0f;
            }
#pragma unroll 4
            for (int r = 0; r < EW_ROWS; ++r) {
                if (r0 + r >= n) break;
                const uint32_t w = reinterpret_cast<const uint32_t *>(codes)[(r0 + r) * CG + c];
                float4 o;
                o.x = sq8_decode_one(loa[0], dfa[0], fast[0], w & 0xffu, mode);
                o.y = sq8_decode_one(loa[1], dfa[1], fast[1], (w >> 8) & 0xffu, mode);
                o.z = sq8_decode_one(loa[2], dfa[2], fast[2], (w >> 16) & 0xffu, mode);
                o.w = sq8_decode_one(loa[3], dfa[3], fast[3], w >> 24, mode);
                reinterpret_cast<float4 *>(x)[(r0 + r) * CG + c] = o;
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void sq8_decode_any_kernel(const float *__restrict__ vmin,
                                                                const float *__restrict__ vdiff, int d,
                                                                const uint8_t *__restrict__ codes, int64_t n,
                                                                float *__restrict__ x, int mode)
{
    for (int64_t r0 = (int64_t)blockIdx.x * EW_ROWS; r0 < n; r0 += (int64_t)gridDim.x * EW_ROWS) {
        for (int c = threadIdx.x; c < d; c += kBlock) {
            const float lo = vmin[c], df = vdiff[c];
            for (int r = 0; r < EW_ROWS && r0 + r < n; ++r)
                x[(r0 + r) * d + c] = sq8_decode_one(lo, df, false, codes[(r0 + r) * d + c], mode);
        }
    }
}

// Decode through a table.  A column has only 256 possible outputs, and the per-element arithmetic above is ~10 fp64
// instructions -- more vector time than the HBM time of the 5 bytes the element moves.  A workgroup therefore takes a slab of
// 128 columns, evaluates all 128 x 256 outputs once (the arithmetic above, bit for bit) into LDS as lut[j][byte][l] (column
// 4 l + j of the slab: the 32 lanes of a half wave read 32 consecutive words whatever their bytes are -- no bank conflict), and
// streams its share of the rows: a half wave per row, a lane loads one dword of codes, looks its four bytes up and stores one
// float4.  128 KB of LDS, one 512-thread workgroup per CU, DEC_U rows per lane in flight.
constexpr int DEC_NT = 512, DEC_U = 8, DEC_SLAB = 128;
__global__ __launch_bounds__(DEC_NT) void sq8_decode_lut_kernel(const float *__restrict__ vmin, const float *__restrict__ vdiff, int d,
                                                                const uint8_t *__restrict__ codes, int64_t n, float *__restrict__ x, int row_splits, int mode)
{
    extern __shared__ float dec_lut[];   // [4][256][32]
    const int tid = threadIdx.x;
    const int slab = blockIdx.x / row_splits, rs = blockIdx.x % row_splits;
    const int col0 = slab * DEC_SLAB;
    for (int e = tid; e < 4 * 256 * 32; e += DEC_NT) {
        const int l = e & 31, b = (e >> 5) & 255, j = e >> 13;
        const int c = col0 + 4 * l + j;
        float v = 0.0f;
        if (c < d) {
            const float df = vdiff[c];
            const float m = fabsf(df);
            v = sq8_decode_one(vmin[c], df, (m >= 0x1p-100f && m <= 0x1p100f) || m == 0.0f, (uint32_t)b, mode);
        }
        dec_lut[e] = v;
    }
    __syncthreads();
    const int l = tid & 31, hw = tid >> 5;                     // 16 half waves
    if (col0 + 4 * l >= d) return;                             // (d % 4 == 0: a lane's four columns are all in or all out)
    const int64_t rows_per = (n + row_splits - 1) / row_splits;
    const int64_t r_begin = rows_per * rs, r_end = r_begin + rows_per < n ? r_begin + rows_per : n;
    const float *t0 = dec_lut + l, *t1 = dec_lut + 8192 + l, *t2 = dec_lut + 16384 + l, *t3 = dec_lut + 24576 + l;
    for (int64_t r0 = r_begin + hw; r0 < r_end; r0 += (int64_t)16 * DEC_U) {
        uint32_t w[DEC_U];
#pragma unroll
        for (int u = 0; u < DEC_U; ++u) {
            const int64_t r = r0 + 16 * u;
            w[u] = r < r_end ? SQ8_LD(reinterpret_cast<const uint32_t *>(codes + r * d + col0 + 4 * l)) : 0u;
        }
#pragma unroll
        for (int u = 0; u < DEC_U; ++u) {
            const int64_t r = r0 + 16 * u;
            const float4 o = make_float4(t0[(w[u] & 0xffu) * 32], t1[((w[u] >> 8) & 0xffu) * 32], t2[((w[u] >> 16) & 0xffu) * 32],
                                         t3[(w[u] >> 24) * 32]);
            if (r < r_end) SQ8_ST(reinterpret_cast<float4 *>(x + r * d + col0 + 4 * l), o);
        }
    }
}

int launch_sq8_decode(const float *vmin, const float *vdiff, int d, const uint8_t *codes, int64_t n, float *x,
                      hipStream_t st, int mode)
{
    if (n <= 0) return CVTMI_OK;
    if ((d & 3) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)codes & 3) == 0 && n >= 16384) {
        const int slabs = (d + DEC_SLAB - 1) / DEC_SLAB;
        int rs = (256 + slabs - 1) / slabs;                    // one workgroup per CU
        if ((int64_t)rs * 2048 > n) rs = (int)std::max<int64_t>(1, n / 2048);
        const size_t lds = 4 * 256 * 32 * sizeof(float);
        CVTMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sq8_decode_lut_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(sq8_decode_lut_kernel, dim3((unsigned)(slabs * rs)), dim3(DEC_NT), lds, st, vmin, vdiff, d, codes, n, x, rs, mode);
        CVTMI_HIP(hipGetLastError());
        return CVTMI_OK;
    }
    int64_t blocks = (n + EW_ROWS - 1) / EW_ROWS;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const bool vec = (d & 3) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)codes & 3) == 0 &&
                     ((uintptr_t)vmin & 15) == 0 && ((uintptr_t)vdiff & 15) == 0;
    if (vec) hipLaunchKernelGGL(sq8_decode_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, vmin, vdiff, d, codes, n, x, mode);
    else hipLaunchKernelGGL(sq8_decode_any_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, vmin, vdiff, d, codes, n, x, mode);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

// ---- train: per-dimension min / max over (normalised) rows; min and max are order-independent ----
__global__ void sq8_train_init_kernel(uint32_t *kmin, uint32_t *kmax, int d)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < d) {
        kmin[i] = f32_key(__uint_as_float(0x7f800000u));  // +inf
        kmax[i] = f32_key(__uint_as_float(0xff800000u));  // -inf
    }
}

constexpr int TRAIN_ROWS = 512;  // rows per workgroup

__global__ __launch_bounds__(kBlock) void sq8_train_kernel(const float *__restrict__ x, int64_t n, int d,
                                                           const float *__restrict__ den, uint32_t *kmin,
                                                           uint32_t *kmax)
{
    const int64_t r0 = (int64_t)blockIdx.x * TRAIN_ROWS;
    int64_t r1 = r0 + TRAIN_ROWS;
    r1 = r1 < n ? r1 : n;
    for (int c = threadIdx.x; c < d; c += kBlock) {  // lanes walk adjacent columns: coalesced rows
        float lo = __uint_as_float(0x7f800000u), hi = __uint_as_float(0xff800000u);
        for (int64_t r = r0; r < r1; ++r) {
            float v = x[r * d + c];
            if (den) v = __fdiv_rn(v, den[r]);
            if (v < lo) lo = v;
            if (v > hi) hi = v;
        }
        atomicMin(&kmin[c], f32_key(lo));
        atomicMax(&kmax[c], f32_key(hi));
    }
}

// ---- train with normalisation, d = 256 / 512: one WAVE per row, no LDS tile, no barriers in the row loop --------------
// The tile kernel above is built for encode (every element is read back from LDS by another thread).  Training only needs
// min / max of x / |x| per column: a wave reads a whole row (NF float4 per lane, one coalesced 1-2 KB piece), adds the squares
// in double (any order: the same proof as in the tile kernel -- when the float roots of sum (1 -+ 2^-42) coincide that float is
// the reference's root; otherwise lane order is replayed, rare), divides its own elements by the row norm and keeps running
// min / max of ITS columns in registers.  Rows are prefetched two ahead; nothing synchronises until the end, where the waves
// of a workgroup fold their column extremes through LDS before one atomic per column.  HBM traffic = the 4 d bytes per row.
template <int NF>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void sq8_train_wave_kernel(const float *__restrict__ x, int64_t n, uint32_t *kmin,
                                                                uint32_t *kmax)
{
    constexpr int CG = 64 * NF, D = 4 * CG;
    __shared__ uint32_t smin[D], smax[D];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < D; c += kBlock) { smin[c] = 0xffffffffu; smax[c] = 0u; }
    __syncthreads();
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    const int64_t nw = (int64_t)gridDim.x * (kBlock / 64), w0 = (int64_t)blockIdx.x * (kBlock / 64) + wave;
    float4 mn[NF], mx[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        mn[i] = make_float4(__uint_as_float(0x7f800000u), __uint_as_float(0x7f800000u), __uint_as_float(0x7f800000u), __uint_as_float(0x7f800000u));
        mx[i] = make_float4(__uint_as_float(0xff800000u), __uint_as_float(0xff800000u), __uint_as_float(0xff800000u), __uint_as_float(0xff800000u));
    }
    // Rows are taken RB at a time: the RB sums of squares are reduced across the wave together, then lanes 0 .. RB - 1 do the
    // per-row work ONCE for their row (two double square roots, the reciprocal of the norm: ~60 instructions that every lane used
    // to repeat for every row), and the RB (norm, reciprocal) pairs come back as scalars (readlane).
    constexpr int RB = 4;
    float4 cur[RB][NF], nxt[RB][NF];
    auto fetch = [&](int64_t row, float4 (&o)[NF]) {
        const int64_t r = row < n ? row : n - 1;  // clamped: the tail re-reads the last row, which changes no extreme
#pragma unroll
        for (int i = 0; i < NF; ++i) o[i] = SQ8_LD(&x4[r * CG + lane + 64 * i]);
    };
#pragma unroll
    for (int p = 0; p < RB; ++p) fetch(w0 + p * nw, nxt[p]);
    for (int64_t row = w0; row < n; row += RB * nw) {
#pragma unroll
        for (int p = 0; p < RB; ++p) {
#pragma unroll
            for (int i = 0; i < NF; ++i) cur[p][i] = nxt[p][i];
            fetch(row + (p + RB) * nw, nxt[p]);
        }
        double s[RB];
#pragma unroll
        for (int p = 0; p < RB; ++p) {
            s[p] = 0.0;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                s[p] += (double)__fmul_rn(cur[p][i].x, cur[p][i].x); s[p] += (double)__fmul_rn(cur[p][i].y, cur[p][i].y);
                s[p] += (double)__fmul_rn(cur[p][i].z, cur[p][i].z); s[p] += (double)__fmul_rn(cur[p][i].w, cur[p][i].w);
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) s[p] += __shfl_xor(s[p], o, 64);
        }
        // lane p < RB finishes row p
        double mine = s[0];
#pragma unroll
        for (int p = 1; p < RB; ++p) mine = lane == p ? s[p] : mine;
        const double rlo = __dsqrt_rn(mine * (1.0 - 0x1p-42)), rhi = __dsqrt_rn(mine * (1.0 + 0x1p-42));
        float den = (float)(rlo > 1e-12 ? rlo : 1e-12);
        const float fhi = (float)(rhi > 1e-12 ? rhi : 1e-12);
        const unsigned long long unproven = __ballot(lane < RB && !(den == fhi));
        if (unproven) {  // rare: not proven, or not finite -- the reference's own order for those rows (int8_quan.cc:48-51)
#pragma unroll
            for (int p = 0; p < RB; ++p) {
                if ((unproven >> p) & 1ull) {  // wave-uniform
                    const int64_t r = row + p * nw < n ? row + p * nw : n - 1;
                    double accum = 0.0;
                    for (int e = 0; e < D; ++e) {
                        const float t = x[r * D + e];
                        accum += (double)__fmul_rn(t, t);
                    }
                    const double nrm = __dsqrt_rn(accum);
                    if (lane == p) den = (float)(nrm > 1e-12 ? nrm : 1e-12);
                }
            }
        }
        const DivBy dl = div_by(den);
#pragma unroll
        for (int p = 0; p < RB; ++p) {
            DivBy dd;
            dd.b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dl.b), p));
            dd.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dl.y), p));
            dd.ok = __builtin_amdgcn_readlane((int)dl.ok, p) != 0;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const float a0 = div_rn(cur[p][i].x, dd), a1 = div_rn(cur[p][i].y, dd), a2 = div_rn(cur[p][i].z, dd), a3 = div_rn(cur[p][i].w, dd);
                mn[i].x = a0 < mn[i].x ? a0 : mn[i].x; mn[i].y = a1 < mn[i].y ? a1 : mn[i].y;
                mn[i].z = a2 < mn[i].z ? a2 : mn[i].z; mn[i].w = a3 < mn[i].w ? a3 : mn[i].w;
                mx[i].x = a0 > mx[i].x ? a0 : mx[i].x; mx[i].y = a1 > mx[i].y ? a1 : mx[i].y;
                mx[i].z = a2 > mx[i].z ? a2 : mx[i].z; mx[i].w = a3 > mx[i].w ? a3 : mx[i].w;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const int c = 4 * (lane + 64 * i);
        const float lo[4] = { mn[i].x, mn[i].y, mn[i].z, mn[i].w }, hi[4] = { mx[i].x, mx[i].y, mx[i].z, mx[i].w };
#pragma unroll
        for (int j = 0; j < 4; ++j) { atomicMin(&smin[c + j], f32_key(lo[j])); atomicMax(&smax[c + j], f32_key(hi[j])); }
    }
    __syncthreads();
    for (int c = tid; c < D; c += kBlock) { atomicMin(&kmin[c], smin[c]); atomicMax(&kmax[c], smax[c]); }
}

__global__ void sq8_train_finish_kernel(const uint32_t *kmin, const uint32_t *kmax, int d, float *vmin, float *vdiff)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < d) {
        const float lo = key_f32(kmin[i]), hi = key_f32(kmax[i]);
        vmin[i] = lo;
        vdiff[i] = __fsub_rn(hi, lo);
    }
}

// Encode with whole rows per wave (d = 256 / 512), the layout of sq8_train_wave_kernel: a lane owns 4 NF columns (their vmin and
// the reciprocal of their vdiff stay in registers), RB rows are in flight, the norm of a row -- when asked for -- is reduced across
// the wave and finished once by one lane, the normalised row goes back to x (the reference normalises its caller's buffer) and the
// codes leave as NF coalesced dword stores per lane.  No LDS tile, no barrier: the tile kernel's phases wait for each other behind
// three barriers with one workgroup per CU.
template <int NF, bool NORM>
__global__ __launch_bounds__(kBlock) void sq8_encode_wave_kernel(const float *__restrict__ vmin, const float *__restrict__ vdiff, float *x, int64_t n,
                                                                int write_back, uint8_t *__restrict__ codes)
{
    constexpr int CG = 64 * NF, D = 4 * CG;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 *x4 = reinterpret_cast<float4 *>(x);
    uint32_t *c4 = reinterpret_cast<uint32_t *>(codes);
    const int64_t nw = (int64_t)gridDim.x * (kBlock / 64), w0 = (int64_t)blockIdx.x * (kBlock / 64) + wave;
    float4 lo[NF];
    DivBy df[NF][4];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        lo[i] = reinterpret_cast<const float4 *>(vmin)[lane + 64 * i];
        const float4 d4 = reinterpret_cast<const float4 *>(vdiff)[lane + 64 * i];
        df[i][0] = div_by(d4.x); df[i][1] = div_by(d4.y); df[i][2] = div_by(d4.z); df[i][3] = div_by(d4.w);
    }
    constexpr int RB = 4;
    float4 cur[RB][NF], nxt[RB][NF];
    auto fetch = [&](int64_t row, float4 (&o)[NF]) {
        const int64_t r = row < n ? row : n - 1;  // clamped: rows past the end are computed, never stored
#pragma unroll
        for (int i = 0; i < NF; ++i) o[i] = SQ8_LD(&x4[r * CG + lane + 64 * i]);
    };
#pragma unroll
    for (int p = 0; p < RB; ++p) fetch(w0 + p * nw, nxt[p]);
    for (int64_t row = w0; row < n; row += RB * nw) {
#pragma unroll
        for (int p = 0; p < RB; ++p) {
#pragma unroll
            for (int i = 0; i < NF; ++i) cur[p][i] = nxt[p][i];
            fetch(row + (p + RB) * nw, nxt[p]);
        }
        DivBy dl;
        dl.b = 1.0f; dl.y = 1.0f; dl.ok = true;
        if constexpr (NORM) {
            double s[RB];
#pragma unroll
            for (int p = 0; p < RB; ++p) {
                s[p] = 0.0;
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    s[p] += (double)__fmul_rn(cur[p][i].x, cur[p][i].x); s[p] += (double)__fmul_rn(cur[p][i].y, cur[p][i].y);
                    s[p] += (double)__fmul_rn(cur[p][i].z, cur[p][i].z); s[p] += (double)__fmul_rn(cur[p][i].w, cur[p][i].w);
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) s[p] += __shfl_xor(s[p], o, 64);
            }
            double mine = s[0];
#pragma unroll
            for (int p = 1; p < RB; ++p) mine = lane == p ? s[p] : mine;
            // same proof as the tile kernel: the float root is order-independent when the roots of sum (1 -+ 2^-42) coincide
            const double rlo = __dsqrt_rn(mine * (1.0 - 0x1p-42)), rhi = __dsqrt_rn(mine * (1.0 + 0x1p-42));
            float den = (float)(rlo > 1e-12 ? rlo : 1e-12);
            const float fhi = (float)(rhi > 1e-12 ? rhi : 1e-12);
            const unsigned long long unproven = __ballot(lane < RB && !(den == fhi));
            if (unproven) {  // rare: the reference's own order for those rows (int8_quan.cc:48-51)
#pragma unroll
                for (int p = 0; p < RB; ++p) {
                    if ((unproven >> p) & 1ull) {  // wave-uniform
                        const int64_t r = row + p * nw < n ? row + p * nw : n - 1;
                        double accum = 0.0;
                        for (int e = 0; e < D; ++e) {
                            const float t = x[r * D + e];
                            accum += (double)__fmul_rn(t, t);
                        }
                        const double nrm = __dsqrt_rn(accum);
                        if (lane == p) den = (float)(nrm > 1e-12 ? nrm : 1e-12);
                    }
                }
            }
            dl = div_by(den);
        }
#pragma unroll
        for (int p = 0; p < RB; ++p) {
            const int64_t r = row + p * nw;
            DivBy dd = dl;
            if constexpr (NORM) {
                dd.b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dl.b), p));
                dd.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dl.y), p));
                dd.ok = __builtin_amdgcn_readlane((int)dl.ok, p) != 0;
            }
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                float4 v = cur[p][i];
                if constexpr (NORM) {
                    v.x = div_rn(v.x, dd); v.y = div_rn(v.y, dd); v.z = div_rn(v.z, dd); v.w = div_rn(v.w, dd);
                    if (write_back && r < n) SQ8_ST(&x4[r * CG + lane + 64 * i], v);
                }
                const uint32_t w = sq8_byte(v.x, lo[i].x, df[i][0]) | (sq8_byte(v.y, lo[i].y, df[i][1]) << 8) |
                                   (sq8_byte(v.z, lo[i].z, df[i][2]) << 16) | (sq8_byte(v.w, lo[i].w, df[i][3]) << 24);
                if (r < n) SQ8_ST(&c4[r * CG + lane + 64 * i], w);
            }
        }
    }
}

// ---- round 5: the same two kernels with the per-element work turned into a DECISION FILTER ----------------------------
// The wave kernels above are bound by their vector work, not by HBM (normalising encode 0.36-0.45 of the peak, training 0.63):
// two correctly rounded divisions per element (8 instructions each with their guards) and the byte conversion.  But what leaves
// per element is one BYTE (encode) or nothing at all (training: a column extreme changes a handful of times per wave), so the
// exact chain is only needed where a cheap approximation cannot decide.
//
// Encode.  Reference chain (int8_quan.cc:46-56, :79-92), u = 2^-24:  a = RN(v / den), b = RN(a - lo), x = RN(b / df),
// x' = clamp(x, 0, 1), y = RN(255 x'), code = (int) y.  Real value t* = 255 (v / den - lo) / df, R = |lo| / |df|.
//   the chain:      |y - t*| <= 255 u (R + |t*| / 255) [rounding of a] + 3.01 u |t*| [b, x, y]   <= u (255 R + 1030)   for |t*| <= 256
//   the filter:     T = fma(p, s, c),  p = RN(v RN(1 / den)) (or the exact a when the row is written back),  s = RN(255 RN(1 / df)),
//                   c = RN(-lo s):  |T - t*| <= 255 (R + |t*| / 255) 5.01 u + 255 R 4.01 u               <= u (2300 R + 1290)
//   so with E = 2^-24 (2600 R + 2400) (+ 2^-60 for products that underflow): a T that is farther than E from every integer has
//   floor(T) = floor(y) whenever 0 <= T < 255, T <= -E means x < 0 (code 0: the sign of b / df is exact), T >= 255 + E means y = 255.
//   => code = clamp(floor(T), 0, 255) when |frac(T) - 1/2| < 1/2 - E; every other element -- about 2 E of them, NaN / inf / huge
//   values (their frac test fails by itself), columns with R > 16 or |df|, |lo| outside 2^-40 .. 2^40 (h = -1: never sure), rows
//   whose norm is outside div_by's guarded range -- takes the chain itself.  Exact zeros (half of a ReLU'd feature matrix) would
//   sit ON an integer when lo = 0: a = +-0 exactly, and their code is a per-column constant computed by the chain once.
// Training.  Only min / max of a = RN(v / den) per column leave.  With p' = RN(v RN(1 / den) 2^60):  p' / 2^60 = a (1 +- 3.01 u), and
//   p' = +-0 only if a = +-0 (the scaling keeps tiny quotients away from the underflow threshold).  An element whose p' is not below
//   tmn' = (mn + 8 u |mn|) 2^60 has a >= mn and cannot change the minimum (same for the maximum); the others -- new extremes and the
//   few inside the 8 u band -- are divided exactly.  Every wave starts from the extremes of the first rows (a sample pass over
//   8192 rows through the same kernel), so "new extreme" is rare from its first row on.
// cvtmi_set_tuning("sq8_filter", 0) restores the kernels above (tests run both; results are identical bit for bit).
// Sum of a double over the 64 lanes of a wave, same value in every lane, in whatever order (the proof above only needs the sum to
// within 2^-43).  The xor butterfly costs 12 ds_bpermute round trips per row (~100 cycles each, dependent); the kernels of round 5 no
// longer have the vector work to hide them behind, so the sum runs on DPP moves inside each row of 16 lanes (4 steps, a few cycles
// each) and the four row sums are combined through scalar registers.
__device__ __forceinline__ double sq8_dpp_step(double s, const int ctrl_sel)
{
    const int lo = __double2loint(s), hi = __double2hiint(s);
    int lo2, hi2;
    if (ctrl_sel == 0) { lo2 = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xf, 0xf, false); hi2 = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xf, 0xf, false); }        // quad_perm [1,0,3,2]
    else if (ctrl_sel == 1) { lo2 = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xf, 0xf, false); hi2 = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xf, 0xf, false); }   // quad_perm [2,3,0,1]
    else if (ctrl_sel == 2) { lo2 = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xf, 0xf, false); hi2 = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xf, 0xf, false); } // row_half_mirror
    else { lo2 = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xf, 0xf, false); hi2 = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xf, 0xf, false); }                    // row_mirror
    return s + __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double sq8_wave_sum(double s, bool dpp)
{
    if (dpp) {
        s = sq8_dpp_step(s, 0); s = sq8_dpp_step(s, 1); s = sq8_dpp_step(s, 2); s = sq8_dpp_step(s, 3);   // every lane: the sum of its row of 16
        const int lo = __double2loint(s), hi = __double2hiint(s);
        const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
        const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
        const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
        const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
        return (r0 + r1) + (r2 + r3);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    return s;
}

struct ColF { float s, c, h; uint32_t code0; };
__device__ __forceinline__ ColF sq8_col_filter(float lo, float df)
{
    ColF f;
    const float adf = fabsf(df), alo = fabsf(lo);
    const float rdf = __fdiv_rn(1.0f, df);
    f.s = __fmul_rn(255.0f, rdf);
    f.c = __fmul_rn(-lo, f.s);
    const float R = __fmul_rn(alo, fabsf(rdf));
    const bool usable = adf >= 0x1p-40f && adf <= 0x1p40f && alo <= 0x1p40f && R <= 16.0f;   // false for NaN anywhere
    const float E = __fmaf_rn(__fmaf_rn(2600.0f, R, 2400.0f), 0x1.01p-24f, 0x1p-60f);
    f.h = usable ? __fsub_rn(0.5f, E) : -1.0f;
    f.code0 = sq8_byte(0.0f, lo, div_by(df));
    return f;
}

// NORM = false: rows are encoded as they are (turn_off_l2norm): a = v exactly, the same filter decides the byte
template <int NF, bool NORM>
__global__ __launch_bounds__(kBlock) void sq8_encode_wave_f_kernel(const float *__restrict__ vmin, const float *__restrict__ vdiff, float *x, int64_t n,
                                                                  int write_back, uint8_t *__restrict__ codes, int flags)
{
    const bool dpp = (flags & 1) != 0;
    constexpr int CG = 64 * NF, D = 4 * CG;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 *x4 = reinterpret_cast<float4 *>(x);
    uint32_t *c4 = reinterpret_cast<uint32_t *>(codes);
    const int64_t nw = (int64_t)gridDim.x * (kBlock / 64), w0 = (int64_t)blockIdx.x * (kBlock / 64) + wave;
    ColF cf[NF][4];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const float4 l4 = reinterpret_cast<const float4 *>(vmin)[lane + 64 * i];
        const float4 d4 = reinterpret_cast<const float4 *>(vdiff)[lane + 64 * i];
        cf[i][0] = sq8_col_filter(l4.x, d4.x); cf[i][1] = sq8_col_filter(l4.y, d4.y);
        cf[i][2] = sq8_col_filter(l4.z, d4.z); cf[i][3] = sq8_col_filter(l4.w, d4.w);
    }
    constexpr int RB = NF <= 2 ? 4 : NF <= 4 ? 2 : 1;   // rows in flight per wave: 8 KB either way
    float4 cur[RB][NF], nxt[RB][NF];
    auto fetch = [&](int64_t row, float4 (&o)[NF]) {
        const int64_t r = row < n ? row : n - 1;  // clamped: rows past the end are computed, never stored
#pragma unroll
        for (int i = 0; i < NF; ++i) o[i] = SQ8_LD(&x4[r * CG + lane + 64 * i]);
    };
#pragma unroll
    for (int p = 0; p < RB; ++p) fetch(w0 + p * nw, nxt[p]);
    for (int64_t row = w0; row < n; row += RB * nw) {
#pragma unroll
        for (int p = 0; p < RB; ++p) {
#pragma unroll
            for (int i = 0; i < NF; ++i) cur[p][i] = nxt[p][i];
            fetch(row + (p + RB) * nw, nxt[p]);
        }
        float den = 1.0f;
        if constexpr (NORM) {
        double s[RB];
#pragma unroll
        for (int p = 0; p < RB; ++p) {
            s[p] = 0.0;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                s[p] += (double)__fmul_rn(cur[p][i].x, cur[p][i].x); s[p] += (double)__fmul_rn(cur[p][i].y, cur[p][i].y);
                s[p] += (double)__fmul_rn(cur[p][i].z, cur[p][i].z); s[p] += (double)__fmul_rn(cur[p][i].w, cur[p][i].w);
            }
            s[p] = sq8_wave_sum(s[p], dpp);
        }
        double mine = s[0];
#pragma unroll
        for (int p = 1; p < RB; ++p) mine = lane == p ? s[p] : mine;
        // same proof as the tile kernel: the float root is order-independent when the roots of sum (1 -+ 2^-42) coincide
        const double rlo = __dsqrt_rn(mine * (1.0 - 0x1p-42)), rhi = __dsqrt_rn(mine * (1.0 + 0x1p-42));
        den = (float)(rlo > 1e-12 ? rlo : 1e-12);
        const float fhi = (float)(rhi > 1e-12 ? rhi : 1e-12);
        const unsigned long long unproven = __ballot(lane < RB && !(den == fhi));
        if (unproven) {  // rare: the reference's own order for those rows (int8_quan.cc:48-51)
#pragma unroll
            for (int p = 0; p < RB; ++p) {
                if ((unproven >> p) & 1ull) {  // wave-uniform
                    const int64_t r = row + p * nw < n ? row + p * nw : n - 1;
                    double accum = 0.0;
                    for (int e = 0; e < D; ++e) {
                        const float t = x[r * D + e];
                        accum += (double)__fmul_rn(t, t);
                    }
                    const double nrm = __dsqrt_rn(accum);
                    if (lane == p) den = (float)(nrm > 1e-12 ? nrm : 1e-12);
                }
            }
        }
        }
        const DivBy dl = div_by(den);
#pragma unroll
        for (int p = 0; p < RB; ++p) {
            const int64_t r = row + p * nw;
            DivBy dd;
            dd.b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dl.b), p));
            dd.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dl.y), p));
            dd.ok = __builtin_amdgcn_readlane((int)dl.ok, p) != 0;
            if (NORM && !dd.ok) {   // wave-uniform, rare: a norm outside the guarded range (zero / huge / non-finite rows) -- the chain for the whole row
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    float4 v = cur[p][i];
                    v.x = div_rn(v.x, dd); v.y = div_rn(v.y, dd); v.z = div_rn(v.z, dd); v.w = div_rn(v.w, dd);
                    if (write_back && r < n) SQ8_ST(&x4[r * CG + lane + 64 * i], v);
                    const float4 l4 = reinterpret_cast<const float4 *>(vmin)[lane + 64 * i];
                    const float4 d4 = reinterpret_cast<const float4 *>(vdiff)[lane + 64 * i];
                    const uint32_t w = sq8_byte(v.x, l4.x, div_by(d4.x)) | (sq8_byte(v.y, l4.y, div_by(d4.y)) << 8) |
                                       (sq8_byte(v.z, l4.z, div_by(d4.z)) << 16) | (sq8_byte(v.w, l4.w, div_by(d4.w)) << 24);
                    if (r < n) SQ8_ST(&c4[r * CG + lane + 64 * i], w);
                }
                continue;
            }
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const float4 v = cur[p][i];
                float e[4] = { v.x, v.y, v.z, v.w };
                if (NORM && write_back) {   // (uniform) the reference's in-place normalisation: the exact quotients are needed anyway
#pragma unroll
                    for (int j = 0; j < 4; ++j) e[j] = div_rn(e[j], dd);
                    if (r < n) SQ8_ST(&x4[r * CG + lane + 64 * i], make_float4(e[0], e[1], e[2], e[3]));
                }
                uint32_t w = 0u;
                bool open[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const ColF &f = cf[i][j];
                    const float pq = (!NORM || write_back) ? e[j] : __fmul_rn(e[j], dd.y);
                    const float T = __fmaf_rn(pq, f.s, f.c);
                    const float fl = floorf(T);
                    const bool sure = fabsf(__fsub_rn(__fsub_rn(T, fl), 0.5f)) < f.h;
                    const bool zero = e[j] == 0.0f;   // +-0 in, +-0 out of the division: the column's constant
                    uint32_t b = (uint32_t)__builtin_amdgcn_fmed3f(fl, 0.0f, 255.0f);
                    b = zero ? f.code0 : b;
                    open[j] = !(sure || zero);
                    w |= b << (8 * j);
                }
                if (open[0] || open[1] || open[2] || open[3]) {   // ONE branch per four elements: ~2 E of the elements, and whatever the
#pragma unroll                                                     // bound does not cover -- the chain itself
                    for (int j = 0; j < 4; ++j) {
                        if (open[j]) {
                            const int col = 4 * (lane + 64 * i) + j;
                            const float a = (!NORM || write_back) ? e[j] : div_rn(e[j], dd);
                            w = (w & ~(0xffu << (8 * j))) | (sq8_byte(a, vmin[col], div_by(vdiff[col])) << (8 * j));
                        }
                    }
                }
                if (r < n) SQ8_ST(&c4[r * CG + lane + 64 * i], w);
            }
        }
    }
}

__device__ __forceinline__ float sq8_thr_lo(float mn)   // p' >= this  =>  a >= mn   (p' = a 2^60 (1 +- 3.01 u))
{
    const float am = fabsf(mn);
    if (am == 0.0f) return 0.0f;
    if (!(am >= 0x1p-120f)) return __uint_as_float(0x7f800000u);   // a denormal extreme: every element is divided exactly
    if (!(am < 0x1p66f)) return __uint_as_float(0x7f800000u);      // the 2^60 scaling would overflow (un-normalised rows only): likewise
    return __fmul_rn(__fmaf_rn(am, 0x1p-21f, mn), 0x1p60f);
}
__device__ __forceinline__ float sq8_thr_hi(float mx)   // p' <= this  =>  a <= mx
{
    const float am = fabsf(mx);
    if (am == 0.0f) return 0.0f;
    if (!(am >= 0x1p-120f)) return __uint_as_float(0xff800000u);
    if (!(am < 0x1p66f)) return __uint_as_float(0xff800000u);      // (an element that large makes p' infinite: above any finite threshold, open)
    return __fmul_rn(__fmaf_rn(am, -0x1p-21f, mx), 0x1p60f);
}

// seeded != 0: every lane starts from the column extremes already in kmin / kmax (the sample pass)
// NORM = false (turn_off_l2norm): a = v exactly -- the thresholds decide on p' = v 2^60 alike, a candidate IS its own quotient
template <int NF, bool NORM>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(NF <= 4 ? 3 : 2, NF <= 4 ? 3 : 2))) void sq8_train_wave_f_kernel(
    const float *__restrict__ x, int64_t n, uint32_t *kmin, uint32_t *kmax, int seeded, int flags)
{
    const bool dpp = (flags & 1) != 0;
    constexpr int CG = 64 * NF, D = 4 * CG;
    __shared__ uint32_t smin[D], smax[D];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < D; c += kBlock) { smin[c] = 0xffffffffu; smax[c] = 0u; }
    __syncthreads();
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    const int64_t nw = (int64_t)gridDim.x * (kBlock / 64), w0 = (int64_t)blockIdx.x * (kBlock / 64) + wave;
    float mn[NF][4], mx[NF][4], tlo[NF][4], thi[NF][4];
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 4 * (lane + 64 * i) + j;
            mn[i][j] = seeded ? key_f32(kmin[c]) : __uint_as_float(0x7f800000u);
            mx[i][j] = seeded ? key_f32(kmax[c]) : __uint_as_float(0xff800000u);
            if (!(mn[i][j] <= mx[i][j])) { mn[i][j] = __uint_as_float(0x7f800000u); mx[i][j] = __uint_as_float(0xff800000u); }   // an empty / all-NaN sample
            tlo[i][j] = sq8_thr_lo(mn[i][j]);
            thi[i][j] = sq8_thr_hi(mx[i][j]);
        }
    constexpr int RB = NF <= 2 ? 4 : NF <= 4 ? 2 : 1;
    float4 cur[RB][NF], nxt[RB][NF];
    auto fetch = [&](int64_t row, float4 (&o)[NF]) {
        const int64_t r = row < n ? row : n - 1;  // clamped: the tail re-reads the last row, which changes no extreme
#pragma unroll
        for (int i = 0; i < NF; ++i) o[i] = SQ8_LD(&x4[r * CG + lane + 64 * i]);
    };
#pragma unroll
    for (int p = 0; p < RB; ++p) fetch(w0 + p * nw, nxt[p]);
    for (int64_t row = w0; row < n; row += RB * nw) {
#pragma unroll
        for (int p = 0; p < RB; ++p) {
#pragma unroll
            for (int i = 0; i < NF; ++i) cur[p][i] = nxt[p][i];
            fetch(row + (p + RB) * nw, nxt[p]);
        }
        float den = 1.0f;
        if constexpr (NORM) {
        double s[RB];
#pragma unroll
        for (int p = 0; p < RB; ++p) {
            s[p] = 0.0;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                s[p] += (double)__fmul_rn(cur[p][i].x, cur[p][i].x); s[p] += (double)__fmul_rn(cur[p][i].y, cur[p][i].y);
                s[p] += (double)__fmul_rn(cur[p][i].z, cur[p][i].z); s[p] += (double)__fmul_rn(cur[p][i].w, cur[p][i].w);
            }
            s[p] = sq8_wave_sum(s[p], dpp);
        }
        double mine = s[0];
#pragma unroll
        for (int p = 1; p < RB; ++p) mine = lane == p ? s[p] : mine;
        const double rlo = __dsqrt_rn(mine * (1.0 - 0x1p-42)), rhi = __dsqrt_rn(mine * (1.0 + 0x1p-42));
        den = (float)(rlo > 1e-12 ? rlo : 1e-12);
        const float fhi = (float)(rhi > 1e-12 ? rhi : 1e-12);
        const unsigned long long unproven = __ballot(lane < RB && !(den == fhi));
        if (unproven) {  // rare: not proven, or not finite -- the reference's own order for those rows (int8_quan.cc:48-51)
#pragma unroll
            for (int p = 0; p < RB; ++p) {
                if ((unproven >> p) & 1ull) {  // wave-uniform
                    const int64_t r = row + p * nw < n ? row + p * nw : n - 1;
                    double accum = 0.0;
                    for (int e = 0; e < D; ++e) {
                        const float t = x[r * D + e];
                        accum += (double)__fmul_rn(t, t);
                    }
                    const double nrm = __dsqrt_rn(accum);
                    if (lane == p) den = (float)(nrm > 1e-12 ? nrm : 1e-12);
                }
            }
        }
        }
        const DivBy dl = div_by(den);
#pragma unroll
        for (int p = 0; p < RB; ++p) {
            DivBy dd;
            dd.b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dl.b), p));
            dd.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dl.y), p));
            dd.ok = __builtin_amdgcn_readlane((int)dl.ok, p) != 0;
            const float ys = __fmul_rn(dd.y, 0x1p60f);   // exact scaling (dd.y <= 2^40 when ok)
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const float e[4] = { cur[p][i].x, cur[p][i].y, cur[p][i].z, cur[p][i].w };
                bool open[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float pp = __fmul_rn(e[j], ys);
                    open[j] = !dd.ok || pp < tlo[i][j] || pp > thi[i][j];   // a candidate for an extreme (or a row the bound does not cover)
                }
                if (open[0] || open[1] || open[2] || open[3]) {   // ONE branch per four elements; the candidates are divided exactly
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (open[j]) {
                            const float a = NORM ? div_rn(e[j], dd) : e[j];
                            if (a < mn[i][j]) { mn[i][j] = a; tlo[i][j] = sq8_thr_lo(a); }
                            if (a > mx[i][j]) { mx[i][j] = a; thi[i][j] = sq8_thr_hi(a); }
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const int c = 4 * (lane + 64 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j) { atomicMin(&smin[c + j], f32_key(mn[i][j])); atomicMax(&smax[c + j], f32_key(mx[i][j])); }
    }
    __syncthreads();
    for (int c = tid; c < D; c += kBlock) { atomicMin(&kmin[c], smin[c]); atomicMax(&kmax[c], smax[c]); }
}

// ---- rows narrower than a wave's 256 floats (round 6): 64-d and 128-d -- the reference's own demo vector is 64-d
// (scalar_quantization/int8_quan_test.cpp:26) and these rows stayed on the tile kernel at 1.0-3.5 TB/s.  Rows are contiguous, so 64 / LPR
// consecutive rows ARE one 1 KB "wave row" (LPR = lanes per data row: 16 at 64-d, 32 at 128-d): the loads, the code stores and the
// write-back keep the wide kernels' coalesced form and only three things change -- the sum of squares stops at the row's own LPR lanes
// (the DPP steps inside a row of 16 lanes, one more exchange at 128-d), the per-row work (two double roots, the reciprocal) is done by
// lane p of every 16-lane row for register-row p, all data rows of the iteration at once, and handed to the row's lanes by one
// v_mov_b32 row_newbcast each; and a lane's columns are those of lane % LPR.  Same decision filters, same chains behind them: same bits.
template <int LPR>
__device__ __forceinline__ double sq8_group_sum(double s)
{
    s = sq8_dpp_step(s, 0); s = sq8_dpp_step(s, 1); s = sq8_dpp_step(s, 2); s = sq8_dpp_step(s, 3);   // every lane: the sum of its row of 16
    if constexpr (LPR == 32) s += __shfl_xor(s, 16, 64);
    return s;
}
template <int P> __device__ __forceinline__ int sq8_row_bcast(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x150 + P, 0xf, 0xf, false); }   // row_newbcast:P
__device__ __forceinline__ DivBy sq8_bcast_divby(const DivBy &dl, int p)   // lane p of every 16-lane row -> that row's lanes (p a constant after unrolling)
{
    DivBy dd;
    const int b = __float_as_int(dl.b), y = __float_as_int(dl.y), ok = (int)dl.ok;
    int rb, ry, rk;
    switch (p) {
        case 0: rb = sq8_row_bcast<0>(b); ry = sq8_row_bcast<0>(y); rk = sq8_row_bcast<0>(ok); break;
        case 1: rb = sq8_row_bcast<1>(b); ry = sq8_row_bcast<1>(y); rk = sq8_row_bcast<1>(ok); break;
        case 2: rb = sq8_row_bcast<2>(b); ry = sq8_row_bcast<2>(y); rk = sq8_row_bcast<2>(ok); break;
        default: rb = sq8_row_bcast<3>(b); ry = sq8_row_bcast<3>(y); rk = sq8_row_bcast<3>(ok); break;
    }
    dd.b = __int_as_float(rb); dd.y = __int_as_float(ry); dd.ok = rk != 0;
    return dd;
}
// the per-row denominators of the RB register-rows of one iteration: leader lanes ((lane & 15) == p) return theirs, everybody else 1.0
// g0: group of register-row 0, gstep: groups between register-rows
template <int LPR, int RB>
__device__ __forceinline__ float sq8_group_den(const float *__restrict__ x, int64_t n, const float4 (&cur)[RB], int64_t g0, int64_t gstep)
{
    constexpr int SUB = 64 / LPR, D = 4 * LPR;
    const int lane = threadIdx.x & 63, l16 = lane & 15;
    double s[RB];
#pragma unroll
    for (int p = 0; p < RB; ++p) {
        s[p] = 0.0;
        s[p] += (double)__fmul_rn(cur[p].x, cur[p].x); s[p] += (double)__fmul_rn(cur[p].y, cur[p].y);
        s[p] += (double)__fmul_rn(cur[p].z, cur[p].z); s[p] += (double)__fmul_rn(cur[p].w, cur[p].w);
        s[p] = sq8_group_sum<LPR>(s[p]);
    }
    double mine = s[0];
#pragma unroll
    for (int p = 1; p < RB; ++p) mine = l16 == p ? s[p] : mine;
    // same proof as the tile kernel: the float root is order-independent when the roots of sum (1 -+ 2^-42) coincide
    const double rlo = __dsqrt_rn(mine * (1.0 - 0x1p-42)), rhi = __dsqrt_rn(mine * (1.0 + 0x1p-42));
    float den = (float)(rlo > 1e-12 ? rlo : 1e-12);
    const float fhi = (float)(rhi > 1e-12 ? rhi : 1e-12);
    unsigned long long unproven = __ballot(l16 < RB && !(den == fhi));
    while (unproven) {  // rare: the reference's own order for those rows (int8_quan.cc:48-51); wave-uniform loop
        const int lb = __ffsll((long long)unproven) - 1;
        unproven &= unproven - 1;
        int64_t r = (g0 + (int64_t)(lb & 15) * gstep) * SUB + lb / LPR;
        r = r < n ? r : n - 1;
        double accum = 0.0;
        for (int e = 0; e < D; ++e) {
            const float t = x[r * D + e];
            accum += (double)__fmul_rn(t, t);
        }
        const double nrm = __dsqrt_rn(accum);
        if (lane == lb) den = (float)(nrm > 1e-12 ? nrm : 1e-12);
    }
    return den;
}

template <int LPR, bool NORM>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(3, 3))) void sq8_train_group_f_kernel(
    const float *__restrict__ x, int64_t n, uint32_t *kmin, uint32_t *kmax, int seeded)
{
    constexpr int SUB = 64 / LPR, D = 4 * LPR, RB = 4;
    __shared__ uint32_t smin[D], smax[D];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lc = lane % LPR, lr = lane / LPR;
    for (int c = tid; c < D; c += kBlock) { smin[c] = 0xffffffffu; smax[c] = 0u; }
    __syncthreads();
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    const int64_t ng = (n + SUB - 1) / SUB;
    const int64_t nw = (int64_t)gridDim.x * (kBlock / 64), w0 = (int64_t)blockIdx.x * (kBlock / 64) + wave;
    float mn[4], mx[4], tlo[4], thi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = 4 * lc + j;
        mn[j] = seeded ? key_f32(kmin[c]) : __uint_as_float(0x7f800000u);
        mx[j] = seeded ? key_f32(kmax[c]) : __uint_as_float(0xff800000u);
        if (!(mn[j] <= mx[j])) { mn[j] = __uint_as_float(0x7f800000u); mx[j] = __uint_as_float(0xff800000u); }   // an empty / all-NaN sample
        tlo[j] = sq8_thr_lo(mn[j]);
        thi[j] = sq8_thr_hi(mx[j]);
    }
    float4 cur[RB], nxt[RB];
    auto fetch = [&](int64_t g, float4 &o) {
        int64_t r = g * SUB + lr;
        r = r < n ? r : n - 1;  // clamped per lane: the tail re-reads the last row, which changes no extreme
        o = SQ8_LD(&x4[r * LPR + lc]);
    };
#pragma unroll
    for (int p = 0; p < RB; ++p) fetch(w0 + p * nw, nxt[p]);
    for (int64_t g = w0; g < ng; g += RB * nw) {
#pragma unroll
        for (int p = 0; p < RB; ++p) {
            cur[p] = nxt[p];
            fetch(g + (p + RB) * nw, nxt[p]);
        }
        float den = 1.0f;
        if constexpr (NORM) den = sq8_group_den<LPR, RB>(x, n, cur, g, nw);
        const DivBy dl = div_by(den);
#pragma unroll
        for (int p = 0; p < RB; ++p) {
            const DivBy dd = sq8_bcast_divby(dl, p);
            const float ys = __fmul_rn(dd.y, 0x1p60f);   // exact scaling (dd.y <= 2^40 when ok)
            const float e[4] = { cur[p].x, cur[p].y, cur[p].z, cur[p].w };
            bool open[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float pp = __fmul_rn(e[j], ys);
                open[j] = !dd.ok || pp < tlo[j] || pp > thi[j];   // a candidate for an extreme (or a row the bound does not cover)
            }
            if (open[0] || open[1] || open[2] || open[3]) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (open[j]) {
                        const float a = NORM ? div_rn(e[j], dd) : e[j];
                        if (a < mn[j]) { mn[j] = a; tlo[j] = sq8_thr_lo(a); }
                        if (a > mx[j]) { mx[j] = a; thi[j] = sq8_thr_hi(a); }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { atomicMin(&smin[4 * lc + j], f32_key(mn[j])); atomicMax(&smax[4 * lc + j], f32_key(mx[j])); }
    __syncthreads();
    for (int c = tid; c < D; c += kBlock) { atomicMin(&kmin[c], smin[c]); atomicMax(&kmax[c], smax[c]); }
}

template <int LPR, bool NORM>
__global__ __launch_bounds__(kBlock) void sq8_encode_group_f_kernel(const float *__restrict__ vmin, const float *__restrict__ vdiff, float *x, int64_t n,
                                                                   int write_back, uint8_t *__restrict__ codes)
{
    constexpr int SUB = 64 / LPR, RB = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lc = lane % LPR, lr = lane / LPR;
    float4 *x4 = reinterpret_cast<float4 *>(x);
    uint32_t *c4 = reinterpret_cast<uint32_t *>(codes);
    const int64_t ng = (n + SUB - 1) / SUB;
    const int64_t nw = (int64_t)gridDim.x * (kBlock / 64), w0 = (int64_t)blockIdx.x * (kBlock / 64) + wave;
    ColF cf[4];
    {
        const float4 l4 = reinterpret_cast<const float4 *>(vmin)[lc];
        const float4 d4 = reinterpret_cast<const float4 *>(vdiff)[lc];
        cf[0] = sq8_col_filter(l4.x, d4.x); cf[1] = sq8_col_filter(l4.y, d4.y);
        cf[2] = sq8_col_filter(l4.z, d4.z); cf[3] = sq8_col_filter(l4.w, d4.w);
    }
    float4 cur[RB], nxt[RB];
    auto fetch = [&](int64_t g, float4 &o) {
        int64_t r = g * SUB + lr;
        r = r < n ? r : n - 1;  // clamped per lane: rows past the end are computed, never stored
        o = SQ8_LD(&x4[r * LPR + lc]);
    };
#pragma unroll
    for (int p = 0; p < RB; ++p) fetch(w0 + p * nw, nxt[p]);
    for (int64_t g = w0; g < ng; g += RB * nw) {
#pragma unroll
        for (int p = 0; p < RB; ++p) {
            cur[p] = nxt[p];
            fetch(g + (p + RB) * nw, nxt[p]);
        }
        float den = 1.0f;
        if constexpr (NORM) den = sq8_group_den<LPR, RB>(x, n, cur, g, nw);
        const DivBy dl = div_by(den);
#pragma unroll
        for (int p = 0; p < RB; ++p) {
            const int64_t r = (g + p * nw) * SUB + lr;
            const bool live = r < n;
            const DivBy dd = sq8_bcast_divby(dl, p);
            const float4 v = cur[p];
            float e[4] = { v.x, v.y, v.z, v.w };
            uint32_t w = 0u;
            if (NORM && !dd.ok) {   // (per data row, rare) a norm outside the guarded range (zero / huge / non-finite rows): the chain for the whole row
                float4 q = v;
                q.x = div_rn(q.x, dd); q.y = div_rn(q.y, dd); q.z = div_rn(q.z, dd); q.w = div_rn(q.w, dd);
                if (write_back && live) SQ8_ST(&x4[r * LPR + lc], q);
                const float4 l4 = reinterpret_cast<const float4 *>(vmin)[lc];
                const float4 d4 = reinterpret_cast<const float4 *>(vdiff)[lc];
                w = sq8_byte(q.x, l4.x, div_by(d4.x)) | (sq8_byte(q.y, l4.y, div_by(d4.y)) << 8) |
                    (sq8_byte(q.z, l4.z, div_by(d4.z)) << 16) | (sq8_byte(q.w, l4.w, div_by(d4.w)) << 24);
            } else {
                if (NORM && write_back) {   // (uniform) the reference's in-place normalisation: the exact quotients are needed anyway
#pragma unroll
                    for (int j = 0; j < 4; ++j) e[j] = div_rn(e[j], dd);
                    if (live) SQ8_ST(&x4[r * LPR + lc], make_float4(e[0], e[1], e[2], e[3]));
                }
                bool open[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const ColF &f = cf[j];
                    const float pq = (!NORM || write_back) ? e[j] : __fmul_rn(e[j], dd.y);
                    const float T = __fmaf_rn(pq, f.s, f.c);
                    const float fl = floorf(T);
                    const bool sure = fabsf(__fsub_rn(__fsub_rn(T, fl), 0.5f)) < f.h;
                    const bool zero = e[j] == 0.0f;   // +-0 in, +-0 out of the division: the column's constant
                    uint32_t b = (uint32_t)__builtin_amdgcn_fmed3f(fl, 0.0f, 255.0f);
                    b = zero ? f.code0 : b;
                    open[j] = !(sure || zero);
                    w |= b << (8 * j);
                }
                if (open[0] || open[1] || open[2] || open[3]) {   // ONE branch per four elements: the chain itself for what the bound does not decide
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (open[j]) {
                            const int col = 4 * lc + j;
                            const float a = (!NORM || write_back) ? e[j] : div_rn(e[j], dd);
                            w = (w & ~(0xffu << (8 * j))) | (sq8_byte(a, vmin[col], div_by(vdiff[col])) << (8 * j));
                        }
                    }
                }
            }
            if (live) SQ8_ST(&c4[r * LPR + lc], w);
        }
    }
}
static bool sq8_group_width(int d) { return d == 64 || d == 128; }

static std::atomic<int> g_sq8_flags{1};    // cvtmi_set_tuning("sq8_flags"): bit 0 = wave sums on DPP instead of the ds_bpermute butterfly
void set_sq8_flags(int v) { g_sq8_flags = v; }
static std::atomic<int> g_sq8_filter{1};   // cvtmi_set_tuning("sq8_filter"): 0 = the exact chain for every element (the round 2 - 4 kernels)
void set_sq8_filter(int v) { g_sq8_filter = v != 0 ? 1 : 0; }
static bool sq8_filter_on() { return g_sq8_filter.load() != 0; }
constexpr int64_t SQ8_SAMPLE_ROWS = 8192;   // rows of the training pass that seeds every wave's extremes

// ---- the sign of a zero minimum (round 5) ----------------------------------------------------------------------------------
// The training reduction keeps column minima as ordered integer keys, where -0.0 sorts below +0.0; the loop it restates (faiss
// train_NonUniform, RS_minmax: strict '<' in row order, sq_train.cpp:100) keeps the FIRST zero it meets when the minimum is zero.
// Equal as numbers, identical codes -- but not the same bits when a column holds zeros of both signs.  So: a column whose key says
// "-0.0" (the only case in which the two can differ: some zero there is negative) is scanned for the first row whose normalised value
// is a zero, and takes that zero's sign (= the sign of the raw value: the norm is positive).  Rows are normalised exactly as the
// reference does (index-order double sum), because a quotient can also be a zero through underflow or an infinite norm.  The kernels
// leave at once unless such a column exists -- three tiny launches per training call otherwise.
__global__ __launch_bounds__(kBlock) void sq8_zero_first_kernel(const float *__restrict__ x, int64_t n, int d, int l2norm,
                                                                const float *__restrict__ vmin, uint32_t *__restrict__ first)
{
    __shared__ int any_s;
    if (threadIdx.x == 0) any_s = 0;
    __syncthreads();
    int mine = 0;
    for (int c = threadIdx.x; c < d; c += kBlock) mine |= __float_as_uint(vmin[c]) == 0x80000000u;
    if (mine) any_s = 1;
    __syncthreads();
    if (!any_s) return;   // the common case
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) {
        const float *row = x + r * d;
        float den = 1.0f;
        if (l2norm) {   // int8_quan.cc:46-56, in its own order
            double accum = 0.0;
            for (int e = 0; e < d; ++e) {
                const float t = row[e];
                accum += (double)__fmul_rn(t, t);
            }
            const double nrm = __dsqrt_rn(accum);
            den = (float)(nrm > 1e-12 ? nrm : 1e-12);
        }
        for (int c = 0; c < d; ++c) {
            if (__float_as_uint(vmin[c]) != 0x80000000u) continue;
            const float a = l2norm ? __fdiv_rn(row[c], den) : row[c];
            if (a == 0.0f) atomicMin(&first[c], (uint32_t)r);
        }
    }
}
__global__ __launch_bounds__(kBlock) void sq8_zero_sign_kernel(const float *__restrict__ x, int d, const uint32_t *__restrict__ first,
                                                               float *__restrict__ vmin)
{
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= d || __float_as_uint(vmin[c]) != 0x80000000u || first[c] == 0xffffffffu) return;
    const float v = x[(int64_t)first[c] * d + c];
    vmin[c] = (__float_as_uint(v) >> 31) ? -0.0f : 0.0f;
}

// workgroups per CU of the wave-per-row training kernel.  A wave keeps RB rows in flight that lie (waves in the grid) rows apart: with a
// power-of-two grid those streams are a power-of-two distance apart and fall on the same HBM channels -- measured on 2 M x 512-d:
// 8 per CU 4.62-4.67 TB/s, 4: 4.38, 16: 4.89, 3: 5.04, 24: 5.09 (tools: cvtmi_set_tuning("sq8_wave_blocks"))
template <bool NORM>
static void launch_sq8_train_wave_f_n(int NF, unsigned blocks, hipStream_t st, const float *x, int64_t n, uint32_t *kmin, uint32_t *kmax, int seeded, int flags)
{
    switch (NF) {
        case 1: hipLaunchKernelGGL((sq8_train_wave_f_kernel<1, NORM>), dim3(blocks), dim3(kBlock), 0, st, x, n, kmin, kmax, seeded, flags); break;
        case 2: hipLaunchKernelGGL((sq8_train_wave_f_kernel<2, NORM>), dim3(blocks), dim3(kBlock), 0, st, x, n, kmin, kmax, seeded, flags); break;
        case 3: hipLaunchKernelGGL((sq8_train_wave_f_kernel<3, NORM>), dim3(blocks), dim3(kBlock), 0, st, x, n, kmin, kmax, seeded, flags); break;
        case 4: hipLaunchKernelGGL((sq8_train_wave_f_kernel<4, NORM>), dim3(blocks), dim3(kBlock), 0, st, x, n, kmin, kmax, seeded, flags); break;
        case 6: hipLaunchKernelGGL((sq8_train_wave_f_kernel<6, NORM>), dim3(blocks), dim3(kBlock), 0, st, x, n, kmin, kmax, seeded, flags); break;
        default: hipLaunchKernelGGL((sq8_train_wave_f_kernel<8, NORM>), dim3(blocks), dim3(kBlock), 0, st, x, n, kmin, kmax, seeded, flags); break;
    }
}
static void launch_sq8_train_wave_f(int NF, bool norm, unsigned blocks, hipStream_t st, const float *x, int64_t n, uint32_t *kmin, uint32_t *kmax, int seeded,
                                    int flags)
{
    if (norm) launch_sq8_train_wave_f_n<true>(NF, blocks, st, x, n, kmin, kmax, seeded, flags);
    else launch_sq8_train_wave_f_n<false>(NF, blocks, st, x, n, kmin, kmax, seeded, flags);
}
static int g_sq8_wave_blocks = 3;
void set_sq8_wave_blocks(int v) { g_sq8_wave_blocks = v; }
static unsigned sq8_group_blocks(int64_t n, int d)
{
    const int64_t groups = (n * d + 255) / 256, per_wg = kBlock / 64;
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>((groups + per_wg - 1) / per_wg, 256 * g_sq8_wave_blocks));
}
static void launch_sq8_train_group(int d, bool norm, unsigned blocks, hipStream_t st, const float *x, int64_t n, uint32_t *kmin, uint32_t *kmax, int seeded)
{
    if (d == 64) {
        if (norm) hipLaunchKernelGGL((sq8_train_group_f_kernel<16, true>), dim3(blocks), dim3(kBlock), 0, st, x, n, kmin, kmax, seeded);
        else hipLaunchKernelGGL((sq8_train_group_f_kernel<16, false>), dim3(blocks), dim3(kBlock), 0, st, x, n, kmin, kmax, seeded);
    } else {
        if (norm) hipLaunchKernelGGL((sq8_train_group_f_kernel<32, true>), dim3(blocks), dim3(kBlock), 0, st, x, n, kmin, kmax, seeded);
        else hipLaunchKernelGGL((sq8_train_group_f_kernel<32, false>), dim3(blocks), dim3(kBlock), 0, st, x, n, kmin, kmax, seeded);
    }
}
static int launch_sq8_encode_group(const float *vmin, const float *vdiff, int d, float *x, int64_t n, int l2norm, int write_back, uint8_t *codes, hipStream_t st)
{
    const unsigned blocks = sq8_group_blocks(n, d);
    if (d == 64) {
        if (l2norm) hipLaunchKernelGGL((sq8_encode_group_f_kernel<16, true>), dim3(blocks), dim3(kBlock), 0, st, vmin, vdiff, x, n, write_back, codes);
        else hipLaunchKernelGGL((sq8_encode_group_f_kernel<16, false>), dim3(blocks), dim3(kBlock), 0, st, vmin, vdiff, x, n, 0, codes);
    } else {
        if (l2norm) hipLaunchKernelGGL((sq8_encode_group_f_kernel<32, true>), dim3(blocks), dim3(kBlock), 0, st, vmin, vdiff, x, n, write_back, codes);
        else hipLaunchKernelGGL((sq8_encode_group_f_kernel<32, false>), dim3(blocks), dim3(kBlock), 0, st, vmin, vdiff, x, n, 0, codes);
    }
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}
// den_scratch: n floats, only used for row widths the tile kernel does not take
int launch_sq8_train(const float *x, int64_t n, int d, int l2norm, float *den_scratch, uint32_t *kmin, uint32_t *kmax,
                     float *vmin, float *vdiff, hipStream_t st)
{
    const unsigned db = (unsigned)((d + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(sq8_train_init_kernel, dim3(db), dim3(kBlock), 0, st, kmin, kmax, d);
    if (n > 0) {
        // the wave-per-row kernels: normalised rows of 256 / 512 floats as before; round 5: through the filter kernel also rows of 768 ...
        // 2048 floats, and those without normalisation (the tile kernel stops at 512-d; the generic kernels ran at 0.8 / 3.3 TB/s)
        const bool wave_f = sq8_filter_on() && sq8_wave_width(d);   // (without normalisation too: 4 M x 512-d 5.75 TB/s through the tile kernel, 6.3 here)
        if (sq8_filter_on() && sq8_group_width(d) && (((uintptr_t)x) & 15) == 0 && n >= 4096) {
            // 64-d / 128-d rows (round 6): several rows per wave, sample pass + seeded pass like the wide rows
            const int64_t ns = n >= 8 * SQ8_SAMPLE_ROWS ? SQ8_SAMPLE_ROWS : 0;
            if (ns) launch_sq8_train_group(d, l2norm != 0, sq8_group_blocks(ns / 16, d), st, x, ns, kmin, kmax, 0);
            launch_sq8_train_group(d, l2norm != 0, sq8_group_blocks(n - ns, d), st, x + ns * d, n - ns, kmin, kmax, ns ? 1 : 0);
        } else if ((wave_f || (l2norm && (d == 256 || d == 512))) && (((uintptr_t)x) & 15) == 0 && n >= (wave_f && d > 512 ? 1 : 4096)) {
            // whole rows per wave, no LDS tile (the tile kernel's phases serialise behind its barriers: 3.3 TB/s at d = 512)
            const int64_t rows_per_wg = kBlock / 64;
            const unsigned blocks = (unsigned)std::min<int64_t>((n + rows_per_wg - 1) / rows_per_wg, 256 * g_sq8_wave_blocks);
            if (wave_f) {
                // sample pass over the first rows (its extremes seed every wave of the main pass: "new extreme" is rare from the start),
                // then the rest; both through the filter kernel, the sample unseeded
                const int64_t ns = n >= 8 * SQ8_SAMPLE_ROWS ? SQ8_SAMPLE_ROWS : 0;
                const float *xr = x + ns * d;
                const unsigned sblocks = (unsigned)((ns / 16 + rows_per_wg - 1) / rows_per_wg);   // 16 rows per wave of the sample
                const int fl = g_sq8_flags.load();
                if (ns) launch_sq8_train_wave_f(d / 256, l2norm != 0, sblocks, st, x, ns, kmin, kmax, 0, fl);
                launch_sq8_train_wave_f(d / 256, l2norm != 0, blocks, st, xr, n - ns, kmin, kmax, ns ? 1 : 0, fl);
            } else if (d == 512) hipLaunchKernelGGL((sq8_train_wave_kernel<2>), dim3(blocks), dim3(kBlock), 0, st, x, n, kmin, kmax);
            else hipLaunchKernelGGL((sq8_train_wave_kernel<1>), dim3(blocks), dim3(kBlock), 0, st, x, n, kmin, kmax);
        } else if (sq8_tile_ok(d, x, nullptr, nullptr, nullptr)) {
            Sq8Args a{};
            a.x = const_cast<float *>(x); a.kmin = kmin; a.kmax = kmax; a.n = n; a.d = d; a.l2norm = l2norm;
            CVTMI_TRY(launch_sq8_tile<true>(a, st));
        } else {
            if (l2norm) {
                if (!den_scratch) return fail(CVTMI_EINVAL, "sq8_train: norm scratch missing");
                CVTMI_TRY(launch_sq8_rownorm(x, n, d, den_scratch, st));
            }
            const int64_t blocks = (n + TRAIN_ROWS - 1) / TRAIN_ROWS;
            if (blocks > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "sq8_train: n too large");
            hipLaunchKernelGGL(sq8_train_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, x, n, d,
                               l2norm ? den_scratch : nullptr, kmin, kmax);
        }
    }
    hipLaunchKernelGGL(sq8_train_finish_kernel, dim3(db), dim3(kBlock), 0, st, kmin, kmax, d, vmin, vdiff);
    if (n > 0 && n < 0xffffffffLL) {   // the sign of a zero minimum (see sq8_zero_first_kernel); kmax is free again: first[d]
        CVTMI_HIP(hipMemsetAsync(kmax, 0xff, (size_t)d * sizeof(uint32_t), st));
        const unsigned zb = (unsigned)std::min<int64_t>((n + kBlock - 1) / kBlock, 1024);
        hipLaunchKernelGGL(sq8_zero_first_kernel, dim3(zb), dim3(kBlock), 0, st, x, n, d, l2norm, vmin, kmax);
        hipLaunchKernelGGL(sq8_zero_sign_kernel, dim3(db), dim3(kBlock), 0, st, x, d, kmax, vmin);
    }
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

static int launch_sq8_encode_wave(const float *vmin, const float *vdiff, int d, float *x, int64_t n, int l2norm, int write_back,
                                  uint8_t *codes, hipStream_t st)
{
    const int64_t rows_per_wg = kBlock / 64;
    const unsigned blocks = (unsigned)std::min<int64_t>((n + rows_per_wg - 1) / rows_per_wg, 256 * g_sq8_wave_blocks);
    if (g_sq8_filter.load()) {
        const int fl = g_sq8_flags.load();
#define CVTMI_SQ8_ENC(NF_)                                                                                                                                    \
        do {                                                                                                                                                  \
            if (l2norm) hipLaunchKernelGGL((sq8_encode_wave_f_kernel<NF_, true>), dim3(blocks), dim3(kBlock), 0, st, vmin, vdiff, x, n, write_back, codes, fl); \
            else hipLaunchKernelGGL((sq8_encode_wave_f_kernel<NF_, false>), dim3(blocks), dim3(kBlock), 0, st, vmin, vdiff, x, n, 0, codes, fl);                 \
        } while (0)
        switch (d / 256) {
            case 1: CVTMI_SQ8_ENC(1); break;
            case 2: CVTMI_SQ8_ENC(2); break;
            case 3: CVTMI_SQ8_ENC(3); break;
            case 4: CVTMI_SQ8_ENC(4); break;
            case 6: CVTMI_SQ8_ENC(6); break;
            default: CVTMI_SQ8_ENC(8); break;
        }
#undef CVTMI_SQ8_ENC
        CVTMI_HIP(hipGetLastError());
        return CVTMI_OK;
    }
    if (d == 512) {
        if (l2norm) hipLaunchKernelGGL((sq8_encode_wave_kernel<2, true>), dim3(blocks), dim3(kBlock), 0, st, vmin, vdiff, x, n, write_back, codes);
        else hipLaunchKernelGGL((sq8_encode_wave_kernel<2, false>), dim3(blocks), dim3(kBlock), 0, st, vmin, vdiff, x, n, write_back, codes);
    } else {
        if (l2norm) hipLaunchKernelGGL((sq8_encode_wave_kernel<1, true>), dim3(blocks), dim3(kBlock), 0, st, vmin, vdiff, x, n, write_back, codes);
        else hipLaunchKernelGGL((sq8_encode_wave_kernel<1, false>), dim3(blocks), dim3(kBlock), 0, st, vmin, vdiff, x, n, write_back, codes);
    }
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

}  // namespace cvtmi
