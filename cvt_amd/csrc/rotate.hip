// rotate.hip -- the "OPQ rotation" applied to every database and query vector.
//
// Reference: IVFOPQ::reorder (opq/src/IVFOPQ.cpp:424-439), run over each row by
// LoadSingleFeatFile (:459-461): y[i] = x[reorder_[i]] -- a permutation, i.e. the special case
// R[i][reorder_[i]] = 1 of a d x d rotation y = R x.
//
//   launch_permute      the reference's gather, verbatim (exact for every input, NaN/Inf included).
//   launch_rotate_gemm  the general rotation as an fp32 MFMA GEMM  Y[n x D] = X[n x D] * R^T.
//                       v_mfma_f32_32x32x2_f32 accumulates one product per k in ascending k with a
//                       single rounding each (== an fmaf chain), so with a 0/1 permutation matrix
//                       it reproduces the gather bit for bit on finite inputs, and with a dense R
//                       it matches a k-ordered fmaf chain (the test oracle restates that) bit for bit.
//
// MI355X mapping of the GEMM (D = 32*NT, NT <= 4): a persistent 8-wave workgroup per CU keeps R in LDS
// (row stride D+1 floats: the 32 lanes that read one k column hit 32 different banks); each wave owns
// 32 rows x D outputs = NT accumulators of 32x32 (16 VGPRs each), holds its 32 x D slab of X in
// registers (loaded straight from HBM) and issues NT MFMAs per k-pair with B read from LDS.
#include "kernels.h"

namespace cvtmi {

// y[r][i] = x[r][perm[i]] (IVFOPQ.cpp:424-439).  Pure data movement, 2 x 4D bytes per row: a thread owns VEC
// adjacent output columns (its perm entries stay in registers), gathers inside the row -- the row's cache
// lines are shared by the threads around it -- and writes one VEC-wide store; rows are walked grid-stride.
template <int VEC>
__global__ __launch_bounds__(kBlock) void permute_kernel(const int32_t *__restrict__ perm, int D,
                                                         const float *__restrict__ x, int64_t n,
                                                         float *__restrict__ y)
{
    const int CG = D / VEC;                 // column groups per row
    const int TY = kBlock / CG;             // rows per sweep of the workgroup (CG <= kBlock, checked by the launcher)
    const int ty = threadIdx.x / CG, tx = threadIdx.x - ty * CG;
    if (ty >= TY) return;
    int p[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) p[c] = perm[tx * VEC + c];
    for (int64_t r = (int64_t)blockIdx.x * TY + ty; r < n; r += (int64_t)gridDim.x * TY) {
        const float *xr = x + r * D;
        float v[VEC];
#pragma unroll
        for (int c = 0; c < VEC; ++c) v[c] = xr[p[c]];
        if constexpr (VEC == 4) reinterpret_cast<float4 *>(y + r * D)[tx] = make_float4(v[0], v[1], v[2], v[3]);
        else y[r * D + tx] = v[0];
    }
}

__global__ __launch_bounds__(kBlock) void permute_wide_kernel(const int32_t *__restrict__ perm, int D,
                                                              const float *__restrict__ x, int64_t n,
                                                              float *__restrict__ y)
{
    for (int64_t r = blockIdx.x; r < n; r += gridDim.x)
        for (int i = threadIdx.x; i < D; i += kBlock) y[r * D + i] = x[r * D + perm[i]];
}

int launch_permute(const int32_t *perm, int D, const float *x, int64_t n, float *y, hipStream_t st)
{
    if (n <= 0) return CVTMI_OK;
    const unsigned grid = 256 * 8;
    if ((D & 3) == 0 && D / 4 <= kBlock && ((uintptr_t)y & 15) == 0)
        hipLaunchKernelGGL(permute_kernel<4>, dim3(grid), dim3(kBlock), 0, st, perm, D, x, n, y);
    else if (D <= kBlock)
        hipLaunchKernelGGL(permute_kernel<1>, dim3(grid), dim3(kBlock), 0, st, perm, D, x, n, y);
    else
        hipLaunchKernelGGL(permute_wide_kernel, dim3(grid), dim3(kBlock), 0, st, perm, D, x, n, y);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

using f32x16 = __attribute__((ext_vector_type(16))) float;

// 512 threads = 8 waves per CU (2 per SIMD).  Each wave walks its 32-row slab in K-chunks of 32 columns:
// chunk c+1 is fetched from HBM with coalesced 16-byte loads (4 per lane) while chunk c feeds 64 MFMAs
// from a wave-private, double-buffered 32 x 33 LDS tile (stride 33: the 32 lanes that read one k column
// hit 32 banks).  With two waves per SIMD the matrix pipe also stays busy across a wave's prologue/epilogue.
constexpr int ROT_THREADS = 512;
constexpr int ROT_KC = 32;             // columns per chunk
constexpr int ROT_XLD = ROT_KC + 1;    // padded chunk stride (floats)

template <int NT>
__global__ __launch_bounds__(ROT_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void rotate_gemm_kernel(
    const float *__restrict__ R, const float *__restrict__ x, int64_t n, float *__restrict__ y)
{
    constexpr int D = 32 * NT;
    constexpr int LD = D + 1;  // padded leading dimension of R in LDS
    constexpr int WAVES = ROT_THREADS / 64;
    constexpr int NCH = D / ROT_KC;  // chunks per slab (== NT)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Rs = smem;                                   // [D][LD], Rs[j][k] = R[j][k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *Xw = smem + D * LD + wave * (2 * 32 * ROT_XLD);  // two chunk buffers per wave
    for (int i = tid; i < D * D; i += ROT_THREADS) {
        const int j = i / D, k = i - j * D;
        Rs[j * LD + k] = R[i];
    }
    __syncthreads();
    const int li = lane & 31, lk = lane >> 5;
    const float *Rl = Rs + li * LD + lk;
    // chunk loader: lane handles float4 number f = p*64 + lane (p < 4) of the 32 x 32 chunk: row f/8, cols 4*(f%8)
    const int64_t n_slabs = (n + 31) / 32;
    const int64_t first = (int64_t)blockIdx.x * WAVES + wave, step = (int64_t)gridDim.x * WAVES;
    auto fetch = [&](int64_t slab, int c, float4 (&v)[4]) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int f = p * 64 + lane, r = f >> 3, c4 = f & 7;
            int64_t row = slab * 32 + r;
            row = row < n ? row : n - 1;  // clamped: tail rows are never stored
            v[p] = *reinterpret_cast<const float4 *>(x + row * D + c * ROT_KC + c4 * 4);
        }
    };
    auto stash = [&](float *buf, const float4 (&v)[4]) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int f = p * 64 + lane, r = f >> 3, c4 = f & 7;
            float *d = buf + r * ROT_XLD + c4 * 4;
            d[0] = v[p].x; d[1] = v[p].y; d[2] = v[p].z; d[3] = v[p].w;
        }
    };
    float4 pre[4];
    if (first < n_slabs) fetch(first, 0, pre);
    int cur = 0;
    // Round 6: a slab's 64 result stores and the zeroing of its accumulators used to sit between two slabs' matrix instructions -- with
    // only four chunks per slab at D = 128 that fill and drain was a third of a slab's life (PMC: 70 % of the wave cycles issue-stalled,
    // 0.57 of the fp32 matrix peak).  Now the accumulators are double-buffered: slab s + 1 starts from SrcC = 0 (no zeroing moves) into
    // the other register set, and the stores of slab s are issued four at a time between the matrix instructions of its first chunk
    // (16 k-steps x NT stores = the slab's NT x 16 values), where they run in the shadow of the 64-cycle instructions.
    // Same products in the same k order per accumulator: bits unchanged.
    f32x16 accA[NT], accB[NT];
    auto store_e = [&](const f32x16 (&acc)[NT], int64_t row0, int e) {   // C/D layout of 32x32: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
        const int r = (e & 3) + 8 * (e >> 2) + 4 * lk;
        if (row0 + r < n) {
#pragma unroll
            for (int t = 0; t < NT; ++t) y[(row0 + r) * D + t * 32 + li] = acc[t][e];
        }
    };
    auto slab_body = [&](f32x16 (&acc)[NT], const f32x16 (&prev)[NT], bool has_prev, int64_t prev_row0, int64_t slab) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            float *buf = Xw + cur * (32 * ROT_XLD);
            stash(buf, pre);  // chunk c of this slab -> LDS (the buffer was last read two chunks ago)
            // next chunk (or chunk 0 of the next slab) goes in flight now, lands while the MFMAs below run
            if (c + 1 < NCH) fetch(slab, c + 1, pre);
            else if (slab + step < n_slabs) fetch(slab + step, 0, pre);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // A[i][k] = X[row0+i][k0+k] (lane: i = lane&31, k = lane>>5);  B[k][j] = R[j0+j][k0+k];
            // k ascends with the instruction order: one product per k, single rounding (== fmaf chain)
            const float *xa = buf + li * ROT_XLD + lk;
#pragma unroll
            for (int kk = 0; kk < ROT_KC; kk += 2) {
                const float a = xa[kk];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float b = Rl[(t * 32) * LD + c * ROT_KC + kk];
                    if (c == 0 && kk == 0) {
                        const f32x16 zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, zero, 0, 0, 0);
                    } else {
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
                    }
                }
                if (c == 0 && has_prev) store_e(prev, prev_row0, kk >> 1);   // (wave-uniform) the previous slab's values e = kk / 2 of every tile
            }
            cur ^= 1;
        }
    };
    bool pending_a = false, pending_b = false;
    int64_t row_a = 0, row_b = 0;
    for (int64_t slab = first; slab < n_slabs; slab += 2 * step) {
        slab_body(accA, accB, pending_b, row_b, slab);
        pending_b = false; pending_a = true; row_a = slab * 32;
        if (slab + step < n_slabs) {
            slab_body(accB, accA, true, row_a, slab + step);
            pending_a = false; pending_b = true; row_b = (slab + step) * 32;
        }
    }
    if (pending_a) {
#pragma unroll
        for (int e = 0; e < 16; ++e) store_e(accA, row_a, e);
    }
    if (pending_b) {
#pragma unroll
        for (int e = 0; e < 16; ++e) store_e(accB, row_b, e);
    }
}

int launch_rotate_gemm(const float *R, int D, const float *x, int64_t n, float *y, hipStream_t st)
{
    if (n <= 0) return CVTMI_OK;
    if (D % 32 != 0 || D < 32 || D > 128)
        return fail(CVTMI_EUNSUPPORTED, "rotate_gemm: D=%d (built for 32, 64, 96, 128)", D);
    const int NT = D / 32;
    const size_t lds = ((size_t)D * (D + 1) + (size_t)(ROT_THREADS / 64) * 2 * 32 * ROT_XLD) * sizeof(float);
    const int64_t n_slabs = (n + 31) / 32;
    int64_t blocks = (n_slabs + 7) / 8;
    if (blocks > 256) blocks = 256;  // one persistent 8-wave workgroup per CU
#define CVTMI_ROT(T)                                                                                              \
    case T:                                                                                                       \
        CVTMI_HIP(hipFuncSetAttribute((const void *)rotate_gemm_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)lds));                                                                 \
        hipLaunchKernelGGL((rotate_gemm_kernel<T>), dim3((unsigned)blocks), dim3(ROT_THREADS), lds, st, R, x, n, y); \
        break;
    switch (NT) {
        CVTMI_ROT(1) CVTMI_ROT(2) CVTMI_ROT(3) CVTMI_ROT(4)
        default: return fail(CVTMI_EUNSUPPORTED, "rotate_gemm: D=%d", D);
    }
#undef CVTMI_ROT
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

}  // namespace cvtmi
