// How exact is the fp32 accumulation of v_mfma_f32_32x32x16_bf16?  The filter kernels (opq_encode.hip, assign_mfma.hip,
// flat_mfma.hip) budget 2u per accumulated term (u = 2^-24) relative to sum |terms|.  This measures it: random bf16
// operands with a wide exponent spread, K = 16 * chain products per output, compared with the exact double sum.
// hipcc --offload-arch=gfx950 -O3 -o mfma_bf16_err mfma_bf16_err.hip && ./mfma_bf16_err
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A [32][K] and B [32][K] as bf16 bits; out [32][32]: out[i][j] = sum_k A[i][k] * B[j][k]
__global__ void k(const unsigned short *A, const unsigned short *B, int K, float *out)
{
    const int lane = threadIdx.x, li = lane & 31, lk = lane >> 5;
    f32x16 acc = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int c = 0; c < K / 16; ++c) {
        union { unsigned short s[8]; bf16x8 v; } a, b;
        for (int e = 0; e < 8; ++e) { a.s[e] = A[li * K + 16 * c + 8 * lk + e]; b.s[e] = B[li * K + 16 * c + 8 * lk + e]; }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
    }
    for (int e = 0; e < 16; ++e) out[((e & 3) + 8 * (e >> 2) + 4 * lk) * 32 + li] = acc[e];
}
static float bf(unsigned short s) { unsigned int u = (unsigned int)s << 16; float f; memcpy(&f, &u, 4); return f; }
int main()
{
    srand(7);
    double worst_u = 0, worst_rel_terms = 0;
    for (int trial = 0; trial < 400; ++trial) {
        const int K = (trial % 4 == 0) ? 16 : (trial % 4 == 1) ? 48 : (trial % 4 == 2) ? 128 : 400;
        const int spread = trial % 3 == 0 ? 2 : trial % 3 == 1 ? 12 : 30;  // exponent spread of the operands
        std::vector<unsigned short> A(32 * K), B(32 * K);
        for (auto *v : { &A, &B })
            for (auto &s : *v) {
                const float f = ldexpf((float)rand() / RAND_MAX + 0.5f, rand() % (spread + 1) - spread / 2) * ((rand() & 1) ? 1.f : -1.f);
                unsigned int u; memcpy(&u, &f, 4); s = (unsigned short)(u >> 16);
            }
        unsigned short *dA, *dB; float *dO;
        hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dO, 32 * 32 * 4);
        hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, K, dO);
        std::vector<float> O(1024); hipMemcpy(O.data(), dO, 4096, hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double s = 0, sa = 0;
                for (int kk = 0; kk < K; ++kk) { const double p = (double)bf(A[i * K + kk]) * bf(B[j * K + kk]); s += p; sa += fabs(p); }
                const double err = fabs((double)O[i * 32 + j] - s);
                const double per_term_u = err / sa / ldexp(1.0, -24) / K;  // error per accumulated term, in u * sum|terms|
                if (per_term_u > worst_u) worst_u = per_term_u;
                if (err / sa / ldexp(1.0, -24) > worst_rel_terms) worst_rel_terms = err / sa / ldexp(1.0, -24);
            }
        hipFree(dA); hipFree(dB); hipFree(dO);
    }
    printf("worst error: %.3f u x sum|terms| in total, %.4f u per accumulated term (the kernels budget 2 u per term)\n", worst_rel_terms, worst_u);
    return 0;
}
