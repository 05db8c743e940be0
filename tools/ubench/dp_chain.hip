// Latency vs issue rate of a dependent v_add_f64 chain on gfx950 (what bounds the sequential L2-norm fold).
// hipcc --offload-arch=gfx950 -O3 -o dp_chain dp_chain.hip && ./dp_chain
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH>
__global__ void chain(const double *in, double *out, int n, long long *cyc)
{
    double acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = in[threadIdx.x + c];
    const double t = in[threadIdx.x + 7];
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __dadd_rn(acc[c], t);
    }
    const long long t1 = clock64();
    double s = 0;
    for (int c = 0; c < CH; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main()
{
    double *in, *out; long long *cyc, h;
    hipMalloc(&in, 4096 * 8); hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
    hipMemset(in, 0, 4096 * 8);
    const int n = 100000;
#define RUN(CH, TH) chain<CH><<<1, TH>>>(in, out, n, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); \
    printf("chains/lane %d, waves %d: %.2f cycles per add step (%.2f per add)\n", CH, TH / 64, (double)h / n, (double)h / n / CH);
    RUN(1, 64) RUN(2, 64) RUN(4, 64) RUN(8, 64) RUN(1, 256) RUN(1, 512) RUN(1, 1024) RUN(2, 1024)
    return 0;
}
