// dist_func_check -- SpaceInterface::get_dist_func() of the three built-in spaces on pseudo-random vectors:
//   dist_func_check <out.bin>
// writes, per case, int32 metric, int32 dim, the two vectors and the distance (fp32 / int32 bits) so that the test can hold
// them to the checker (tests/test_host_dist_func.py).  Host only: nothing here touches the GPU.
#include <cstdio>
#include <cstdint>
#include <vector>

#include "../hnswlib/hnswlib.h"

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

int main(int argc, char **argv)
{
    if (argc != 2) { fprintf(stderr, "usage: dist_func_check <out.bin>\n"); return 2; }
    FILE *f = fopen(argv[1], "wb");
    if (!f) return 1;
    const int dims[] = { 1, 5, 7, 20, 36, 64, 128, 100, 512 };
    for (int metric = 0; metric < 3; ++metric)
        for (int di = 0; di < 9; ++di)
            for (int rep = 0; rep < 4; ++rep) {
                const int32_t d = dims[di];
                fwrite(&metric, 4, 1, f); fwrite(&d, 4, 1, f);
                if (metric == 2) {
                    std::vector<unsigned char> a(d), b(d);
                    for (int i = 0; i < d; ++i) { a[i] = (unsigned char)rnd(); b[i] = (unsigned char)rnd(); }
                    hnswlib::L2SpaceI sp(d);
                    const int r = sp.get_dist_func()(a.data(), b.data(), sp.get_dist_func_param());
                    fwrite(a.data(), 1, d, f); fwrite(b.data(), 1, d, f); fwrite(&r, 4, 1, f);
                } else {
                    std::vector<float> a(d), b(d);
                    for (int i = 0; i < d; ++i) { a[i] = ((int)(rnd() % 20001) - 10000) * 1e-4f; b[i] = ((int)(rnd() % 20001) - 10000) * 3e-4f; }
                    float r;
                    if (metric == 0) { hnswlib::InnerProductSpace sp(d); r = sp.get_dist_func()(a.data(), b.data(), sp.get_dist_func_param()); }
                    else { hnswlib::L2Space sp(d); r = sp.get_dist_func()(a.data(), b.data(), sp.get_dist_func_param()); }
                    fwrite(a.data(), 4, d, f); fwrite(b.data(), 4, d, f); fwrite(&r, 4, 1, f);
                }
            }
    fclose(f);
    printf("OK\n");
    return 0;
}
