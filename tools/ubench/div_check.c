// Brute-force check of the 3-instruction correctly rounded division used by sq8.hip:
//   y = RN(1/b) (once per row / column), q = RN(a*y), e = fma(-b, q, a), q' = fma(e, y, q)  ==  RN(a/b)
// for b normal in [2^-40, 2^40] whose significand is not all ones and |a| in [2^-60, 2^60].
// Build: gcc -O2 -mfma -o div_check div_check.c -lm ; prints the number of mismatches (must be 0).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static uint64_t s = 0x9E3779B97F4A7C15ull;
static inline uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static inline float mk(uint32_t sign, int e, uint32_t man) { uint32_t u = (sign << 31) | ((uint32_t)(e + 127) << 23) | (man & 0x7fffffu); float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static long check(float a, float b)
{
    const float y = 1.0f / b;
    const float q = a * y;
    const float e = fmaf(-b, q, a);
    const float q1 = fmaf(e, y, q);
    const float t = a / b;
    return bits(q1) != bits(t);
}
int main(void)
{
    long bad = 0, n = 0;
    // (1) every significand of b (exponent 0), 48 random a each
    for (uint32_t man = 0; man < 0x7fffffu; ++man) {
        const float b = mk(0, 0, man);
        for (int i = 0; i < 48; ++i) { bad += check(mk(rnd() & 1, (int)(rnd() % 8) - 4, (uint32_t)rnd()), b); ++n; }
    }
    // (2) random exponents in range
    for (long i = 0; i < 600000000L; ++i) {
        uint32_t man = (uint32_t)rnd() & 0x7fffffu;
        if (man == 0x7fffffu) man = 0;
        const float b = mk(0, (int)(rnd() % 81) - 40, man);
        const float a = mk(rnd() & 1, (int)(rnd() % 121) - 60, (uint32_t)rnd());
        bad += check(a, b); ++n;
    }
    // (3) a near powers of two / with few bits, b with few bits
    for (long i = 0; i < 100000000L; ++i) {
        uint32_t mb = ((uint32_t)rnd() & 0x7fffffu) & ~((1u << (rnd() % 23)) - 1u);
        uint32_t ma = ((uint32_t)rnd() & 0x7fffffu) | ((rnd() & 1) ? ((1u << (rnd() % 23)) - 1u) : 0u);
        if (mb == 0x7fffffu) mb = 0;
        bad += check(mk(0, (int)(rnd() % 121) - 60, ma), mk(0, (int)(rnd() % 81) - 40, mb)); ++n;
    }
    printf("checked %ld quotients, mismatches %ld\n", n, bad);
    return bad != 0;
}
