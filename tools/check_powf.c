// tools/check_powf.c -- invesalius3_amd/csrc/glibc_powf.h against the libm of this machine, bit for bit.
//   g++ -O2 -ffp-contract=off -o /tmp/check_powf -x c++ tools/check_powf.c -lm && /tmp/check_powf [millions]
// Reports, for each of the two restated builds (plain / FMA), how many of the inputs differ from libm's powf -- the one
// with 0 is the variant this machine's glibc runs (x86-64: the FMA build when the CPU has FMA + AVX2) -- and checks that
// ivx's host-side selector names that variant.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "../invesalius3_amd/csrc/glibc_powf.h"

static inline int same(float a, float b) {
    if (a != a && b != b) return 1; // any NaN
    return glibc_powf::asuint(a) == glibc_powf::asuint(b);
}

int main(int argc, char **argv) {
    const long millions = argc > 1 ? atol(argv[1]) : 100;
    unsigned long long s = 88172645463325252ull;
    long bad[2] = {0, 0}, n = 0, differ = 0;
    auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (long i = 0; i < millions * 1000000L; i++) {
        const unsigned long long r = next();
        float x, y;
        switch (i & 7) {
        case 0: case 1: case 2: // the contour MIP's domain: base in [0, 1], exponents a user would type
            x = (float)((r >> 40) & 0xffffff) / 16777216.0f;
            y = (float)((r >> 8) & 0xfffff) / 65536.0f;
            break;
        case 3: // base = 1 - |d / gm| as the kernel forms it
            { const float d = (float)((r >> 40) & 0xffff), g = d + (float)((r >> 8) & 0xffff) + 1.0f; x = 1.0f - fabsf(d / g); y = 0.25f * (float)(1 + (r & 31)); }
            break;
        case 4: case 5: // arbitrary bit patterns (NaNs, infinities, subnormals, negatives)
            x = glibc_powf::asfloat((uint32_t)(r >> 32));
            y = glibc_powf::asfloat((uint32_t)r);
            break;
        case 6: // negative bases with integer exponents, overflow / underflow range
            x = -(float)((r >> 40) & 0xffff) / 256.0f;
            y = (float)((long)((r >> 8) & 0xff) - 128);
            break;
        default: // near 1, large exponents
            x = 1.0f + ((float)((r >> 40) & 0xffff) - 32768.0f) / 1048576.0f;
            y = ((float)((r >> 8) & 0xffffff) - 8388608.0f) / 16.0f;
        }
        const float ref = powf(x, y), a = glibc_powf::powf_glibc<false>(x, y), b = glibc_powf::powf_glibc<true>(x, y);
        bad[0] += !same(ref, a);
        bad[1] += !same(ref, b);
        differ += !same(a, b);
        n++;
        if ((!same(ref, a) && !same(ref, b)) && bad[0] + bad[1] < 40) printf("x=%a y=%a libm=%a plain=%a fma=%a\n", x, y, ref, a, b);
    }
    printf("inputs %ld: plain build differs from libm on %ld, FMA build on %ld; the two builds differ from each other on %ld\n", n, bad[0], bad[1], differ);
    const int fma_cpu = __builtin_cpu_supports("fma") && __builtin_cpu_supports("avx2");
    printf("cpu has fma+avx2: %d -> glibc's selector runs the %s build\n", fma_cpu, fma_cpu ? "FMA" : "plain");
    return (fma_cpu ? bad[1] : bad[0]) == 0 ? 0 : 1;
}
