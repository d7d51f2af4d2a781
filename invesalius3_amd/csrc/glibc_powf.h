// glibc_powf.h -- f32 `powf` with the bits glibc's gives, on the device.
//
// The reference's contour MIP raises to a float exponent with Rust's `f32::powf` (invesalius_rs/src/mips.rs:211), which
// is the platform libm's `powf`: glibc 2.28+ ships the ARM optimized-routines algorithm (sysdeps/ieee754/flt-32/e_powf.c,
// e_powf_log2_data.c, e_exp2f_data.c): log2(x) from a 16-entry table + a degree-5 polynomial, exp2 from a 32-entry table +
// a degree-3 polynomial, everything in IEEE double, one rounding to float at the end.  It is NOT correctly rounded (about
// 0.82 ULP + 0.5), so "pow in double, rounded once" differs from it by an ulp on a small share of inputs -- which is what
// tests/test_gpu_rays.py::test_fcm_volume_other_exponents used to tolerate.  This header restates the published algorithm with
// its published constants (checked here against /lib/x86_64-linux-gnu/libm.so.6 of glibc 2.35: tests/test_oracle_mips.py and
// tools/check_powf.c, both variants below bit for bit on 10^9 inputs, special cases included).
//
// glibc builds that function twice on x86-64 (sysdeps/x86_64/fpu/multiarch/e_powf.c): the plain one, and `__powf_fma`
// compiled with -mfma -mavx2, chosen at load time when the CPU has FMA and AVX2 -- every `a * b + c` of the source is then one
// fused operation, and a few results in a million change in their last bit.  FMA = true restates that build (the products
// feeding an addition fused, exactly the contractions GCC makes), FMA = false the portable one; the host picks the variant
// the machine's own libm would run (ivx_powf_variant(): the same CPU test as glibc's selector), so the GPU's bits are the
// bits the reference produces on the box it runs on.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define GPF_HD __host__ __device__ __forceinline__
#else
#define GPF_HD static inline
#endif

namespace glibc_powf {

struct LogEntry {
    double invc, logc;
};

GPF_HD uint32_t asuint(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
GPF_HD float asfloat(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
GPF_HD uint64_t asuint64(double d) {
    uint64_t u;
    memcpy(&u, &d, 8);
    return u;
}
GPF_HD double asdouble(uint64_t u) {
    double d;
    memcpy(&d, &u, 8);
    return d;
}

template <bool FMA> GPF_HD double mad(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return FMA ? fma(a, b, c) : __dadd_rn(__dmul_rn(a, b), c);
#else
    if (FMA) return __builtin_fma(a, b, c);
    volatile double p = a * b; // (host build: keep the product's own rounding whatever -ffp-contract says)
    return p + c;
#endif
}

// e_powf_log2_data.c (POWF_SCALE == 1: TOINT_INTRINSICS is 0 on x86-64)
GPF_HD LogEntry log_tab(int i) {
    constexpr double T[16][2] = {
        {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
        {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
        {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
        {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
        {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
        {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
        {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
        {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2},
    };
    return LogEntry{T[i][0], T[i][1]};
}

// e_exp2f_data.c: 2^(i/32) with the exponent bits pre-adjusted
GPF_HD uint64_t exp2_tab(int i) {
    constexpr uint64_t T[32] = {
        0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
        0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
        0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
        0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
        0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
        0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
        0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
        0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
    };
    return T[i];
}

// 0 = y is not an integer, 1 = odd integer, 2 = even integer
GPF_HD int checkint(uint32_t iy) {
    const int e = (int)(iy >> 23 & 0xff);
    if (e < 0x7f) return 0;
    if (e > 0x7f + 23) return 2;
    if (iy & ((1u << (0x7f + 23 - e)) - 1u)) return 0;
    if (iy & (1u << (0x7f + 23 - e))) return 1;
    return 2;
}
GPF_HD bool zeroinfnan(uint32_t ix) { return 2u * ix - 1u >= 2u * 0x7f800000u - 1u; }
GPF_HD bool issignaling(uint32_t ix) { return 2u * (ix ^ 0x00400000u) > 2u * 0x7fc00000u; }

// lt: the 16 (invc, logc) pairs as 32 doubles; et: the 32 exp2 words -- the caller may hand in copies that are cheaper to index
// (the kernels keep them in LDS); the overloads without tables index the constant arrays above
template <bool FMA> GPF_HD double log2_inline(uint32_t ix, const double *lt) {
    constexpr double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2, A3 = -0x1.7154748bef6c8p-1,
                     A4 = 0x1.71547652ab82bp0;
    // x = 2^k z, z in [OFF, 2 OFF) exact; the i-th of 16 subintervals holds z, c is near its centre
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> (23 - 4)) % 16u);
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int32_t)top >> 23; // arithmetic shift
    const double invc = lt[2 * i], logc = lt[2 * i + 1];
    const double z = (double)asfloat(iz);
    // log2(x) = log1p(z/c - 1)/ln2 + log2(c) + k
    const double r = mad<FMA>(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    double y = mad<FMA>(A0, r, A1);
    const double p = mad<FMA>(A2, r, A3);
    const double r4 = r2 * r2;
    double q = mad<FMA>(A4, r, y0);
    q = mad<FMA>(p, r2, q);
    y = mad<FMA>(y, r4, q);
    return y;
}

template <bool FMA> GPF_HD float exp2_inline(double xd, uint32_t sign_bias, const uint64_t *et) {
    constexpr double SHIFT = 0x1.8p+52 / 32.0, C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;
    // x = k/N + r with r in [-1/(2N), 1/(2N)]
    double kd = xd + SHIFT;
    const uint64_t ki = asuint64(kd);
    kd -= SHIFT; // k/N
    const double r = xd - kd;
    // exp2(x) = 2^(k/N) * 2^r ~= s * (C0 r^3 + C1 r^2 + C2 r + 1)
    uint64_t t = et[(int)(ki % 32u)];
    const uint64_t ski = ki + sign_bias;
    t += ski << (52 - 5);
    const double s = asdouble(t);
    const double z = mad<FMA>(C0, r, C1);
    const double r2 = r * r;
    double y = mad<FMA>(C2, r, 1.0);
    y = mad<FMA>(z, r2, y);
    y = y * s;
    return (float)y;
}

// __powf (e_powf.c).  Error paths return what __math_oflowf / _uflowf / _invalidf / _divzerof return; errno and the
// floating-point exception flags are the host's business and do not exist here.
template <bool FMA> GPF_HD float powf_glibc(float x, float y, const double *lt, const uint64_t *et) {
    constexpr uint32_t SIGN_BIAS = 1u << (5 + 11);
    uint32_t sign_bias = 0;
    uint32_t ix = asuint(x);
    const uint32_t iy = asuint(y);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || zeroinfnan(iy)) {
        // either (x < 0x1p-126 or inf or nan) or (y is 0 or inf or nan)
        if (zeroinfnan(iy)) {
            if (2u * iy == 0u) return issignaling(ix) ? x + y : 1.0f;
            if (ix == 0x3f800000u) return issignaling(iy) ? x + y : 1.0f;
            if (2u * ix > 2u * 0x7f800000u || 2u * iy > 2u * 0x7f800000u) return x + y;
            if (2u * ix == 2u * 0x3f800000u) return 1.0f;
            if ((2u * ix < 2u * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f; // |x| < 1 && y == inf, or |x| > 1 && y == -inf
            return y * y;
        }
        if (zeroinfnan(ix)) {
            float x2 = x * x;
            if ((ix & 0x80000000u) && checkint(iy) == 1) x2 = -x2;
            // (x == 0 with y < 0: __math_divzerof = +-inf, which 1 / x2 is as well)
            return (iy & 0x80000000u) ? 1.0f / x2 : x2;
        }
        // x and y are non-zero finite
        if (ix & 0x80000000u) { // finite x < 0
            const int yint = checkint(iy);
            if (yint == 0) return asfloat(0x7fc00000u); // __math_invalidf: (x - x) / (x - x), a quiet NaN (its sign is the FPU's)
            if (yint == 1) sign_bias = SIGN_BIAS;
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) { // normalise a subnormal x so that its exponent becomes negative
            ix = asuint(x * 0x1p23f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    const double logx = log2_inline<FMA>(ix, lt);
    const double ylogx = (double)y * logx; // cannot overflow: y is single precision
    if ((asuint64(ylogx) >> 47 & 0xffff) >= (asuint64(126.0) >> 47)) {
        // |y * log2(x)| >= 126
        if (ylogx > 0x1.fffffffd1d571p+6) return sign_bias ? -__builtin_inff() : __builtin_inff();             // __math_oflowf
        if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;                                                    // __math_uflowf
    }
    return exp2_inline<FMA>(ylogx, sign_bias, et);
}

// fills caller-provided copies of the two tables (lt: 32 doubles, et: 32 words): entry i by the caller's lane i < 32
GPF_HD void copy_table_entry(int i, double *lt, uint64_t *et) {
    const LogEntry e = log_tab(i >> 1);
    lt[i] = (i & 1) ? e.logc : e.invc;
    et[i] = exp2_tab(i);
}

template <bool FMA> GPF_HD float powf_glibc(float x, float y) {
    double lt[32];
    uint64_t et[32];
    for (int i = 0; i < 32; i++) copy_table_entry(i, lt, et);
    return powf_glibc<FMA>(x, y, lt, et);
}

} // namespace glibc_powf
