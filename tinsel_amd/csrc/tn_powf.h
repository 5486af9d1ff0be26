// tn_powf.h -- powf as glibc 2.35 evaluates it on an FMA-capable x86-64 host (sysdeps/ieee754/flt-32/e_powf.c,
// the __powf_fma ifunc variant: log2 of x from a 16-entry table + degree-5 polynomial in double, times y,
// then exp2 from the 32-entry table of exp2f).  The display stage (ToneMap / LinearToSrgb: util.h:25-42,
// maths.h:1545-1555) is the only caller; restating the host algorithm keeps the displayed image
// bit-identical to the reference's.  Table values read from this image's libm.so.6 .rodata.
//
// Plain C++ (no HIP intrinsics) so that the same text is checked on the host against libm
// (tests/test_display.py builds it with g++ and sweeps the float range).
#pragma once

#include <stdint.h>
#include <string.h>

#ifndef TN_HD
#define TN_HD inline
#define TN_POWF_HOST_ONLY
#endif

#ifdef __HIP_DEVICE_COMPILE__
#define TN_POWF_CONST __device__ const
#else
#define TN_POWF_CONST static const
#endif

namespace tn {

TN_POWF_CONST double kPowfLog2Tab[16][2] = {    // { invc, logc }
    { 0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2 }, { 0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2 },
    { 0x1.49539f0f010b0p+0, -0x1.7418b0a1fb77bp-2 }, { 0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2 },
    { 0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2 }, { 0x1.25e227b0b8ea0p+0, -0x1.97c1d1b3b7af0p-3 },
    { 0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3 }, { 0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4 },
    { 0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5 }, { 0x1.0000000000000p+0, 0x0.0p+0 },
    { 0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4 }, { 0x1.ca4b31f026aa0p-1, 0x1.476a9543891bap-3 },
    { 0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3 }, { 0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2 },
    { 0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2 }, { 0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2 },
};

TN_POWF_CONST unsigned long long kPowfExp2Tab[32] = {
    0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL, 0x3fef72b83c7d517bULL,
    0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL, 0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL,
    0x3feedea64c123422ULL, 0x3feece086061892dULL, 0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL,
    0x3feea47eb03a5585ULL, 0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,
    0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL, 0x3feee89f995ad3adULL,
    0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL, 0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL,
    0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL,
};

TN_HD uint32_t powf_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
TN_HD float powf_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
TN_HD uint64_t powf_bits64(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
TN_HD double powf_double(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

// x^y for the display stage's domain: any x (negative or NaN -> NaN), finite y > 0 that is not an integer.
// `log2tab` / `exp2tab`: the two tables above or copies of them (the display kernel keeps copies in LDS).
template <class LogTab, class ExpTab>
TN_HD float m_powf_tab(float x, float y, const LogTab& log2tab, const ExpTab& exp2tab)
{
    uint32_t ix = powf_bits(x);
    if (x != x || ix >= 0x80000000u)
    {
        // NaN, or negative base with a non-integer exponent (-0 included: pow(-0, y>0 non-odd) = +0)
        if (x == 0.0f)
            return 0.0f;
        if (ix == 0xff800000u)
            return powf_float(0x7f800000u);     // pow(-inf, y > 0 not an odd integer) = +inf
        return powf_float(0x7fc00000u);
    }
    if (ix == 0u)
        return 0.0f;
    if (ix == 0x7f800000u)
        return x;
    if (ix < 0x00800000u)
    {
        // subnormal x: normalise
        ix = powf_bits(x*0x1p23f);
        ix &= 0x7fffffffu;
        ix -= 23u << 23;
    }

    // log2(x) = log2(z/c) + log2(c) + k, z in [OFF, 2 OFF)
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) & 15u;
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int32_t)top >> 23;
    const double invc = log2tab[i][0], logc = log2tab[i][1];
    const double z = (double)powf_float(iz);

    const double r = __builtin_fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    double yy = __builtin_fma(0x1.27616c9496e0bp-2, r, -0x1.71969a075c67ap-2);
    const double p = __builtin_fma(0x1.ec70a6ca7baddp-2, r, -0x1.7154748bef6c8p-1);
    const double r2 = r*r;
    double q = __builtin_fma(0x1.71547652ab82bp+0, r, y0);
    const double r4 = r2*r2;
    q = __builtin_fma(p, r2, q);
    yy = __builtin_fma(yy, r4, q);

    const double ylogx = (double)y*yy;
    if (((powf_bits64(ylogx) >> 47) & 0xffffu) > 0x80beu)
    {
        // |y log2 x| >= 126: possible over/underflow (round-to-nearest only)
        if (ylogx > 0x1.fffffffd1d571p+6)
            return powf_float(0x7f800000u);
        if (ylogx <= -150.0)
            return 0.0f;
        if (ylogx < -149.0)
            return powf_float(1u);          // 0x1.4p-75f*0x1.4p-75f rounds to the smallest subnormal
    }

    // 2^(y log2 x): k/32 + r
    double kd = ylogx + 0x1.8p+47;
    const uint64_t ki = powf_bits64(kd);
    kd -= 0x1.8p+47;
    const double rr = ylogx - kd;
    uint64_t t = exp2tab[ki & 31u];
    t += ki << 47;
    const double s = powf_double(t);
    const double zz = __builtin_fma(0x1.c6af84b912394p-5, rr, 0x1.ebfce50fac4f3p-3);
    const double rr2 = rr*rr;
    double e = __builtin_fma(0x1.62e42ff0c52d6p-1, rr, 1.0);
    e = __builtin_fma(zz, rr2, e);
    return (float)(e*s);
}

TN_HD float m_powf(float x, float y) { return m_powf_tab(x, y, kPowfLog2Tab, kPowfExp2Tab); }

} // namespace tn
