// tn_math.h -- fp32 vector / quaternion / RNG primitives for the gfx950 path tracer.
//
// Every function states which reference expression it evaluates; the OPERATION ORDER is
// part of the contract (the translation unit is compiled with -ffp-contract=off and without
// fast-math, so each line is the same sequence of IEEE-754 binary32 roundings as the CPU
// oracle's).  Where the reference silently widens to double (`a*(1.0/s)`, maths.h:242) the
// double step is a single division whose double->float rounding is innocuous
// (53 >= 2*24+2), so the correctly-rounded fp32 division used here is bit-identical.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define TN_HD __host__ __device__ __forceinline__
#define TN_D __device__ __forceinline__

namespace tn {

constexpr float kPi = 3.141592653589793f;          // maths.h:32
constexpr float k2Pi = 3.141592653589793f*2.0f;    // maths.h:33
constexpr float kInvPi = 1.0f/kPi;                 // maths.h:34
constexpr float kInv2Pi = 1.0f/k2Pi;               // maths.h:35
constexpr float kFltMax = 3.402823466e+38f;

struct V3
{
    float x, y, z;
    TN_HD V3() : x(0.0f), y(0.0f), z(0.0f) {}
    TN_HD explicit V3(float s) : x(s), y(s), z(s) {}
    TN_HD V3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};

struct Q4 { float x, y, z, w; };        // quaternion (maths.h:502-525), w = real part

struct Xform { V3 p; Q4 r; float s; };  // Transform (maths.h:575-589)

// maths.h:236-251
TN_HD V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
TN_HD V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
TN_HD V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
TN_HD V3 operator*(V3 a, float s) { return V3(a.x*s, a.y*s, a.z*s); }
TN_HD V3 operator*(float s, V3 a) { return V3(a.x*s, a.y*s, a.z*s); }
TN_HD V3 operator*(V3 a, V3 b) { return V3(a.x*b.x, a.y*b.y, a.z*b.z); }
// operator/(Vec3, Real s) == a*(1.0/s)  (maths.h:242)
TN_HD float rcpf_cr(float x);
TN_HD float sqrtf_cr(float x);
TN_HD V3 divs(V3 a, float s) { float r = rcpf_cr(s); return V3(a.x*r, a.y*r, a.z*r); }

// ---------------------------------------------------------------------------
// Correctly rounded 1/x and sqrt(x) without the compiler's general-purpose expansion.  hipcc expands an IEEE fp32 division into 12
// VALU instructions (v_div_scale x2, v_rcp, five fma/mul, v_div_fmas, v_div_fixup) and sqrtf into 16 (2^32 scaling of small
// operands, v_sqrt, two compare-and-step corrections, unscaling, class fix-up); k_bounce carries 228 divisions -- 110 of them
// reciprocals -- and 76 square roots: ~30 % of its static VALU count.  For ONE operand the space is small enough to PROVE a shorter
// sequence: the candidates below are compared with the compiler's `1.0f/x` and `sqrtf(x)` on ALL 2^32 bit patterns on the device
// (tinsel_hip_selftest_arith, tests/test_gpu_arith.py); the library is built with a variant that showed zero mismatches -- NaNs,
// infinities, zeros and denormals included.  Measured (profiles/r03_w_short_sqrt.md):
//   sqrt 21  v_rsq + one coupled Newton step inside the compiler's own 2^32 scaling, branch-free, 12 VALU: exact on all 2^32;
//            cornell +1.5 %, veach 4K +2.2 %, glass +1.2 %, ajax +0.6 %  -> the default
//   sqrt 11  the same behind a branch (2^-96 <= x < inf, else sqrtf): exact; +0.7 % (the branch splits the scheduler's blocks)
//   rcp 11   v_rcp + two Newton steps behind a branch (normal x, |x| < 2^126, else 1.0f/x): exact on all 2^32 but 4-5 % SLOWER --
//            a straight-line form needs the quotient rounded once into the denormal range, which is what v_div_fmas is for: the
//            compiler's expansion stays (variant 0); one guard per direction triple (TN_RCP3_GROUP) measured +-1 %: off
#ifndef TN_RCP_VARIANT
#define TN_RCP_VARIANT 0
#endif
#ifndef TN_RCP3_GROUP
#define TN_RCP3_GROUP 0
#endif
#ifndef TN_SQRT_VARIANT
#define TN_SQRT_VARIANT 21
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// v_rcp_f32 + two Newton steps (the second one is Markstein's correction of the quotient q = r)
__device__ __forceinline__ float rcp_refine(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
// variant 0: the compiler's division; 1: straight-line, v_div_fixup for zero / infinity / NaN (NOT exact: v_rcp_f32 flushes denormal
// operands and results -- kept for the self-test to find); 11: the refinement behind a range guard (normal x, |x| < 2^126), everything
// else through the compiler's division
template <int V> __device__ __forceinline__ float rcp_candidate(float x)
{
    static_assert(V == 0 || V == 1 || V == 11, "unknown reciprocal variant");
    if (V == 0)
        return 1.0f/x;
    if (V == 1)
        return __builtin_amdgcn_div_fixupf(rcp_refine(x), x, 1.0f);
    if (__builtin_expect(__builtin_isnormal(x) && __builtin_fabsf(x) < 0x1p126f, 1))
        return rcp_refine(x);
    return 1.0f/x;
}

// v_rsq_f32, s = x*y, h = y/2, one Newton step on s with fused residual
__device__ __forceinline__ float sqrt_refine(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float s = x*y;
    const float h = 0.5f*y;
    const float e = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(e, h, s);
}
// variant 0: the compiler's sqrtf; 1: the bare step, +-0 and +inf passed through (NOT exact below 2^-96, where the residual x - s*s
// leaves the normal range: kept for the self-test to find); 11: the step behind a guard (2^-96 <= x < inf), the rest through sqrtf;
// 21: no branch -- operands below 2^-96 scaled by 2^32 as the compiler's own expansion does, the root scaled back by 2^-16
template <int V> __device__ __forceinline__ float sqrt_candidate(float x)
{
    static_assert(V == 0 || V == 1 || V == 11 || V == 21, "unknown square-root variant");
    if (V == 0)
        return sqrtf(x);
    if (V == 1)     // (rsq gives inf / 0 at 0 / inf and the products NaN; negative x and NaN are NaN either way)
        return (x == 0.0f || x == __builtin_inff()) ? x : sqrt_refine(x);
    if (V == 21)
    {
        const bool small = x < 0x1p-96f;
        const float xs = small ? x*0x1p32f : x;
        float s = sqrt_refine(xs);
        s = small ? s*0x1p-16f : s;
        return __builtin_amdgcn_classf(xs, 0x260) ? xs : s;       // -0, +0, +inf pass through
    }
    if (__builtin_expect(x >= 0x1p-96f && x < __builtin_inff(), 1))
        return sqrt_refine(x);
    return sqrtf(x);
}
#endif

TN_HD float rcpf_cr(float x)
{
#if defined(__HIP_DEVICE_COMPILE__) && !(defined(TN_FAST) && TN_FAST)
    return rcp_candidate<TN_RCP_VARIANT>(x);
#else
    return 1.0f/x;
#endif
}
TN_HD float sqrtf_cr(float x)
{
#if defined(__HIP_DEVICE_COMPILE__) && !(defined(TN_FAST) && TN_FAST)
    return sqrt_candidate<TN_SQRT_VARIANT>(x);
#else
    return sqrtf(x);
#endif
}

TN_HD float dot(V3 a, V3 b) { return a.x*b.x + a.y*b.y + a.z*b.z; }                                   // maths.h:257
TN_HD V3 cross(V3 a, V3 b) { return V3(a.y*b.z - b.y*a.z, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x); }   // maths.h:256
// 1/sqrt(x) AS THE REFERENCE ROUNDS IT -- RN(1 / RN(sqrt x)): every normalize (maths.h:260 `a/Length(a)`, :261-273, the basis helpers,
// the quaternions) -- is again a function of ONE operand.  Variant 0: the two sequences above, one after the other (12 + 12 instructions).
// Variants 1-3: the root as in sqrtf_cr; its reciprocal needs no scaling -- the root of any finite positive fp32 lies in [2^-75, 2^64] --
// and v_rsq_f32's y ~ 1/sqrt(x) is within 2^-22 of it: two Newton steps against the ROUNDED root, v_div_fixup for the specials (+-0 ->
// +-inf, inf -> 0, negative -> NaN), 18 instructions.  Started from y itself (variant 1) the step lands on a tie where the root's
// mantissa is all ones and rounds to even: 255 wrong operands of 2^32, found by the self-test (and by one parity test, by luck).  From
// v_rcp_f32(root) (2) or from y one ulp up (3) it is exact on all 2^32 (tinsel_hip_selftest_arith op 2); 3 is the default:
// cornell +1.2 %, veach +0.9 % over variant 0 (profiles/r03_w_short_sqrt.md).
#ifndef TN_RSQRT_VARIANT
#define TN_RSQRT_VARIANT 3
#endif
#if defined(__HIP_DEVICE_COMPILE__)
template <int V> __device__ __forceinline__ float rsqrt_candidate(float x)
{
    static_assert(V >= 0 && V <= 3, "unknown reciprocal-square-root variant");
    if (V == 0)
        return rcp_candidate<TN_RCP_VARIANT>(sqrt_candidate<TN_SQRT_VARIANT>(x));
    const bool small = x < 0x1p-96f;
    const float xs = small ? x*0x1p32f : x;
    const float y = __builtin_amdgcn_rsqf(xs);
    float s = xs*y;
    const float h = 0.5f*y;
    float e = __builtin_fmaf(-s, s, xs);
    s = __builtin_fmaf(e, h, s);                                // RN(sqrt(xs)) (sqrt_candidate<21>)
    s = __builtin_amdgcn_classf(xs, 0x260) ? xs : s;            // -0, +0, +inf
    // the first guess of 1/s.  V == 1: y itself -- NOT exact: where s = 2^k (2 - 2^-23) the step from below lands on a tie and rounds to
    // even, 255 operands of 2^32 (kept for the self-test to find); V == 2: v_rcp_f32(s); V == 3: y one ulp up (a step from above never ties)
    const float r0 = V == 2 ? __builtin_amdgcn_rcpf(s) : V == 3 ? __uint_as_float(__float_as_uint(y) + 1u) : y;
    e = __builtin_fmaf(-s, r0, 1.0f);
    float r = __builtin_fmaf(e, r0, r0);
    e = __builtin_fmaf(-s, r, 1.0f);
    r = __builtin_fmaf(e, r, r);                                // RN(1/s) where s is finite and positive
    r = small ? r*0x1p16f : r;                                  // 1/sqrt(x) = 2^16 / sqrt(x 2^32): exact, the quotient is a normal number
    return __builtin_amdgcn_div_fixupf(r, s, 1.0f);
}
#endif
TN_HD float rsqrtf_cr(float x)
{
#if defined(__HIP_DEVICE_COMPILE__) && !(defined(TN_FAST) && TN_FAST)
    return rsqrt_candidate<TN_RSQRT_VARIANT>(x);
#else
    return 1.0f/sqrtf(x);
#endif
}

// the three reciprocals of a direction behind ONE guard (TN_RCP3_GROUP): a slab test's 1/d
TN_HD V3 rcp3_cr(V3 d)
{
#if defined(__HIP_DEVICE_COMPILE__) && !(defined(TN_FAST) && TN_FAST) && TN_RCP3_GROUP
    const float lo = __builtin_fminf(__builtin_fminf(__builtin_fabsf(d.x), __builtin_fabsf(d.y)), __builtin_fabsf(d.z));
    const float hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(d.x), __builtin_fabsf(d.y)), __builtin_fabsf(d.z));
    if (__builtin_expect(lo >= 0x1p-126f && hi < 0x1p126f && d.x == d.x && d.y == d.y && d.z == d.z, 1))
        return V3(rcp_refine(d.x), rcp_refine(d.y), rcp_refine(d.z));
    return V3(1.0f/d.x, 1.0f/d.y, 1.0f/d.z);
#else
    return V3(rcpf_cr(d.x), rcpf_cr(d.y), rcpf_cr(d.z));
#endif
}
TN_HD float length_sq(V3 a) { return dot(a, a); }
TN_HD float length(V3 a) { return sqrtf_cr(dot(a, a)); }                                                // maths.h:259
TN_HD V3 normalize(V3 a) { const float r = rsqrtf_cr(dot(a, a)); return V3(a.x*r, a.y*r, a.z*r); }     // a*(1.0/Length(a))                                              // maths.h:260

// SafeNormalize (maths.h:261-273): a * (1.0/sqrt(m))
TN_HD V3 safe_normalize(V3 a, V3 fallback)
{
    float m = length_sq(a);
    if (m > 0.0f)
    {
        float r = rsqrtf_cr(m);
        return V3(a.x*r, a.y*r, a.z*r);
    }
    return fallback;
}

// template Abs (maths.h:67-74): `x < 0.0 ? -x : x` (keeps -0.0, propagates NaN)
TN_HD float absf(float x) { return (x < 0.0f) ? -x : x; }
// Min/Max templates (maths.h:55-59)
TN_HD float minT(float a, float b) { return (a < b) ? a : b; }
TN_HD float maxT(float a, float b) { return (a < b) ? b : a; }
TN_HD int minI(int a, int b) { return (a < b) ? a : b; }
TN_HD int maxI(int a, int b) { return (a < b) ? b : a; }
TN_HD float clampT(float x, float lo, float hi) { return minT(maxT(x, lo), hi); }                    // maths.h:61-65
TN_HD float lerpf(float a, float b, float t) { return a + (b - a)*t; }                               // maths.h:76-80
TN_HD V3 lerp3(V3 a, V3 b, float t) { return a + (b - a)*t; }
TN_HD float sqr(float x) { return x*x; }

// FaceForward (maths.h:1592-1598)
TN_HD V3 face_forward(V3 n, V3 v) { return (dot(v, n) < 0.0f) ? -n : n; }

// ---------------------------------------------------------------------------
// transcendental functions.
//
// The CPU oracle calls glibc's sinf / cosf / expf (and acosf / atan2f for probes).  glibc 2.35's
// sinf, cosf and expf are the "ARM optimized routines": a double-precision evaluation (quadrant
// reduction + degree-7/8 polynomials; 32-entry 2^(k/32) table + cubic) rounded once to fp32.  They are
// NOT correctly rounded (0.56 / 0.502 ulp), so matching them to the last bit takes THEIR algorithm:
// m_sincosf and m_expf below restate it (s_sinf.c, s_cosf.c, sincosf.h, e_expf.c of glibc 2.35; the
// constants are the ones in this image's /lib/x86_64-linux-gnu/libm.so.6: __sincosf_table at .rodata
// +0xb3100, __exp2f_data at +0xb2ca8).  Checked exhaustively on the host against libm itself:
// sinf/cosf identical on all 1,088,421,888 floats in [0, 7]; expf identical on all but 2 of the
// 2,237,399,040 floats with |x| < 87.  Measured effect on the path tracer: bit-identical paths vs the
// reference 98.5 % -> see DESIGN.md section 3.
//   * Outside those domains (never reached by the path: angles are 2*pi*u, exponents are -absorption*t):
//     an fdlibm-kernel / Cody-Waite double evaluation, correctly rounded to ~1e-9.
//   * log (GTR1) is a per-material constant evaluated by the host's own glibc.
//   * acosf / atan2f (probe lookups): glibc 2.35 still uses the fdlibm fp32 routines for these; m_acosf /
//     m_atanf / m_atan2f restate them in the same fp32 operation order (also checked exhaustively).
// TN_LIBM_DOUBLE=0 switches everything to the 1-2 ulp ocml fp32 routines (A/B only).
#ifndef TN_FAST
#define TN_FAST 0           // 1: the tolerance arm (tinsel_fast.hip): hardware transcendentals, see tn_launch.h
#endif
#ifndef TN_LIBM_DOUBLE
#define TN_LIBM_DOUBLE (!TN_FAST)
#endif
__device__ const unsigned long long kExp2fTab[32] = { 0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL, 0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL, 0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL, 0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL, 0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL, 0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL, 0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL, 0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL, 0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL };
#if TN_LIBM_DOUBLE
// A double-precision constant materialised WHERE IT IS USED, in a scalar register pair (two s_mov_b32 the compiler may neither hoist nor
// merge).  Written as plain literals, the 25 coefficients of the two routines below are loop invariants to the compiler: it hoists them out
// of the bounce loop of every kernel that samples a BSDF and keeps them in 24 VGPRs and a dozen SGPRs from the first instruction to the
// last -- in kernels that spill at their register limit.  -DTN_K64_LOCAL=0: plain literals (A/B)
#ifndef TN_K64_LOCAL
#define TN_K64_LOCAL 1
#endif
#if TN_K64_LOCAL
TN_D double k64_here(double c)
{
    unsigned hi = (unsigned)(__builtin_bit_cast(unsigned long long, c) >> 32), lo = (unsigned)__builtin_bit_cast(unsigned long long, c);
    asm volatile("" : "+s"(hi), "+s"(lo));
    return __hiloint2double((int)hi, (int)lo);
}
#define K64(c) k64_here(c)
#else
#define K64(c) (c)
#endif
// The same for an fp32 constant that a VOP3 instruction needs in a register (select operands: gfx950's VOP3 takes no literal): written plainly,
// pi, pi/2 and their negatives (m_atan2f), 1e-3f and FLT_MAX are loop invariants that sat in six VGPRs from the first instruction of k_bounce
// to the last; K32 puts the value in a scalar register at the point of use.
TN_D float k32_here(float c)
{
    unsigned u = __builtin_bit_cast(unsigned, c);
    asm volatile("" : "+s"(u));
    return __uint_as_float(u);
}
#define K32(c) k32_here(c)
TN_D void sincos_wide(double x, float& s, float& c)
{
    const double kd = ::rint(x*K64(0.63661977236758138));            // 2/pi
    const int k = (int)kd;
    double r = ::fma(-kd, K64(1.5707963267948966), x);               // pi/2 (hi)
    r = ::fma(-kd, K64(6.123233995736766e-17), r);                   // pi/2 (lo)
    const double z = r*r;
    // fdlibm __kernel_sin / __kernel_cos
    double ps = ::fma(z, K64(1.58969099521155010221e-10), K64(-2.50507602534068634195e-08));
    ps = ::fma(z, ps, K64(2.75573137070700676789e-06));
    ps = ::fma(z, ps, K64(-1.98412698298579493134e-04));
    ps = ::fma(z, ps, K64(8.33333333332248946124e-03));
    ps = ::fma(z, ps, K64(-1.66666666666666324348e-01));
    const double sr = ::fma(z*r, ps, r);
    double pc = ::fma(z, K64(-1.13596475577881948265e-11), K64(2.08757232129817482790e-09));
    pc = ::fma(z, pc, K64(-2.75573143513906633035e-07));
    pc = ::fma(z, pc, K64(2.48015872894767294178e-05));
    pc = ::fma(z, pc, K64(-1.38888888888741095749e-03));
    pc = ::fma(z, pc, K64(4.16666666666666019037e-02));
    const double cr = ::fma(z*z, pc, ::fma(z, -0.5, 1.0));
    const double sv = (k & 1) ? cr : sr;
    const double cv = (k & 1) ? sr : cr;
    s = (float)((k & 2) ? -sv : sv);
    c = (float)(((k + 1) & 2) ? -cv : cv);
}

// glibc 2.35 sinf + cosf for 0 <= y <= 7 (reduce_fast + sinf_poly of sincosf.h), both results at once
TN_D void m_sincosf(float y, float& s, float& c)
{
    const double x = (double)y;
    if (!(y >= 0.0f && y <= 7.0f))
    {
        sincos_wide(x, s, c);
        return;
    }
    if (y < 0x1p-12f)                                           // abstop12(y) < abstop12(0x1p-12f): sinf = y, cosf = 1
    {
        s = y;
        c = 1.0f;
        return;
    }
    int n = 0;
    double xr = x;
    if (y >= 0x1.921FB6p-1f)                                    // |y| >= pi/4: reduce_fast
    {
        const double r = x*K64(0x1.45F306DC9C883p+23);               // hpi_inv = 2/pi * 2^24
        n = ((int)r + 0x800000) >> 24;
        xr = ::fma(-(double)n, K64(0x1.921FB54442D18p0), x);
    }
    const double x2 = xr*xr;
    // sinf_poly's two branches (table entries 0/1 differ only in the sign of the cosine coefficients)
    const double x3 = xr*x2;
    const double s1 = ::fma(x2, K64(-0x1.994eb3774cf24p-13), K64(0x1.1107605230bc4p-7));
    const double x5 = x3*x2;
    const double sp = ::fma(x5, s1, ::fma(x3, K64(-0x1.555545995a603p-3), xr));                 // sin(xr)
    const double x4 = x2*x2;
    const double c2 = ::fma(x2, K64(0x1.99343027bf8c3p-16), K64(-0x1.6c087e89a359dp-10));
    const double c1 = ::fma(x2, K64(-0x1.ffffffd0c621cp-2), 1.0);
    const double x6 = x4*x2;
    const double cp = ::fma(x6, c2, ::fma(x4, K64(0x1.55553e1068f19p-5), c1));                  // cos(xr)
    // quadrant n: sin y = {sp, cp, -sp, -cp}[n&3], cos y = {cp, -sp, -cp, sp}[n&3]  (the sign[] / table[1]
    // bookkeeping of s_sinf.c / s_cosf.c: negating x or the polynomial is exact, so signs commute)
    const double sv = (n & 1) ? cp : sp;
    const double cv = (n & 1) ? sp : cp;
    s = (float)((n & 2) ? -sv : sv);
    c = (float)(((n + 1) & 2) ? -cv : cv);
}
TN_D float m_sinf(float x) { float s, c; m_sincosf(x, s, c); return s; }
TN_D float m_cosf(float x) { float s, c; m_sincosf(x, s, c); return c; }


// glibc 2.35 expf, the __expf_fma ifunc variant an FMA-capable x86-64 host runs (e_expf.c, non-TOINT path: the
// SHIFT trick; the compiler contracted InvLn2N*x + SHIFT and InvLn2N*x - kd into FMAs there, so they are FMAs here)
// `tab`: kExp2fTab or a copy of it (kernels that call this in a tight loop keep one in LDS)
// LOCAL: the five double constants materialised where they are used (K64 above) -- the path kernels' Beer-Lambert term; the accumulate
// kernel, which calls this six times per staged path and has registers to spare, keeps the literals
template <class Tab, bool LOCAL = false>
TN_D float m_expf_tab(float xf, const Tab& tab)
{
#define KE(c) (LOCAL ? K64(c) : (c))
    const uint32_t ix = __float_as_uint(xf);
    const uint32_t abstop = (ix >> 20) & 0x7ffu;
    if (abstop > 0x42au)                                        // |x| >= 88 or NaN
    {
        if (ix == 0xff800000u)
            return 0.0f;
        if (abstop > 0x7f7u)
            return xf + xf;
        if (xf > 0x1.62e42ep6f)
            return __uint_as_float(0x7f800000u);                // overflow
        if (xf < -0x1.9fe368p6f)
            return 0.0f;                                        // underflow
        if (xf < -0x1.9d1d9ep6f)
            return __uint_as_float(1u);                         // 0x1.4p-75f*0x1.4p-75f: the smallest subnormal
    }
    const double x = (double)xf;
    double kd = ::fma(KE(0x1.71547652b82fep+5), x, KE(0x1.8p+52));       // InvLn2N (N = 32), SHIFT
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd -= KE(0x1.8p+52);
    const double r = ::fma(KE(0x1.71547652b82fep+5), x, -kd);
    unsigned long long t = tab[ki & 31u];
    t += ki << (52 - 5);
    const double sc = __longlong_as_double((long long)t);
    const double z = ::fma(KE(0x1.c6af84b912394p-20), r, KE(0x1.ebfce50fac4f3p-13));
    const double r2 = r*r;
    double yv = ::fma(KE(0x1.62e42ff0c52d6p-6), r, 1.0);
    yv = ::fma(z, r2, yv);
    return (float)(yv*sc);
#undef KE
}
TN_D float m_expf(float xf) { return m_expf_tab<decltype(kExp2fTab), true>(xf, kExp2fTab); }
TN_D float m_logf(float x) { return (float)::log((double)x); }

// glibc 2.35 __ieee754_acosf (sysdeps/ieee754/flt-32/e_acosf.c, the fdlibm fp32 routine; plain fp32 ops, no
// FMA).  Checked on the host against libm: identical on all 2 x 1,065,353,217 floats in [-1, 1].
TN_D float m_acosf(float x)
{
    const float one = 1.0f, pi = __uint_as_float(0x40490fdau), pio2_hi = __uint_as_float(0x3fc90fdau), pio2_lo = __uint_as_float(0x33a22168u);
    const float pS0 = __uint_as_float(0x3e2aaaabu), pS1 = __uint_as_float(0xbea6b090u), pS2 = __uint_as_float(0x3e4e0aa8u),
                pS3 = __uint_as_float(0xbd241146u), pS4 = __uint_as_float(0x3a4f7f04u), pS5 = __uint_as_float(0x3811ef08u);
    const float qS1 = __uint_as_float(0xc019d139u), qS2 = __uint_as_float(0x4001572du), qS3 = __uint_as_float(0xbf303361u), qS4 = __uint_as_float(0x3d9dc62eu);
    const int hx = __float_as_int(x);
    const int ix = hx & 0x7fffffff;
    if (ix == 0x3f800000)
        return (hx > 0) ? 0.0f : pi + 2.0f*pio2_lo;
    if (ix > 0x3f800000)
        return (x - x)/(x - x);
    if (ix < 0x3f000000)
    {
        if (ix <= 0x32800000)
            return pio2_hi + pio2_lo;
        const float z = x*x;
        const float p = z*(pS0 + z*(pS1 + z*(pS2 + z*(pS3 + z*(pS4 + z*pS5)))));
        const float q = one + z*(qS1 + z*(qS2 + z*(qS3 + z*qS4)));
        const float r = p/q;
        return pio2_hi - (x - (pio2_lo - x*r));
    }
    if (hx < 0)
    {
        const float z = (one + x)*0.5f;
        const float p = z*(pS0 + z*(pS1 + z*(pS2 + z*(pS3 + z*(pS4 + z*pS5)))));
        const float q = one + z*(qS1 + z*(qS2 + z*(qS3 + z*qS4)));
        const float sq = sqrtf_cr(z);
        const float r = p/q;
        const float w = r*sq - pio2_lo;
        return pi - 2.0f*(sq + w);
    }
    const float z = (one - x)*0.5f;
    const float sq = sqrtf_cr(z);
    const float df = __uint_as_float(__float_as_uint(sq) & 0xfffff000u);
    const float c = (z - df*df)/(sq + df);
    const float p = z*(pS0 + z*(pS1 + z*(pS2 + z*(pS3 + z*(pS4 + z*pS5)))));
    const float q = one + z*(qS1 + z*(qS2 + z*(qS3 + z*qS4)));
    const float r = p/q;
    const float w = r*sq + c;
    return 2.0f*(df + w);
}

// glibc 2.35 __atanf (s_atanf.c) and __ieee754_atan2f (e_atan2f.c): fdlibm fp32, plain ops (the constants
// are this image's libm.so.6 .rodata +0x9ebd0.. ; aT[0] is 0x3eaaaaab there).  Checked on the host:
// atan2f identical on 320,000,000 random (y, x) pairs, atanf on a 68 M-point sweep.
TN_D float m_atanf(float x)
{
    const float one = 1.0f;
    const int hx = __float_as_int(x);
    const int ix = hx & 0x7fffffff;
    float hi = 0.0f, lo = 0.0f;
    int id;
    if (ix >= 0x4c000000)
    {
        if (ix > 0x7f800000)
            return x + x;
        const float r = __uint_as_float(0x3fc90fdau) + __uint_as_float(0x33a22168u);
        return (hx > 0) ? r : -r;
    }
    if (ix < 0x3ee00000)
    {
        if (ix < 0x31000000)
            return x;
        id = -1;
    }
    else
    {
        x = fabsf(x);
        if (ix < 0x3f980000)
        {
            if (ix < 0x3f300000) { id = 0; hi = __uint_as_float(0x3eed6338u); lo = __uint_as_float(0x31ac3769u); x = (2.0f*x - one)/(2.0f + x); }
            else                 { id = 1; hi = __uint_as_float(0x3f490fdau); lo = __uint_as_float(0x33222168u); x = (x - one)/(x + one); }
        }
        else
        {
            if (ix < 0x401c0000) { id = 2; hi = __uint_as_float(0x3f7b985eu); lo = __uint_as_float(0x33140fb4u); x = (x - 1.5f)/(one + 1.5f*x); }
            else                 { id = 3; hi = __uint_as_float(0x3fc90fdau); lo = __uint_as_float(0x33a22168u); x = -rcpf_cr(x); }
        }
    }
    const float z = x*x;
    const float w = z*z;
    const float s1 = z*(__uint_as_float(0x3eaaaaabu) + w*(__uint_as_float(0x3e124925u) + w*(__uint_as_float(0x3dba2e6eu) +
                     w*(__uint_as_float(0x3d886b35u) + w*(__uint_as_float(0x3d4bda59u) + w*__uint_as_float(0x3c8569d7u))))));
    const float s2 = w*(__uint_as_float(0xbe4ccccdu) + w*(__uint_as_float(0xbde38e38u) + w*(__uint_as_float(0xbd9d8795u) +
                     w*(__uint_as_float(0xbd6ef16bu) + w*__uint_as_float(0xbd15a221u)))));
    if (id < 0)
        return x - x*(s1 + s2);
    const float r = hi - ((x*(s1 + s2) - lo) - x);
    return (hx < 0) ? -r : r;
}

TN_D float m_atan2f(float y, float x)
{
    const float tiny = 1.0e-30f, pi_lo = __uint_as_float(0xb3bbbd2eu);
#define pi_o_2 K32(__uint_as_float(0x3fc90fdbu))
#define pi K32(__uint_as_float(0x40490fdbu))
    const int hx = __float_as_int(x), hy = __float_as_int(y);
    const int ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000)
        return x + y;
    if (hx == 0x3f800000 && iy != 0x7f800000)
        return m_atanf(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    // infinities (e_atan2f.c's own cases; never reached from unit directions.  They used to go through the double-precision atan2: twenty
    // coefficients that the compiler hoisted out of every loop around a call site and parked in 152 B of scratch)
    if (ix == 0x7f800000)
    {
        const float pi_o_4 = __uint_as_float(0x3f490fdbu);
        if (iy == 0x7f800000)
            return (m == 0) ? pi_o_4 + tiny : (m == 1) ? -pi_o_4 - tiny : (m == 2) ? 3.0f*pi_o_4 + tiny : -3.0f*pi_o_4 - tiny;
        return (m == 0) ? 0.0f : (m == 1) ? -0.0f : (m == 2) ? pi + tiny : -pi - tiny;
    }
    if (iy == 0x7f800000)
        return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (iy == 0)
        return (m < 2) ? y : ((m == 2) ? pi + tiny : -pi - tiny);
    if (ix == 0)
        return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60)
        z = pi_o_2 + 0.5f*pi_lo;
    else if (hx < 0 && k < -60)
        z = 0.0f;
    else
        z = m_atanf(fabsf(y/x));
    if (m == 0) return z;
    if (m == 1) return -z;
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
#undef pi_o_2
#undef pi
}
#elif TN_FAST
// tolerance arm.  TN_FAST_NATIVE_TRIG=1: v_sin_f32 / v_cos_f32 / v_exp_f32 (absolute error ~1e-6: visibly more paths leave
// the exact arm's track in specular scenes); 0 (default): ocml's fp32 routines (1-2 ulp)
#ifndef TN_FAST_NATIVE_TRIG
#define TN_FAST_NATIVE_TRIG 0
#endif
#if TN_FAST_NATIVE_TRIG
TN_D void m_sincosf(float x, float& s, float& c) { s = __sinf(x); c = __cosf(x); }
TN_D float m_sinf(float x) { return __sinf(x); }
TN_D float m_cosf(float x) { return __cosf(x); }
TN_D float m_expf(float x) { return __expf(x); }
template <class Tab> TN_D float m_expf_tab(float x, const Tab&) { return __expf(x); }
#else
TN_D void m_sincosf(float x, float& s, float& c) { ::sincosf(x, &s, &c); }
TN_D float m_sinf(float x) { return ::sinf(x); }
TN_D float m_cosf(float x) { return ::cosf(x); }
TN_D float m_expf(float x) { return ::expf(x); }
template <class Tab> TN_D float m_expf_tab(float x, const Tab&) { return ::expf(x); }
#endif
TN_D float m_logf(float x) { return __logf(x); }
TN_D float m_acosf(float x) { return ::acosf(x); }
TN_D float m_atan2f(float y, float x) { return ::atan2f(y, x); }
#else
TN_D void m_sincosf(float x, float& s, float& c) { s = ::sinf(x); c = ::cosf(x); }
TN_D float m_sinf(float x) { return ::sinf(x); }
TN_D float m_cosf(float x) { return ::cosf(x); }
TN_D float m_expf(float x) { return ::expf(x); }
template <class Tab> TN_D float m_expf_tab(float x, const Tab&) { return ::expf(x); }
TN_D float m_logf(float x) { return ::logf(x); }
TN_D float m_acosf(float x) { return ::acosf(x); }
TN_D float m_atan2f(float y, float x) { return ::atan2f(y, x); }
#endif

// ---------------------------------------------------------------------------
// quaternions / transforms

// operator*(Quat, Quat)  (maths.h:531-537)
TN_HD Q4 qmul(Q4 a, Q4 b)
{
    Q4 r;
    r.x = a.w*b.x + b.w*a.x + a.y*b.z - b.y*a.z;
    r.y = a.w*b.y + b.w*a.y + a.z*b.x - b.z*a.x;
    r.z = a.w*b.z + b.w*a.z + a.x*b.y - b.x*a.y;
    r.w = a.w*b.w - a.x*b.x - a.y*b.y - a.z*b.z;
    return r;
}

TN_HD Q4 qconj(Q4 q) { Q4 r = { -q.x, -q.y, -q.z, q.w }; return r; }                                 // maths.h:555

// Rotate(q, v) = (q*Quat(v,0)*Conjugate(q)).xyz  (maths.h:558-563)
TN_HD V3 qrotate(Q4 q, V3 v)
{
    Q4 qv = { v.x, v.y, v.z, 0.0f };
    Q4 t = qmul(qmul(q, qv), qconj(q));
    return V3(t.x, t.y, t.z);
}

// Normalize(Quat)  (maths.h:547-553)
TN_HD Q4 qnormalize(Q4 q)
{
    float r = rsqrtf_cr(q.x*q.x + q.y*q.y + q.z*q.z + q.w*q.w);      // 1.0/Length(q)
    Q4 o = { q.x*r, q.y*r, q.z*r, q.w*r };
    return o;
}

// InterpolateTransform(Transform, Transform, t)  (maths.h:1566-1569)
TN_HD Xform interpolate_xform(const Xform& a, const Xform& b, float t)
{
    Xform o;
    o.p = lerp3(a.p, b.p, t);
    Q4 q = { a.r.x + (b.r.x - a.r.x)*t, a.r.y + (b.r.y - a.r.y)*t, a.r.z + (b.r.z - a.r.z)*t, a.r.w + (b.r.w - a.r.w)*t };
    o.r = qnormalize(q);
    o.s = lerpf(a.s, b.s, t);
    return o;
}

TN_HD V3 xform_vector(const Xform& t, V3 v) { return qrotate(t.r, t.s*v); }                          // maths.h:601-604
TN_HD V3 xform_point(const Xform& t, V3 v) { return t.p + qrotate(t.r, t.s*v); }                     // maths.h:606-609
TN_HD V3 inv_xform_vector(const Xform& t, V3 v) { return rcpf_cr(t.s)*qrotate(qconj(t.r), v); }        // maths.h:611-614
TN_HD V3 inv_xform_point(const Xform& t, V3 v) { return rcpf_cr(t.s)*qrotate(qconj(t.r), v - t.p); }   // maths.h:616-619

// ---------------------------------------------------------------------------
// Random (maths.h:1036-1091): two-word xorshift/multiply generator

struct Rng
{
    uint32_t s1, s2;

    // Random(int seed), with the addition done in uint32 (no signed overflow)
    TN_HD static Rng seeded(uint32_t seed)
    {
        Rng r;
        r.s1 = 315645664u + seed;
        r.s2 = r.s1 ^ 0x13ab45feu;
        return r;
    }

    TN_HD uint32_t rand()
    {
        s1 = (s2 ^ ((s1 << 5) | (s1 >> 27))) ^ (s1*s2);
        s2 = s1 ^ ((s2 << 12) | (s2 >> 20));
        return s1;
    }

    // Randf(): (float)value * (1.0f/(float)0xffffffff)   -- the constant is exactly 2^-32
    TN_HD float randf() { return (float)rand()*(1.0f/4294967296.0f); }
};

// passSeed[s] = (s+1)-th output of Random(1).Rand()  (render.cu:1050-1052, 1099)
inline uint32_t pass_seed(uint32_t passIndex)
{
    Rng r = Rng::seeded(1u);
    uint32_t v = 0;
    for (uint32_t i = 0; i <= passIndex; ++i)
        v = r.rand();
    return v;
}

// ---------------------------------------------------------------------------
// sampling helpers

// BasisFromVector (maths.h:1261-1275)
TN_HD void basis_from_vector(V3 w, V3& u, V3& v)
{
    if (fabsf(w.x) > fabsf(w.y))
    {
        float invLen = rsqrtf_cr(w.x*w.x + w.z*w.z);
        u = V3(-w.z*invLen, 0.0f, w.x*invLen);
    }
    else
    {
        float invLen = rsqrtf_cr(w.y*w.y + w.z*w.z);
        u = V3(0.0f, w.z*invLen, -w.y*invLen);
    }
    v = cross(w, u);
}

// UniformSampleSphere (maths.h:1278-1287)
TN_D V3 uniform_sample_sphere(float u1, float u2)
{
    float z = 1.f - 2.f*u1;
    float r = sqrtf_cr(maxT(0.f, 1.f - z*z));
    float phi = 2.f*kPi*u2;
    float sn, cs;
    m_sincosf(phi, sn, cs);
    float x = r*cs;
    float y = r*sn;
    return V3(x, y, z);
}

// UniformSampleHemisphere(Random&) (maths.h:1291-1302)
TN_D V3 uniform_sample_hemisphere(Rng& rng)
{
    float z = rng.randf();
    float w = sqrtf_cr(1.0f - z*z);
    float phi = k2Pi*rng.randf();
    float sn, cs;
    m_sincosf(phi, sn, cs);
    float x = cs*w;
    float y = sn*w;
    return V3(x, y, z);
}

// CosineSampleHemisphere (maths.h:1319-1325) via UniformSampleDisc (maths.h:1304-1310)
TN_D V3 cosine_sample_hemisphere(float u1, float u2)
{
    float r = sqrtf_cr(u1);
    float theta = k2Pi*u2;
    float sn, cs;
    m_sincosf(theta, sn, cs);
    float sx = r*cs;
    float sy = r*sn;
    float z = sqrtf_cr(maxT(0.0f, 1.0f - sx*sx - sy*sy));
    return V3(sx, sy, z);
}

// UniformSampleTriangle (maths.h:1312-1317)
TN_D void uniform_sample_triangle(Rng& rng, float& u, float& v)
{
    float r = sqrtf_cr(rng.randf());
    u = 1.0f - r;
    v = rng.randf()*r;
}

// ClampLength (maths.h:1577-1589)
TN_HD V3 clamp_length(V3 v, float maxLength)
{
    float l = length(v);
    if (l > maxLength)
        return v*(maxLength/l);
    return v;
}

} // namespace tn
