// tn_selftest.h -- exhaustive check of the short reciprocal / square-root sequences of tn_math.h (rcp_candidate, sqrt_candidate)
// against the compiler's IEEE expansions, on every one of the 2^32 fp32 bit patterns (tinsel_hip_selftest_arith,
// include/tinsel_hip.h).  The parity arm may only be built with a variant that comes back with zero mismatches.
// Not part of the render path.
#pragma once

#include "tn_math.h"

namespace tn {

// counts[0] mismatches, [1] of them with a denormal operand, [2] with |x| >= 2^126 (rcp) / x < 0 (sqrt), [3] any other;
// counts[4 + e]: mismatches by the operand's exponent field e (0..255); firstBad: the smallest mismatching bit pattern
template <int OP, int V>
__global__ __launch_bounds__(256) void k_selftest_arith(unsigned long long* __restrict__ counts, uint32_t* __restrict__ firstBad)
{
#if defined(__HIP_DEVICE_COMPILE__)     // (the candidates are device-only builtins)
    const uint32_t base = (blockIdx.x*blockDim.x + threadIdx.x);
    uint32_t bad = 0, badDen = 0, badBig = 0, first = 0xffffffffu;
    for (uint32_t k = 0; k < 256u; ++k)
    {
        const uint32_t bits = k*(1u << 24) + base;     // grid = 2^24 threads
        const float x = __uint_as_float(bits);
        float want, got;
        if constexpr (OP == 0)      { want = 1.0f/x; got = rcp_candidate<V>(x); }
        else if constexpr (OP == 1) { want = sqrtf(x); got = sqrt_candidate<V>(x); }
        else                        { want = 1.0f/sqrtf(x); got = rsqrt_candidate<V>(x); }
        const uint32_t wb = __float_as_uint(want), gb = __float_as_uint(got);
        const bool same = wb == gb || (want != want && got != got);     // any NaN equals any NaN (payloads are not consumed anywhere)
        if (!same)
        {
            ++bad;
            const uint32_t mag = bits & 0x7fffffffu;
            if (mag < 0x00800000u) ++badDen;
            else if (OP == 0 ? mag >= 0x7e800000u : (bits >> 31) != 0u) ++badBig;
            first = bits < first ? bits : first;
            atomicAdd(&counts[4 + ((bits >> 23) & 255u)], 1ull);
        }
    }
    if (bad)
    {
        atomicAdd(&counts[0], (unsigned long long)bad);
        atomicAdd(&counts[1], (unsigned long long)badDen);
        atomicAdd(&counts[2], (unsigned long long)badBig);
        atomicAdd(&counts[3], (unsigned long long)(bad - badDen - badBig));
        atomicMin(firstBad, first);
    }
#endif
}

template <int OP, int V>
static void launch_selftest_arith(unsigned long long* counts, uint32_t* firstBad)
{
    hipLaunchKernelGGL((k_selftest_arith<OP, V>), dim3(1u << 16), dim3(256), 0, nullptr, counts, firstBad);
}

} // namespace tn
