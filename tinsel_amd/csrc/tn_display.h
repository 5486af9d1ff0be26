// tn_display.h -- the display stage that follows the accumulator in the reference's frame loop
// (main.cpp:258-282): normalise by the filter weight, filmic tone map, display gamma, and the optional
// non-local-means filter.  Per-pixel streaming kernels: HBM-bound (16 B in, 16 B out per pixel).
//
//   k_present     g_filtered[i] = LinearToSrgb(ToneMap(g_pixels[i]*(exposure/w), limit))     main.cpp:262-271
//   k_nlm_means   AverageFilter (nlm.cpp:4-33)
//   k_nlm         NonLocalMeansFilter (nlm.cpp:35-77)
//
// Same operation order as the host code, powf/expf as the host libm evaluates them (tn_powf.h, tn_math.h),
// so the displayed float image -- and hence the 8-bit file written from it -- is bit-identical.
#pragma once

#include "tn_math.h"
#include "tn_powf.h"

namespace tn {

// Max<T>(a,b) = (a < b) ? b : a  (maths.h:58-59)
TN_D float max_ref(float a, float b) { return (a < b) ? b : a; }

// ToneMap (util.h:25-42): the filmic curve; SrgbToLinear (maths.h:1551-1555) is applied by the caller
TN_D float filmic_channel(float c)
{
    const float x = max_ref(0.0f, c - 0.004f);
    const float num = x*(6.2f*x + 0.5f);
    const float den = x*(6.2f*x + 1.7f) + 0.06f;       // Vec3(0.06): the double literal narrows to float in the ctor
    return num/den;
}
TN_D float tonemap_channel(float c) { return m_powf(filmic_channel(c), 2.2f); }

__global__ __launch_bounds__(256) void k_present(const float4* __restrict__ accum, float4* __restrict__ out, int n,
                                                 float exposure, float limit)
{
    // powf's two tables (768 B) in LDS: six powf per pixel, each with two data-dependent table reads
    __shared__ double s_log2[16][2];
    __shared__ unsigned long long s_exp2[32];
    if (threadIdx.x < 32)
    {
        s_log2[threadIdx.x >> 1][threadIdx.x & 1] = kPowfLog2Tab[threadIdx.x >> 1][threadIdx.x & 1];
        s_exp2[threadIdx.x] = kPowfExp2Tab[threadIdx.x];
    }
    __syncthreads();

    const int i = blockIdx.x*256 + threadIdx.x;
    if (i >= n)
        return;
    const float4 p = accum[i];
    const float s = exposure/p.w;
    const float kInvGamma = 1.0f/2.2f;
    float4 r;
    // Color*s scales w too; ToneMap returns SrgbToLinear(Color(rgb, 0)): w = powf(0, 2.2) = 0; LinearToSrgb keeps w
    r.x = m_powf_tab(m_powf_tab(filmic_channel(p.x*s), 2.2f, s_log2, s_exp2), kInvGamma, s_log2, s_exp2);
    r.y = m_powf_tab(m_powf_tab(filmic_channel(p.y*s), 2.2f, s_log2, s_exp2), kInvGamma, s_log2, s_exp2);
    r.z = m_powf_tab(m_powf_tab(filmic_channel(p.z*s), 2.2f, s_log2, s_exp2), kInvGamma, s_log2, s_exp2);
    r.w = 0.0f;
    (void)limit;                                        // only the commented-out Reinhard operator used it
    out[i] = r;
}

// AverageFilter: box mean over the clipped (2r+1)^2 window, summed column by column (fx outer, fy inner)
__global__ __launch_bounds__(256) void k_nlm_means(const float4* __restrict__ in, float4* __restrict__ means, int width, int height, int radius)
{
    const int x = blockIdx.x*16 + (threadIdx.x & 15);
    const int y = blockIdx.y*16 + (threadIdx.x >> 4);
    if (x >= width || y >= height)
        return;
    const int xlower = maxI(0, x - radius), xupper = minI(width - 1, x + radius);
    const int ylower = maxI(0, y - radius), yupper = minI(height - 1, y + radius);

    int count = 0;
    float4 sum = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int fx = xlower; fx <= xupper; ++fx)
    {
        for (int fy = ylower; fy <= yupper; ++fy)
        {
            const float4 c = in[fy*width + fx];
            sum.x += c.x; sum.y += c.y; sum.z += c.z; sum.w += c.w;
            count += 1;
        }
    }
    const float rc = 1.0f/count;
    means[y*width + x] = make_float4(sum.x*rc, sum.y*rc, sum.z*rc, sum.w*rc);
}

__global__ __launch_bounds__(256) void k_nlm(const float4* __restrict__ in, const float4* __restrict__ means, float4* __restrict__ out,
                                             int width, int height, float falloff, int radius)
{
    __shared__ unsigned long long s_exp[32];            // expf's table: (2r+1)^2 data-dependent reads per pixel
    if (threadIdx.x < 32)
        s_exp[threadIdx.x] = kExp2fTab[threadIdx.x];
    __syncthreads();

    const int x = blockIdx.x*16 + (threadIdx.x & 15);
    const int y = blockIdx.y*16 + (threadIdx.x >> 4);
    if (x >= width || y >= height)
        return;
    const int xlower = maxI(0, x - radius), xupper = minI(width - 1, x + radius);
    const int ylower = maxI(0, y - radius), yupper = minI(height - 1, y + radius);

    float totalWeight = 0.0f;
    float4 sum = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const float4 mean = means[y*width + x];

    for (int fx = xlower; fx <= xupper; ++fx)
    {
        for (int fy = ylower; fy <= yupper; ++fy)
        {
            const float4 m = means[fy*width + fx];
            const float dx = mean.x - m.x, dy = mean.y - m.y, dz = mean.z - m.z, dw = mean.w - m.w;
            const float lsq = dx*dx + dy*dy + dz*dz + dw*dw;       // LengthSq(Vec4) (maths.h:331-332)
            const float weight = m_expf_tab(-falloff*lsq, s_exp);
            const float4 c = in[fy*width + fx];
            sum.x += c.x*weight; sum.y += c.y*weight; sum.z += c.z*weight; sum.w += c.w*weight;
            totalWeight += weight;
        }
    }
    const float rc = 1.0f/totalWeight;
    out[y*width + x] = make_float4(sum.x*rc, sum.y*rc, sum.z*rc, sum.w*rc);
}

} // namespace tn
