// tn_ubench.h -- two yard-sticks measured on the GPU the renderer runs on (tinsel_hip_ubench, include/tinsel_hip.h).
//
// bench.py quotes the path kernels against them and calibrates the rocprofv3 byte counters on them, IN THE SAME RUN:
//   k_ub_copy     a float4 stream copy: the HBM rate this chip sustains (the guide: 8 TB/s spec, ~6.3 achieved) and, under
//                 --pmc, how many bytes one count of FETCH_SIZE / WRITE_SIZE stands for on wide coalesced streams;
//   k_ub_gather   dependent chases through a table of 64-B records, one chain per lane, four dwordx4 per visit -- the access
//                 pattern of a BVH walk (Node64, tn_walk.h).  With a table far beyond the 256 MiB Infinity Cache every visit is
//                 a miss all the way: FETCH_SIZE per visit calibrates the counter for random 64-B gathers; with a table the
//                 size of a walked tree, or one that fits an XCD's L2, the visit rate is the ceiling k_walk is quoted against.
// Not part of the render path; nothing here touches a renderer's state.
#pragma once

#include "tn_math.h"

namespace tn {

// eight 16-B loads in flight per lane, then eight stores; consecutive lanes on consecutive float4s (the guide's streaming shape);
// NT: non-temporal loads and stores (a stream that nobody re-reads need not displace what the caches hold)
typedef float UbCopyF4 __attribute__((ext_vector_type(4)));
template <bool NT, bool CONTIG>
__global__ __launch_bounds__(256) void k_ub_copy(const float4* __restrict__ inV, float4* __restrict__ outV, size_t n)
{
    const UbCopyF4* in = reinterpret_cast<const UbCopyF4*>(inV);
    UbCopyF4* out = reinterpret_cast<UbCopyF4*>(outV);
    constexpr int kUnroll = 8;
    // CONTIG: a workgroup's eight loads are eight consecutive 4-KB rows (32 KB contiguous per workgroup and step); else the whole
    // grid sweeps one row per load
    const size_t stride = CONTIG ? (size_t)blockDim.x : (size_t)gridDim.x*blockDim.x;
    const size_t jump = CONTIG ? (size_t)gridDim.x*blockDim.x*kUnroll : stride*kUnroll;
    size_t i = CONTIG ? (size_t)blockIdx.x*blockDim.x*kUnroll + threadIdx.x : (size_t)blockIdx.x*blockDim.x + threadIdx.x;
    for (; i + (kUnroll - 1)*stride < n; i += jump)
    {
        UbCopyF4 v[kUnroll];
#pragma unroll
        for (int k = 0; k < kUnroll; ++k)
            v[k] = NT ? __builtin_nontemporal_load(in + i + (size_t)k*stride) : in[i + (size_t)k*stride];
#pragma unroll
        for (int k = 0; k < kUnroll; ++k)
        {
            if (NT) __builtin_nontemporal_store(v[k], out + i + (size_t)k*stride);
            else out[i + (size_t)k*stride] = v[k];
        }
    }
    if (!CONTIG)
        for (; i < n; i += stride)
            out[i] = in[i];     // (the CONTIG shapes are launched on sizes that divide evenly)
}

// record i: 16 floats, the link to the next record of its chain in word 12.  next(i) = (a*i + c) mod nrec with nrec a power
// of two, a = 1 (mod 4), c odd: one cycle through all records (Hull-Dobell), consecutive visits 64-B-random
__global__ __launch_bounds__(256) void k_ub_fill(float4* __restrict__ recs, uint32_t nrec)
{
    for (uint32_t i = blockIdx.x*blockDim.x + threadIdx.x; i < nrec; i += gridDim.x*blockDim.x)
    {
        const uint32_t nxt = (i*2891336453u + 1442695041u) & (nrec - 1u);
        const float f = 1e-3f*(float)(i & 1023u);
        recs[(size_t)i*4 + 0] = make_float4(f, f, f, f);
        recs[(size_t)i*4 + 1] = make_float4(f, f, f, f);
        recs[(size_t)i*4 + 2] = make_float4(f, f, f, f);
        recs[(size_t)i*4 + 3] = make_float4(__uint_as_float(nxt), f, f, f);
    }
}

typedef float UbF4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) UbF4* UbGlobalF4;

// SET names the table for the profiler's kernel list: 0 beyond the Infinity Cache, 1 the size of a walked tree, 2 inside one L2
template <int SET>
__global__ __launch_bounds__(256, 4) void k_ub_gather(const float4* __restrict__ recsIn, uint32_t nrec, int steps, float* __restrict__ out)
{
    UbGlobalF4 recs = (UbGlobalF4)(uintptr_t)recsIn;
    const uint32_t tid = blockIdx.x*blockDim.x + threadIdx.x;
    uint32_t idx = (tid*2654435761u + 40503u) & (nrec - 1u);
    float acc = 0.0f;
    for (int s = 0; s < steps; ++s)
    {
        UbGlobalF4 p = recs + (size_t)idx*4;
        const UbF4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z + d.w;
        idx = __float_as_uint(d.x);
    }
    out[tid] = acc + (float)idx;
}

} // namespace tn
