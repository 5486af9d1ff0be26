// tn_split.h -- the SPLIT wavefront pipeline (meshes or a scene BVH in HBM: ajax, glass, many_spheres, motionblur, table, transmission):
// k_generate, then per bounce k_extend (+ light sampling) / k_lights / k_shadow / k_shade (k_shade_sorted), the region-order and work-list
// kernels (k_region_order, k_seg_prefix, k_seg_expand).  k_walk is tn_walk.h, k_swalk tn_swalk.h.
#pragma once

#include "tn_path_state.h"

namespace tn {

// ---------------------------------------------------------------------------
// Longest regions first.  A workgroup takes four consecutive regions, the dispatcher hands workgroups to CUs in index order,
// and paths die in patches of the image (sky): launched in image order, a bounce ends with a few waves still working through
// full regions while the rest of the chip idles.  k_region_order sorts the workgroups' region groups by the power of two of
// their live entries, largest first (one workgroup, LDS histogram; the order inside a class is whatever the atomics made
// it: order never changes a result), and the kernels that read `order` take group order[blockIdx.x].
constexpr int kOrderBlock = 1024;
constexpr int kOrderClasses = 33;

__global__ __launch_bounds__(kOrderBlock) void k_region_order(const uint32_t* __restrict__ front, const uint32_t* __restrict__ back, uint32_t numRegions,
                                                              uint32_t* __restrict__ order)
{
    __shared__ uint32_t s_count[kOrderClasses], s_start[kOrderClasses];
    const uint32_t groups = numRegions/kRegionsPerBlock;
    if (threadIdx.x < kOrderClasses)
        s_count[threadIdx.x] = 0;
    __syncthreads();
    auto cls = [&](uint32_t g) -> uint32_t {
        uint32_t n = 0;
        for (uint32_t k = 0; k < kRegionsPerBlock; ++k)
            n += front[g*kRegionsPerBlock + k] + back[g*kRegionsPerBlock + k];
        return n ? 32u - (uint32_t)__clz((int)n) : 0u;         // 0: empty, else 1 + floor(log2 n)
    };
    for (uint32_t g = threadIdx.x; g < groups; g += kOrderBlock)
        atomicAdd(&s_count[cls(g)], 1u);
    __syncthreads();
    if (threadIdx.x == 0)
    {
        uint32_t run = 0;
        for (int c = kOrderClasses - 1; c >= 0; --c)
        {
            s_start[c] = run;
            run += s_count[c];
        }
    }
    __syncthreads();
    for (uint32_t g = threadIdx.x; g < groups; g += kOrderBlock)
        order[atomicAdd(&s_start[cls(g)], 1u)] = g;
}


// ===========================================================================
// The SPLIT variant of the pipeline (TINSEL_PIPELINE_WAVEFRONT_SPLIT): the same bounce cut into k_extend / k_lights /
// k_shadow / k_shade with hit and shadow-ray records parked in HBM in between: lean trace kernels (and k_walk ahead of
// them) for scenes with meshes in HBM or many shadow rays per bounce.

// ---------------------------------------------------------------------------
// Queues sorted by "enters a big mesh" (split pipeline, scenes with a mesh in HBM).  k_shade, which produces the next
// bounce's extension queue and this bounce's shadow queue, fills each from both ends: in front the rays whose leaf-box
// test against one of the LARGE meshes succeeds, at the back all others.  trace() is unchanged and results do not
// depend on queue order; what changes is that a wave of k_extend / k_shadow is either full of rays that walk the big
// mesh's BVH or has none (measured on the 524k-triangle config: 60 % of the rays enter the mesh, and unsorted,
// practically every wave paid for the walk with 23 % of its lanes active).

struct BinPrims
{
    int count;
    int prim[7];
};

TN_D bool ray_enters_big_mesh(const PrimBox* __restrict__ primBoxes, const BinPrims& bp, V3 o, V3 d)
{
    const V3 rcp = rcp3_cr(d);
    bool hit = !ray_sane(o);        // rays the flat scan refuses reach the mesh without a box test (trace, tn_isect.h)
    // fully unrolled with constant indices: bp lives in kernel-argument SGPRs, a dynamic index would spill it to scratch
#pragma unroll
    for (int k = 0; k < 7; ++k)
    {
        if (k < bp.count && !hit)
        {
            const float4* b = reinterpret_cast<const float4*>(primBoxes + bp.prim[k]);
            const float4 b0 = b[0], b1 = b[1];
            float tb;
            hit = ray_aabb(o, rcp, b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, tb);
        }
    }
    return hit;
}

// ---------------------------------------------------------------------------
// k_generate: camera paths of the batch into buffer 0; region r takes the generation indices [r*regionLen, (r+1)*regionLen)

__global__ __launch_bounds__(kBlock, 4) void k_generate(SplitState ss, QueueCtl q, CameraParams cam, FrameParams fp,
                                                     const uint32_t* __restrict__ passSeeds, const PrimBox* __restrict__ primBoxes, BinPrims bp)
{
    const uint32_t lane = __lane_id();
    uint32_t samples = 0;
    for (uint32_t r = blockIdx.x*(kBlock/kWave) + wave_in_block(); r < ss.numRegions; r += gridDim.x*(kBlock/kWave))
    {
        const uint32_t begin = region_base(ss, r), rLen = region_len(ss, r);
        RegionAppend out = { begin, rLen, 0u, 0u };
        const uint32_t end = (begin + rLen) < fp.genCount ? (begin + rLen) : fp.genCount;
        for (uint32_t i0 = begin; i0 < end; i0 += kWave)
        {
            const uint32_t idx = i0 + lane;
            uint32_t slot = 0;
            bool live = false, front = true;
            PathRegs p;
            if (idx < end && gen_slot(fp, idx, slot))
            {
                float rx, ry;
                if (begin_path(cam, fp, passSeeds, slot, p, rx, ry))
                {
                    live = true;
                    // camera rays that enter a mesh in HBM in front (k_walk takes those)
                    front = bp.count == 0 || ray_enters_big_mesh(primBoxes, bp, p.o, p.d);
                    samples++;
                }
                else
                    ss.radOut[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            const uint32_t pos = out.push(live, front);
            if (live)
            {
                // ray and RNG only: the rest of a fresh path's state is constant and k_shade knows it (ShadeFetch::issue)
                const uint32_t at = sidx(pos);
                ss.rayO[0][at] = make_float4(p.o.x, p.o.y, p.o.z, p.time);
                ss.rayD[0][at] = make_float4(p.d.x, p.d.y, p.d.z, p.bsdfPdf);
                ss.rngId[0][at] = make_float4(__uint_as_float(p.rng.s1), __uint_as_float(p.rng.s2), __uint_as_float(slot), __int_as_float(-1));
            }
        }
        if (lane == 0)
        {
            ss.segFront[r] = out.nFront;
            ss.segBack[r] = out.nBack;
        }
    }
    wave_add_stat(q.stats, 1, samples);
}

// ---------------------------------------------------------------------------
// k_extend: closest hit of every live path

// WONLY: every mesh of the scene is walked by k_walk (2: or is a quad, SceneT): the kernel is the flat scan + record reads, built for more waves
#ifndef TN_WAVES_SCAN
#define TN_WAVES_SCAN 6
#endif
TN_D void store_nee_ray(const SplitState& ss, uint32_t q, int k, const NeeGeo& g)
{
    float4* dst = ss.neeRay + (size_t)(k*2)*ss.capacity + q;
    dst[0] = make_float4(g.o.x, g.o.y, g.o.z, g.dist);
    dst[ss.capacity] = make_float4(g.wi.x, g.wi.y, g.wi.z, g.nl);
}

// SampleLights' RNG draws (render.cpp:107-116, 158-170) for one path per lane: the K shadow rays of every path that hit
// something (`has`; 32 B each), packed like the paths themselves -- in front the paths with a shadow ray that enters a mesh
// in HBM.  Needs only the hit point, its normal and the path's RNG.  The stream is the oracle's: these draws come before
// k_shade's BSDF sample, as SampleLights comes before BSDFSample.  Call with the wave converged (it appends).
template <class SC>
TN_D void draw_shadow_rays(const SC& sc, const SplitState& ss, const BinPrims& bp, int cur, uint32_t pos, bool has, V3 hitP, V3 hitN, float time, RegionAppend& out)
{
    const int K = ss.neePerPath;
    Rng rng;
    NeeGeo ray0;
    V3 skyColor;
    float skyPdf = 0.0f;
    LightCursor lights;
    bool front = bp.count == 0;         // no big mesh: everything goes to the front
    if (has)
    {
        const float2 rr = *reinterpret_cast<const float2*>(ss.rngId[cur] + sidx(pos));
        rng.s1 = __float_as_uint(rr.x); rng.s2 = __float_as_uint(rr.y);

        // the first shadow ray stays in registers across the append; the others are drawn after it
        if (sc.probe.valid)
            nee_sample_probe(sc, hitP, hitN, rng, ray0, skyColor, skyPdf);
        else
            nee_sample_light(sc, hitP, hitN, time, lights.next(sc), rng, ray0);
        front = front || ray_enters_big_mesh(sc.primBoxes, bp, ray0.o, ray0.wi);
        if (!front && K > 1)
        {
            // does ANY of the path's rays enter a mesh in HBM?  a replay of the remaining draws on a copy of the stream
            Rng replay = rng;
            LightCursor lc = lights;
            for (int k = 1; k < K && !front; ++k)
            {
                NeeGeo g;
                nee_sample_light(sc, hitP, hitN, time, lc.next(sc), replay, g);
                front = ray_enters_big_mesh(sc.primBoxes, bp, g.o, g.wi);
            }
        }
    }
    const uint32_t qn = out.push(has, front);
    if (has)
    {
        store_nee_ray(ss, qn, 0, ray0);
        if (sc.probe.valid)
            ss.neeSky[qn] = make_float4(skyColor.x, skyColor.y, skyColor.z, skyPdf);
        for (int k = 1; k < K; ++k)
        {
            NeeGeo g;
            nee_sample_light(sc, hitP, hitN, time, lights.next(sc), rng, g);
            store_nee_ray(ss, qn, k, g);
        }
        ss.neeTime[qn] = time;
        ss.pathNee[hidx1(pos)] = qn;
        *reinterpret_cast<float2*>(ss.rngId[cur] + sidx(pos)) = make_float2(__uint_as_float(rng.s1), __uint_as_float(rng.s2));
    }
}


// k_extend: closest hit of every live path.  The lean variant (WONLY: every mesh of the scene is walked by k_walk, the
// kernel is the flat scan + record reads) also draws the light samples, the hit still in registers: there a kernel of its
// own for them costs more than it saves (524k-triangle config: 2.1 + 2.9 ms apart, 3.9 together); behind the inline mesh
// walk it is the other way round (the fused kernel needs 170 VGPRs: glass 21.2 + 7.9 apart, 31.5 together at 3 waves).
// LIGHTS: the kernel draws the light samples too (always in the lean variant; in the others an A/B: TINSEL_HIP_LIGHTS_IN_EXTEND)
// (five waves per SIMD since the end of round 4: with the libm coefficients out of its registers the staged-arena variant needs 106 VGPRs,
// one granule above the limit; at 96 + 36 B of scratch glass's k_extend runs 6.93 -> 6.32 ms, profiles/r04_w_ab_extend5.md)
#ifndef TN_WAVES_EXTEND_LIGHTS
#define TN_WAVES_EXTEND_LIGHTS 5
#endif
template <bool COUNT, bool LDS, int WONLY = 0, bool MIXED = false, bool LIGHTS = (WONLY != 0)>
__global__ __launch_bounds__(kBlock, WONLY ? TN_WAVES_SCAN_EXTEND : LIGHTS ? TN_WAVES_EXTEND_LIGHTS : TN_WAVES_TRACE) void k_extend(DevScene scIn, SplitState ss, QueueCtl q, int bounce, int stackEntries,
                                                                  const float4* __restrict__ walkRec, uint32_t walkPrims, BinPrims bp, const uint32_t* __restrict__ order)
{
    extern __shared__ uint32_t s_stack[];      // [stackEntries][kBlock], sized at launch
    LdsStack<kBlock> st = { s_stack + threadIdx.x };
    SceneT<LDS, WONLY, 2, MIXED> sc;
    stage_scene_lds(sc, scIn, s_stack + stackEntries*kBlock + kScanWords);

    const uint32_t lane = __lane_id();
    const int cur = bounce & 1;
    const bool lights = LIGHTS && ss.neePerPath > 0;
    uint32_t rays = 0;
    TraceCounters ctr = { 0, 0, 0 };
    sc.walkRec = walkRec;           // k_walk's records of the front rays (null: meshes are walked inline)

    for (uint32_t b = blockIdx.x; b < ss.numRegions/kRegionsPerBlock; b += gridDim.x)
    {
        const uint32_t r = (order ? order[b] : b)*kRegionsPerBlock + wave_in_block();
        const uint32_t nFront = wave_uniform(ss.segFront[(size_t)bounce*ss.numRegions + r]);
        const uint32_t n = nFront + wave_uniform(ss.segBack[(size_t)bounce*ss.numRegions + r]);
        const uint32_t rBase = region_base(ss, r), rLen = region_len(ss, r);
        RegionAppend out = { rBase, rLen, 0u, 0u };
        for (uint32_t j0 = 0; j0 < n; j0 += kWave)
        {
            const uint32_t j = j0 + lane;
            const uint32_t pos = region_pos(rBase, rLen, nFront, j < n ? j : 0u);
            bool has = false;
            V3 hitP, hitN;
            float time = 0.0f;
            if (j < n)
            {
                const float4 ro = ss.rayO[cur][sidx(pos)];
                const float4 rd = ss.rayD[cur][sidx(pos)];
                sc.walkItem = pos*walkPrims;        // only front rays ever reach a walked primitive

                float t;
                const int prim = trace<SceneT<LDS, WONLY, 2, MIXED>, LdsStack<kBlock>, COUNT>(sc, st, V3(ro.x, ro.y, ro.z), V3(rd.x, rd.y, rd.z), ro.w, t, hitN, ctr);

                ss.hit[hidx(pos)] = make_float4(t, hitN.x, hitN.y, hitN.z);
                ss.hitPrim[hidx1(pos)] = prim;
                rays++;
                has = prim >= 0;
                hitP = V3(ro.x, ro.y, ro.z) + V3(rd.x, rd.y, rd.z)*t;       // on_hit_begin's h.p (render.cpp:275)
                time = ro.w;
            }
            if (lights)
                draw_shadow_rays(sc, ss, bp, cur, pos, has, hitP, hitN, time, out);
        }
        if (lights && lane == 0)
        {
            ss.neeFront[(size_t)bounce*ss.numRegions + r] = out.nFront;
            ss.neeBack[(size_t)bounce*ss.numRegions + r] = out.nBack;
        }
    }

    wave_add_stat(q.stats, 0, rays);
    if (COUNT)
    {
        wave_add_stat(q.stats, 2, ctr.internal);
        wave_add_stat(q.stats, 3, ctr.tris);
        wave_add_stat(q.stats, 4, ctr.prims);
    }
}

// k_lights: the light samples of a bounce as a kernel of its own (scenes whose k_extend walks meshes inline)
template <bool LDS, bool MIXED = false>
__global__ __launch_bounds__(kBlock, TN_WAVES_LIGHTS) void k_lights(DevScene scIn, SplitState ss, int bounce, BinPrims bp, const uint32_t* __restrict__ order)
{
    extern __shared__ uint32_t s_arena[];
    SceneT<LDS, false, 2, MIXED> sc;
    stage_scene_lds(sc, scIn, s_arena);
    const uint32_t lane = __lane_id();
    const int cur = bounce & 1;

    for (uint32_t b = blockIdx.x; b < ss.numRegions/kRegionsPerBlock; b += gridDim.x)
    {
        const uint32_t r = (order ? order[b] : b)*kRegionsPerBlock + wave_in_block();
        const uint32_t nFront = wave_uniform(ss.segFront[(size_t)bounce*ss.numRegions + r]);
        const uint32_t n = nFront + wave_uniform(ss.segBack[(size_t)bounce*ss.numRegions + r]);
        const uint32_t rBase = region_base(ss, r), rLen = region_len(ss, r);
        RegionAppend out = { rBase, rLen, 0u, 0u };
        // the next round's records are requested before this round's samples are drawn (see k_shade)
        float4 nro, nrd, nhh;
        int nprim = -1;
        uint32_t npos = region_pos(rBase, rLen, nFront, lane < n ? lane : 0u);
        if (lane < n)
        {
            nro = ss.rayO[cur][sidx(npos)]; nrd = ss.rayD[cur][sidx(npos)]; nhh = ss.hit[hidx(npos)]; nprim = ss.hitPrim[hidx1(npos)];
        }
        for (uint32_t j0 = 0; j0 < n; j0 += kWave)
        {
            const uint32_t j = j0 + lane;
            const uint32_t pos = npos;
            const float4 ro = nro, rd = nrd, hh = nhh;
            const bool has = j < n && nprim >= 0;
            {
                const uint32_t jn = j + kWave;
                npos = region_pos(rBase, rLen, nFront, jn < n ? jn : 0u);
                if (jn < n)
                {
                    nro = ss.rayO[cur][sidx(npos)]; nrd = ss.rayD[cur][sidx(npos)]; nhh = ss.hit[hidx(npos)]; nprim = ss.hitPrim[hidx1(npos)];
                }
            }
            V3 hitP, hitN;
            float time = 0.0f;
            if (has)
            {
                hitP = V3(ro.x, ro.y, ro.z) + V3(rd.x, rd.y, rd.z)*hh.x;       // on_hit_begin's h.p (render.cpp:275)
                hitN = V3(hh.y, hh.z, hh.w);
                time = ro.w;
            }
            draw_shadow_rays(sc, ss, bp, cur, pos, has, hitP, hitN, time, out);
        }
        if (lane == 0)
        {
            ss.neeFront[(size_t)bounce*ss.numRegions + r] = out.nFront;
            ss.neeBack[(size_t)bounce*ss.numRegions + r] = out.nBack;
        }
    }
}

// ---------------------------------------------------------------------------
// The shading half of a bounce (the light samples are drawn):
//   k_shadow   traces the shadow rays and parks, per ray, which primitive's emission arrives (8 B)
//   k_shade    on_hit_begin / on_miss, the BSDF terms of the arriving samples only, totalRadiance += throughput*sum
//              (render.cpp:314), the BSDF step, the survivor to its new position
// (cut where the registers say, -Rpass-analysis=kernel-resource-usage: light sampling with its mesh / moving-primitive
// branches wants ~170 VGPRs beside a live material record, the BSDF step 125).

// k_shadow: the Trace() calls of SampleLights (render.cpp:117, 172) and the tests that follow them (:118, :175-196): one
// lane per path traces its K shadow rays and leaves, per ray, the primitive whose emission arrives (or -1) and its t.
template <bool COUNT, bool LDS, int WONLY = 0, bool MIXED = false>
__global__ __launch_bounds__(kBlock, WONLY ? TN_WAVES_SCAN : TN_WAVES_TRACE) void k_shadow(DevScene scIn, SplitState ss, QueueCtl q, int bounce, int stackEntries,
                                                                  const float4* __restrict__ walkRec, uint32_t walkPrims, const uint32_t* __restrict__ order)
{
    extern __shared__ uint32_t s_stack[];      // [stackEntries][kBlock], sized at launch
    LdsStack<kBlock> st = { s_stack + threadIdx.x };
    SceneT<LDS, WONLY, 2, MIXED> sc;
    stage_scene_lds(sc, scIn, s_stack + stackEntries*kBlock + kScanWords);

    const uint32_t lane = __lane_id();
    uint32_t rays = 0;
    TraceCounters ctr = { 0, 0, 0 };
    sc.walkRec = walkRec;
    const int K = ss.neePerPath;

    for (uint32_t b = blockIdx.x; b < ss.numRegions/kRegionsPerBlock; b += gridDim.x)
    {
        const uint32_t r = (order ? order[b] : b)*kRegionsPerBlock + wave_in_block();
        const uint32_t nFront = wave_uniform(ss.neeFront[(size_t)bounce*ss.numRegions + r]);
        const uint32_t n = nFront + wave_uniform(ss.neeBack[(size_t)bounce*ss.numRegions + r]);
        const uint32_t rBase = region_base(ss, r), rLen = region_len(ss, r);
        for (uint32_t j0 = 0; j0 < n; j0 += kWave)
        {
            const uint32_t j = j0 + lane;
            if (j >= n)
                continue;
            const uint32_t qn = region_pos(rBase, rLen, nFront, j);
            const float time = ss.neeTime[qn];

            for (int k = 0; k < K; ++k)
            {
                const float4* src = ss.neeRay + (size_t)(k*2)*ss.capacity + qn;
                const float4 a = src[0], b = src[ss.capacity];
                NeeGeo ray;
                ray.o = V3(a.x, a.y, a.z); ray.dist = a.w;
                ray.wi = V3(b.x, b.y, b.z); ray.nl = b.w;
                float t;
                V3 n3;
                sc.walkItem = (qn*(uint32_t)K + (uint32_t)k)*walkPrims;
                // the walks of a shadow ray stop at an occluder that decides the sample (shadow_stop, tn_isect.h: the scene BVH here,
                // meshes in HBM in k_walk; many_spheres 1309 -> 1369 Msamples/s, config 3 1923 -> 1959)
                const int hp = trace<SceneT<LDS, WONLY, 2, MIXED>, LdsStack<kBlock>, COUNT, !COUNT>(sc, st, ray.o, ray.wi, time, t, n3, ctr, shadow_stop(ray.dist));
                rays++;
                int arrives;
                if (ray.dist < 0.0f)
                    arrives = (hp < 0) ? 0 : -1;            // probe sample: contributes iff unoccluded
                else
                    arrives = nee_light_reached(ray, hp, t) ? hp : -1;
                ss.neeRes[(size_t)k*ss.capacity + qn] = make_float2(__int_as_float(arrives), t);
            }
        }
    }

    wave_add_stat(q.stats, 0, rays);
    wave_add_stat(q.stats, 5, rays);
    if (COUNT)
    {
        wave_add_stat(q.stats, 2, ctr.internal);
        wave_add_stat(q.stats, 3, ctr.tris);
        wave_add_stat(q.stats, 4, ctr.prims);
    }
}

// what k_shade reads of a path before it can do anything: its state, its hit, where its shadow rays are
struct ShadeFetch
{
    float4 ro, rd, th, ra, rr, hh;
    int prim;
    uint32_t qn;

    // `fresh`: bounce 0 -- throughput, radiance, medium and ray type are path_begin's constants (render.cpp:233-248), which
    // k_generate therefore does not write
    TN_D void issue(const StateBuf& sb, const float4* hit, const int32_t* hitPrim, const uint32_t* pathNee, uint32_t pos, bool valid, bool hasNee, bool fresh)
    {
        if (!valid)
            return;
        const uint32_t at = sidx(pos);
        ro = sb.rayO[at]; rd = sb.rayD[at];
        if (fresh)
        {
            th = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
            ra = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(kReflected));
        }
        else
        {
            th = sb.thr[at]; ra = sb.rad[at];
        }
        rr = sb.rngId[at];
        hh = hit[hidx(pos)];
        prim = hitPrim[hidx1(pos)];
        qn = hasNee ? pathNee[hidx1(pos)] : 0u;
    }
    TN_D void issue(const SplitState& ss, int buf, uint32_t pos, bool valid, bool hasMedia, bool hasNee, bool fresh)
    {
        issue(state_buf(ss, buf), ss.hit, ss.hitPrim, ss.pathNee, pos, valid, hasNee, fresh);
    }

    // (the medium's absorption is the caller's to look up: medium_absorption)
    TN_D void unpack(PathRegs& p, uint32_t& slot) const
    {
        p.o = V3(ro.x, ro.y, ro.z); p.time = ro.w;
        p.d = V3(rd.x, rd.y, rd.z); p.bsdfPdf = rd.w;
        p.thr = V3(th.x, th.y, th.z); p.eta = th.w;
        p.rad = V3(ra.x, ra.y, ra.z); p.rayType = __float_as_int(ra.w);
        p.absorption = V3(0.0f);
        p.rng.s1 = __float_as_uint(rr.x); p.rng.s2 = __float_as_uint(rr.y);
        slot = __float_as_uint(rr.z);
        p.medium = __float_as_int(rr.w);
    }
};

// the shading half of one path (shared by the two k_shade kernels): on_hit_begin / on_miss, the BSDF terms of the arriving light
// samples, the BSDF step; true when the path goes on (its state in `p`, `front`: its next ray enters a mesh in HBM)
template <class SC>
TN_D bool shade_path(SC& sc, const SplitState& ss, const ShadeFetch& f, int bounce, int maxDepth, int rrStart, const BinPrims& bp,
                     PathRegs& p, uint32_t& slot, bool& front)
{
    const int K = ss.neePerPath;
    bool alive = false;
    f.unpack(p, slot);
    p.absorption = medium_absorption(sc, p.medium, sc.hasMedia != 0);
    const int prim = f.prim;
    if (prim < 0)
    {
        on_miss(sc, p, bounce);
    }
    else
    {
        const float4 hh = f.hh;
        const Mat mat = load_mat(sc.mats, prim);

        HitCtx h;
        on_hit_begin(p, mat, hh.x, V3(hh.y, hh.z, hh.w), bounce, h, prim);

        // SampleLights, after the traces (render.cpp:118-139, 171-224): k_shadow left, per shadow ray, the primitive
        // whose emission arrives; the BSDF terms are evaluated for those rays only
        if (K > 0)
        {
            const uint32_t qn = f.qn;
            LightCursor lights;
            const float2* res = ss.neeRes + qn;
            const float4* wis = ss.neeRay + (size_t)ss.capacity + qn;       // {wi, nl} of ray k at wis[k*2*capacity]
            V3 sum = nee_sum(sc, [&](int k) -> V3 {
                const float2 rk = res[(size_t)k*ss.capacity];
                const int hp = __float_as_int(rk.x);
                if (sc.probe.valid && k == 0)
                {
                    if (hp < 0)
                        return V3(0.0f);
                    const float4 w = wis[0], sky = ss.neeSky[qn];
                    return nee_contrib_probe(mat, h, V3(w.x, w.y, w.z), V3(sky.x, sky.y, sky.z), sky.w);
                }
                const int light = lights.next(sc);
                if (hp < 0)
                    return V3(0.0f);
                const float4 w = wis[(size_t)(k*2)*ss.capacity];
                return nee_contrib_light(sc, mat, h, V3(w.x, w.y, w.z), w.w, light, hp, rk.y);
            });
            p.rad = p.rad + p.thr*sum;
        }

        // the last iteration's BSDF sample is never used by the oracle's loop (render.cpp:250)
        if (bounce + 1 < maxDepth)
            alive = bsdf_step(p, mat, h) == kContinue;
        if (alive && rrStart > 0 && bounce + 1 >= rrStart)
            alive = roulette_survives(p);
        if (alive)
            front = bp.count == 0 || ray_enters_big_mesh(sc.primBoxes, bp, p.o, p.d);
    }
    if (!alive)
        ss.radOut[slot] = make_float4(p.rad.x, p.rad.y, p.rad.z, __int_as_float(p.rayType));
    return alive;
}

template <bool LDS, bool MIXED = false>
__global__ __launch_bounds__(kBlock, TN_WAVES_SHADE) void k_shade(DevScene scIn, SplitState ss, int bounce, int maxDepth, int rrStart, BinPrims bp, const uint32_t* __restrict__ order)
{
    extern __shared__ uint32_t s_arena[];
    SceneT<LDS, false, 2, MIXED> sc;
    stage_scene_lds(sc, scIn, s_arena);
    const uint32_t lane = __lane_id();
    const int cur = bounce & 1, nxt = cur ^ 1;
    const int K = ss.neePerPath;
    const bool hasMedia = sc.hasMedia != 0;

    for (uint32_t b = blockIdx.x; b < ss.numRegions/kRegionsPerBlock; b += gridDim.x)
    {
        const uint32_t r = (order ? order[b] : b)*kRegionsPerBlock + wave_in_block();
        const uint32_t nFront = wave_uniform(ss.segFront[(size_t)bounce*ss.numRegions + r]);
        const uint32_t n = nFront + wave_uniform(ss.segBack[(size_t)bounce*ss.numRegions + r]);
        const uint32_t rBase = region_base(ss, r), rLen = region_len(ss, r);
        RegionAppend out = { rBase, rLen, 0u, 0u };
        // A round's records are requested at its start.  (Round 2 requested round i + 1's before shading round i: at two waves per SIMD
        // that hid a latency.  At three, the 30 registers of a second ShadeFetch are spilled ones, and the wait for the shadow-ray
        // records in the middle of the round -- vmcnt counts in order -- waited for the early request as well: without it k_shade spills
        // 124 B instead of 196 and runs 3-9 % faster on the 524k-triangle config and many_spheres, +-1 % on glass,
        // profiles/r03_z3_ab_shade_fetch.md; -DTN_SHADE_PREFETCH=1 is the old arm.  Requesting the shadow-ray records a round ahead too, in
        // registers or through LDS with global_load_lds, spills 352-400 B and doubles the kernel's time.)
        for (uint32_t j0 = 0; j0 < n; j0 += kWave)
        {
            const uint32_t j = j0 + lane;
            ShadeFetch f;
            f.issue(ss, cur, region_pos(rBase, rLen, nFront, j < n ? j : 0u), j < n, hasMedia, K > 0, bounce == 0);
            bool alive = false, front = true;
            PathRegs p;
            uint32_t slot = 0;
            if (j < n)
                alive = shade_path(sc, ss, f, bounce, maxDepth, rrStart, bp, p, slot, front);
            const uint32_t np = out.push(alive, front);
            if (alive)
                store_state(ss, nxt, np, p, slot);
        }
        if (lane == 0)
        {
            ss.segFront[(size_t)(bounce + 1)*ss.numRegions + r] = out.nFront;
            ss.segBack[(size_t)(bounce + 1)*ss.numRegions + r] = out.nBack;
        }
    }
}

// k_shade_sorted: the same shading, the paths of a region taken CLASS BY CLASS instead of in position order -- rays that left
// the scene / surfaces with a transmission or sub-surface lobe / plain opaque surfaces / lights (reference disney.h:172, 178,
// 243, 246: the stochastic lobe choices; render.cpp:365-384 the miss branch, :322 the light-hit termination).  A wave of k_shade
// mixes them and runs each branch with the lanes that take it (71 % of the lanes active on glass, round 2).  Here a wave reads
// the 4-B hit primitive of its region's entries 64 at a time, drops each position into one of four LDS lists (2 KB per wave,
// positions from a wave64 ballot), and whenever a list holds 64 it shades those 64 paths: every branch with a full wave.  What is
// left at the end of the region (fewer than 64 per class) is shaded in mixed rounds, so a region costs at most one round more
// than before.  The price: a class's 64 positions are scattered over the region (16-B gathers inside a 16-KB window per array
// instead of one run), and the next round's records cannot be requested ahead.  Order never changes a result (the paths append
// to the next bounce in another order, that is all).
constexpr int kShadeClasses = 4;
constexpr int kShadeListLen = 128;
constexpr int kShadeListWordsPerWave = kShadeClasses*kShadeListLen;
constexpr int kShadeListWords = kShadeListWordsPerWave*(kBlock/kWave);     // 8 KB per workgroup

template <bool LDS, bool MIXED = false>
__global__ __launch_bounds__(kBlock, TN_WAVES_SHADE) void k_shade_sorted(DevScene scIn, SplitState ss, int bounce, int maxDepth, int rrStart, BinPrims bp, const uint32_t* __restrict__ order)
{
    extern __shared__ uint32_t s_arena[];       // the waves' class lists, then the staged arena
    uint32_t* const list = s_arena + wave_in_block()*kShadeListWordsPerWave;
    SceneT<LDS, false, 2, MIXED> sc;
    stage_scene_lds(sc, scIn, s_arena + kShadeListWords);
    const uint32_t lane = __lane_id();
    const int cur = bounce & 1, nxt = cur ^ 1;
    const int K = ss.neePerPath;
    const bool hasMedia = sc.hasMedia != 0;

    for (uint32_t b = blockIdx.x; b < ss.numRegions/kRegionsPerBlock; b += gridDim.x)
    {
        const uint32_t r = (order ? order[b] : b)*kRegionsPerBlock + wave_in_block();
        const uint32_t nFront = wave_uniform(ss.segFront[(size_t)bounce*ss.numRegions + r]);
        const uint32_t n = nFront + wave_uniform(ss.segBack[(size_t)bounce*ss.numRegions + r]);
        const uint32_t rBase = region_base(ss, r), rLen = region_len(ss, r);
        RegionAppend out = { rBase, rLen, 0u, 0u };
        uint32_t cnt[kShadeClasses] = { 0u, 0u, 0u, 0u };       // wave-uniform

        // One loop, ONE shading site (the shading code is 6,000 instructions: it must not be instantiated per class): every turn either
        // shades 64 paths of a class whose list is full, or -- no list full -- reads the next 64 hit primitives of the region and files
        // their positions, or -- region read -- shades what is left, class after class, in as few rounds as the leftovers' sum needs.
        int nextPrim = -1;          // the hit primitives of the next round are requested a round ahead (4 B per path)
        if (lane < n)
            nextPrim = ss.hitPrim[hidx1(region_pos(rBase, rLen, nFront, lane))];
        uint32_t j0 = 0, e0 = 0;
        for (;;)
        {
            uint32_t pos = 0;
            bool valid = false;
            const int full = cnt[0] >= (uint32_t)kWave ? 0 : cnt[1] >= (uint32_t)kWave ? 1 : cnt[2] >= (uint32_t)kWave ? 2 : cnt[3] >= (uint32_t)kWave ? 3 : -1;
            if (full >= 0)
            {
                uint32_t c0 = 0;
#pragma unroll
                for (int c = 0; c < kShadeClasses; ++c)
                    if (c == full)
                    {
                        cnt[c] -= (uint32_t)kWave;
                        c0 = cnt[c];
                    }
                pos = list[full*kShadeListLen + c0 + lane];
                valid = true;
            }
            else if (j0 < n)
            {
                const uint32_t j = j0 + lane;
                const uint32_t at = region_pos(rBase, rLen, nFront, j < n ? j : 0u);
                const int prim = nextPrim;
                if (j + kWave < n)
                    nextPrim = ss.hitPrim[hidx1(region_pos(rBase, rLen, nFront, j + kWave))];
                int cls = -1;
                if (j < n)
                {
                    if (prim < 0)
                        cls = 0;
                    else
                    {
                        const float4* mp = reinterpret_cast<const float4*>(sc.mats + prim);
                        const float subsurface = mp[2].w, transmission = mp[4].w;
                        const int lightSamples = __float_as_int(mp[5].w);
                        cls = lightSamples ? 3 : (transmission > 0.0f || subsurface > 0.0f) ? 1 : 2;
                    }
                }
#pragma unroll
                for (int c = 0; c < kShadeClasses; ++c)
                {
                    const unsigned long long m = __ballot(cls == c);
                    if (cls == c)
                        list[c*kShadeListLen + cnt[c] + bits_below(m)] = at;
                    cnt[c] += (uint32_t)__popcll(m);
                }
                j0 += kWave;
                continue;
            }
            else
            {
                const uint32_t s1 = cnt[0], s2 = s1 + cnt[1], s3 = s2 + cnt[2], total = s3 + cnt[3];
                if (e0 >= total)
                    break;
                const uint32_t e = e0 + lane;
                if (e < total)
                {
                    const uint32_t c = e >= s3 ? 3u : e >= s2 ? 2u : e >= s1 ? 1u : 0u;
                    pos = list[c*kShadeListLen + (e - (c == 3u ? s3 : c == 2u ? s2 : c == 1u ? s1 : 0u))];
                    valid = true;
                }
                e0 += kWave;
            }

            ShadeFetch f;
            f.issue(ss, cur, pos, valid, hasMedia, K > 0, bounce == 0);
            bool alive = false, front = true;
            PathRegs p;
            uint32_t slot = 0;
            if (valid)
                alive = shade_path(sc, ss, f, bounce, maxDepth, rrStart, bp, p, slot, front);
            const uint32_t np = out.push(alive, front);
            if (alive)
                store_state(ss, nxt, np, p, slot);
        }
        if (lane == 0)
        {
            ss.segFront[(size_t)(bounce + 1)*ss.numRegions + r] = out.nFront;
            ss.segBack[(size_t)(bounce + 1)*ss.numRegions + r] = out.nBack;
        }
    }
}

// ---------------------------------------------------------------------------
// k_walk's work list: the front entries of every region as ONE list of positions, so that its workgroups can cut the
// work into equal static ranges.  k_seg_prefix: exclusive prefix of the per-region front counts (one workgroup);
// k_seg_expand: region r writes base_r + i at prefix[r] + i.
constexpr int kSegBlock = 1024;

// (counts2: a second array added to the first -- front + back counts: every live entry, k_swalk's list; null: the front entries only)
__global__ __launch_bounds__(kSegBlock) void k_seg_prefix(const uint32_t* __restrict__ counts, const uint32_t* __restrict__ counts2, uint32_t numRegions, uint32_t step,
                                                          uint32_t* __restrict__ prefix)
{
    constexpr uint32_t kWaves = kSegBlock/kWave;
    __shared__ uint32_t s_wave[kWaves];
    extern __shared__ uint32_t s_counts[];      // [numRegions]: the counts (and later the prefixes) by region
    const uint32_t lane = __lane_id(), wave = wave_in_block();
    // The scan visits the regions `step` apart: read through that permutation the counts would be 2 x numRegions scattered 4-B loads
    // by ONE workgroup (53 us per launch, 616 launches per default bench run: 4 % of glass's frame).  So they are staged into LDS
    // with coalesced loads first (numRegions <= 32768: 128 KB), scanned there, and the prefixes leave coalesced too.
    for (uint32_t r = threadIdx.x; r < numRegions; r += kSegBlock)
        s_counts[r] = counts[r] + (counts2 ? counts2[r] : 0u);
    __syncthreads();

    // every wave scans one contiguous piece of the PERMUTED sequence, 64 entries per step
    const uint32_t piece = ((numRegions + kWaves - 1u)/kWaves + kWave - 1u)/kWave*kWave;
    const uint32_t begin = wave*piece < numRegions ? wave*piece : numRegions;
    const uint32_t end = (begin + piece) < numRegions ? (begin + piece) : numRegions;

    // entry i of the scan is region i*step mod numRegions (step coprime to numRegions <= 65535, checked by the host: the product
    // fits 32 bits); the remainder is carried along instead of divided out: r(i + 64) = r(i) + 64*step mod numRegions
    const uint32_t stride = (uint32_t)(((unsigned long long)kWave*step) % numRegions);
    uint32_t reg = (uint32_t)(((unsigned long long)(begin + lane)*step) % numRegions);
    uint32_t sum = 0;
    {
        uint32_t rr = reg;
        for (uint32_t i = begin + lane; i < end; i += kWave)
        {
            sum += s_counts[rr];
            rr += stride;
            if (rr >= numRegions) rr -= numRegions;
        }
    }
    for (int off = 32; off > 0; off >>= 1)
        sum += __shfl_xor(sum, off);
    if (lane == 0)
        s_wave[wave] = sum;
    __syncthreads();

    uint32_t run = 0, total = 0;
    for (uint32_t w = 0; w < kWaves; ++w)
    {
        if (w < wave) run += s_wave[w];
        total += s_wave[w];
    }
    for (uint32_t i0 = begin; i0 < end; i0 += kWave)
    {
        const uint32_t i = i0 + lane;
        const uint32_t v = i < end ? s_counts[reg] : 0u;
        uint32_t x = v;                                   // inclusive scan across the wave
        for (int off = 1; off < kWave; off <<= 1)
        {
            const uint32_t y = __shfl_up(x, off);
            if ((int)lane >= off) x += y;
        }
        if (i < end)
            s_counts[reg] = run + x - v;                  // (each region is visited once: the permutation is a bijection)
        run += __shfl(x, kWave - 1);
        reg += stride;
        if (reg >= numRegions) reg -= numRegions;
    }
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < numRegions; r += kSegBlock)
        prefix[r] = s_counts[r];
    if (threadIdx.x == 0)
        prefix[numRegions] = total;
}

__global__ __launch_bounds__(kBlock) void k_seg_expand(const uint32_t* __restrict__ counts, const uint32_t* __restrict__ prefix, SplitState ss, uint32_t* __restrict__ list)
{
    const uint32_t lane = __lane_id();
    for (uint32_t r = blockIdx.x*(kBlock/kWave) + wave_in_block(); r < ss.numRegions; r += gridDim.x*(kBlock/kWave))
    {
        const uint32_t n = wave_uniform(counts[r]), at = wave_uniform(prefix[r]), base = region_base(ss, r);
        for (uint32_t i = lane; i < n; i += kWave)
            list[at + i] = base + i;
    }
}

} // namespace tn
