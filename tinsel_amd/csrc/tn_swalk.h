// tn_swalk.h -- k_swalk: the SCENE-level walk (QueryBVH, intersection.h:751-799, under Trace, render.cpp:17-62) as a kernel with
// ray replacement, for scenes whose primitives do not fit the wave-uniform flat scan (more than 64 primitives: many_spheres).
//
// There k_extend / k_shadow (tn_kernels.h) gave one lane one ray per round, and a round lasted as long as its slowest ray:
// 12.2 + 12.1 of many_spheres' 35 ms (round 2), four waves per SIMD, lanes waiting on their wave-mates.  k_swalk is k_walk's
// machinery (tn_walk.h) one level up: a workgroup owns a contiguous range of the bounce's rays and an LDS cursor into it, a lane
// whose ray is finished takes the next one; per iteration a lane pops ONE entry of its LDS stack -- an internal node (both
// children's boxes in one 64-B record, pushed in the oracle's order) or a leaf, whose PrimitiveIntersect waits until enough
// lanes of the wave have one too.  What a ray does is what trace() does, test for test in the same order (no closest-t cull
// at this level, strict `t < minT`, meshes walked inline on the stack above the scene level), so the results are the
// oracle's bit for bit; shadow rays stop at an occluder that decides them (shadow_stop, tn_isect.h) like k_shadow's.
// Extension rays leave the hit where k_shade looks for it (SplitState::hit / hitPrim), shadow rays their 8-B verdict (neeRes).
#pragma once

#include "tn_split.h"

namespace tn {

#ifndef TN_WAVES_SWALK
#define TN_WAVES_SWALK 4
#endif

struct SwalkJob
{
    const uint32_t* list;           // positions of the bounce's live paths / shadow-ray bundles (k_seg_expand_all), *count of them
    const uint32_t* count;
    int neePerPath;                 // 0: extension rays; K > 0: the K shadow rays of every listed bundle
    int stackEntries;               // LDS stack entries per lane (scene level + the deepest mesh)
    int refillMin;                  // idle lanes that trigger a refill
    int leafMin;                    // lanes waiting at a leaf that trigger the leaf phase
};

constexpr int kSwalkCtlWords = 16;

// the listed positions of a region packed at both ends: entry i of the region's n = nFront + nBack live entries
__global__ __launch_bounds__(kBlock) void k_seg_expand_all(const uint32_t* __restrict__ front, const uint32_t* __restrict__ back, const uint32_t* __restrict__ prefix,
                                                           SplitState ss, uint32_t* __restrict__ list)
{
    const uint32_t lane = __lane_id();
    for (uint32_t r = blockIdx.x*(kBlock/kWave) + wave_in_block(); r < ss.numRegions; r += gridDim.x*(kBlock/kWave))
    {
        const uint32_t nF = wave_uniform(front[r]), n = nF + wave_uniform(back[r]), at = wave_uniform(prefix[r]);
        const uint32_t base = region_base(ss, r), len = region_len(ss, r);
        for (uint32_t i = lane; i < n; i += kWave)
            list[at + i] = region_pos(base, len, nF, i);
    }
}

// BLOCK / LDSMODE: 1024-thread workgroups (one per CU) whose LDS holds the stacks AND the whole scene arena -- the scene BVH, the
// primitive and material records: a node visit is four ds_read_b128 instead of four global loads, and the walk of an
// L1-resident tree is bound by the rate a CU's vector-memory front end retires records (tn_walk.h) -- with every scene access at a
// compile-time LDS address (1: every mesh rides in the arena too; 2: some meshes live in HBM); 0: 256-thread workgroups and
// generic pointers, for arenas that do not fit beside the stacks.
template <bool SHADOW, int BLOCK, int LDSMODE>
__global__ __launch_bounds__(BLOCK, BLOCK == 1024 ? 4 : TN_WAVES_SWALK) void k_swalk(DevScene scIn, SplitState ss, QueueCtl q, int bounce, SwalkJob job)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_sw[];     // [stackEntries][BLOCK] stack words, control words, the staged arena (if any)
    LdsStack<BLOCK> st = { s_sw + threadIdx.x };
    uint32_t* const s_ctl = s_sw + job.stackEntries*BLOCK;
    typedef SceneT<LDSMODE != 0, false, 2, LDSMODE == 2> SC;
    SC sc;
    stage_scene_lds(sc, scIn, s_ctl + kSwalkCtlWords, BLOCK);

    const int lane = (int)__lane_id();
    const unsigned long long below = (1ull << lane) - 1ull;
    const int cur = bounce & 1;
    const uint32_t Kx = SHADOW ? (uint32_t)job.neePerPath : 1u;
    const uint32_t total = (*job.count)*Kx;

    // static ranges: workgroup b -> the b-th contiguous piece of the rays; its waves share it through an LDS cursor
    const uint32_t chunk = (total + gridDim.x - 1u)/gridDim.x;
    const uint32_t bbeg = blockIdx.x*chunk < total ? blockIdx.x*chunk : total;
    const uint32_t end = (bbeg + chunk) < total ? (bbeg + chunk) : total;
    if (threadIdx.x == 0)
        s_ctl[0] = bbeg;
    __syncthreads();

    // per-lane walk state
    bool active = false, pending = false;
    bool exhausted = bbeg >= end;               // wave-uniform: the workgroup's range has been handed out
    uint32_t pos = 0, kray = 0, ref = 0;
    bool haveRef = false;                       // a popped entry waits in `ref` (a leaf waiting for its phase)
    int sp = 0;
    V3 o, d, rcp;
    float time = 0.0f, minT = kFltMax, tStop = 0.0f, dist = 0.0f, nl = 0.0f;
    int closest = -1;
    V3 cn;
    uint32_t rays = 0;
    TraceCounters ctr = { 0, 0, 0 };

    auto write_result = [&]() {
        if (SHADOW)
        {
            int arrives;
            if (dist < 0.0f)
                arrives = (closest < 0) ? 0 : -1;           // probe sample: contributes iff unoccluded (render.cpp:118)
            else
            {
                NeeGeo g;
                g.dist = dist; g.nl = nl;
                arrives = nee_light_reached(g, closest, minT) ? closest : -1;
            }
            ss.neeRes[(size_t)kray*ss.capacity + pos] = make_float2(__int_as_float(arrives), minT);
        }
        else
        {
            const V3 n = face_forward(cn, -d);              // render.cpp:59
            ss.hit[hidx(pos)] = make_float4(minT, n.x, n.y, n.z);
            ss.hitPrim[hidx1(pos)] = closest;
        }
    };

    const uint32_t KxM = (uint32_t)__builtin_amdgcn_readfirstlane((int)(0xffffffffu/Kx));
    for (;;)
    {
        // ---- refill: idle lanes take the next rays of the workgroup's range ------------------------------------------------
        const unsigned long long idleMask = __ballot(!active);
        const int nIdle = __popcll(idleMask);
        if (!exhausted && nIdle >= job.refillMin)
        {
            uint32_t c0 = 0;
            if (lane == 0)
                c0 = atomicAdd(&s_ctl[0], (uint32_t)nIdle);         // LDS atomic: one per refill
            c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c0);
            exhausted = c0 + (uint32_t)nIdle >= end;
            if (!active)
            {
                if (pending)
                {
                    write_result();         // (here, beside the next ray's loads: stores count against vmcnt like loads, tn_walk.h)
                    pending = false;
                }
                const uint32_t my = c0 + (uint32_t)__popcll(idleMask & below);
                if (my < end)
                {
                    uint32_t qi = __umulhi(my, KxM);                 // my/Kx with the uniform reciprocal in an SGPR (tn_walk.h)
                    kray = my - qi*Kx;
                    if (kray >= Kx) { ++qi; kray -= Kx; }
                    pos = job.list[qi];
                    float4 ro, rd;
                    if (SHADOW)
                    {
                        const float4* np = ss.neeRay + (size_t)(kray*2u)*ss.capacity + pos;
                        ro = np[0]; rd = np[ss.capacity];
                        time = ss.neeTime[pos];
                        dist = ro.w; nl = rd.w;
                        tStop = shadow_stop(dist);
                    }
                    else
                    {
                        ro = ss.rayO[cur][sidx(pos)]; rd = ss.rayD[cur][sidx(pos)];
                        time = ro.w;
                    }
                    o = V3(ro.x, ro.y, ro.z);
                    d = V3(rd.x, rd.y, rd.z);
                    rcp = rcp3_cr(d);
                    minT = kFltMax;
                    closest = -1;
                    cn = V3(0.0f);
                    st.set(0, sc.root);             // Trace -> QueryBVH from the root (trace(), tn_isect.h)
                    sp = 1;
                    haveRef = false;
                    active = true;
                    rays++;
                }
            }
        }

        if (__ballot(active) == 0ull)
        {
            if (exhausted)
                break;
            continue;
        }

        // ---- pop: a lane without a waiting entry takes the top of its stack ---------------------------------------------------
        if (active && !haveRef)
        {
            ref = st.get(--sp);
            haveRef = true;
        }

        // ---- node phase: lanes at an internal node ----------------------------------------------------------------------
        if (active && !(ref & kLeafBit))
        {
            const Node64 nd = load_node(sc.nodes, ref);
            float tL, tR;
            const bool hL = ray_aabb(o, rcp, nd.lminx, nd.lminy, nd.lminz, nd.lmaxx, nd.lmaxy, nd.lmaxz, tL);
            const bool hR = ray_aabb(o, rcp, nd.rminx, nd.rminy, nd.rminz, nd.rmaxx, nd.rmaxy, nd.rmaxz, tR);
            uint32_t first = nd.left, second = nd.right;
            if (hL && hR && (tL < tR))
            {
                first = nd.right;
                second = nd.left;
            }
            if (hL)
                st.set(sp++, first);
            if (hR)
                st.set(sp++, second);
            haveRef = false;
        }

        // ---- leaf phase: once enough lanes wait at a leaf (or nobody has a node to visit) ---------------------------------------
        const bool atLeaf = active && haveRef;
        const unsigned long long leafMask = __ballot(atLeaf);
        if (leafMask != 0ull && (__popcll(leafMask) >= job.leafMin || __ballot(active && !atLeaf && sp > 0) == 0ull))
        {
            if (atLeaf)
            {
                float t;
                V3 n;
                const int index = (int)(ref & ~kLeafBit);
                if (prim_intersect<SC, LdsStack<BLOCK>, false, SHADOW>(sc, index, st, sp, o, d, time, t, n, ctr, tStop))
                {
                    if (t < minT && t > 0.0f)
                    {
                        minT = t;
                        closest = index;
                        cn = n;
                        if (SHADOW && t < tStop)
                            sp = 0;             // decided (shadow_stop): drop what is left on the stack
                    }
                }
                haveRef = false;
            }
        }

        // ---- the ray is done when nothing waits and the stack is empty --------------------------------------------------------
        if (active && !haveRef && sp == 0)
        {
            active = false;
            pending = true;             // the result stays in registers until the lane's next refill
        }
    }
    if (pending)
        write_result();

    wave_add_stat(q.stats, 0, rays);
    if (SHADOW)
        wave_add_stat(q.stats, 5, rays);
}

} // namespace tn
