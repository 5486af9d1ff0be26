// NOT PART OF THE LIBRARY: kept for the record (lead ii).  Built, wired behind a tuning field (walk_two_rays), bit-identical on
// every fixture through both wavefront pipelines (30 tests, call 2q) -- and slower: the 524k-triangle config's walk 10.2-10.5 ms -> 13.4 as one
// 1024-thread workgroup per CU (four waves per SIMD), 10.9-11.6 as two 768-thread ones (six waves), thresholds swept
// (profiles/r06_2q_ab_walk_two_rays.md, r06_2r_ab_walk_two_rays_six_waves.md; EXPERIMENTS.md).
//
// tn_walk2.h -- k_walk2: k_walk (tn_walk.h, ONE walked primitive) with TWO rays per lane.
//
// What bounds k_walk is the CU's texture addresser: a 64-B node is four global_load_dwordx4, each of them occupies the TA for 16 cycles
// per WAVE whatever the number of lanes that ask (TA busy 83 % of the kernel's time on the 524k-triangle config, 21 busy cycles per wave
// instruction; profiles/r06_2p_k_walk_ta_counters.txt), and k_walk's node phases run with 38 of 64 lanes, its triangle phases with 10: a lane
// whose ray waits at a leaf, or is finished and waits for the refill, leaves its share of every instruction unused.  Here a lane holds two
// rays (slots A and B).  A phase takes, per lane, whichever slot has work of its kind -- slot A first --, so a lane sits a node phase out only
// when BOTH its rays wait at a leaf or are finished; the triangle phase likewise.  One workgroup of 1024 threads per CU (four waves per SIMD,
// as many rays in flight as k_walk's eight waves of one), whose LDS holds two stacks per lane and a LARGER top of the tree.
//
// A ray's own sequence of node visits, triangle tests and pops is exactly k_walk's (and therefore IntersectRayMesh's, intersection.h:678-749):
// only which rays share an instruction changes, so the records are the same bit for bit.
#pragma once

#include "tn_walk.h"

namespace tn {

constexpr int kWalk2LaneRows = 4;       // per slot: the ray's record index, a shadow ray's stop distance

template <int BLOCK, int WAVES>
__global__ __launch_bounds__(BLOCK, WAVES) void k_walk2(DevScene sc, WalkJob job)
{
    constexpr uint32_t kAtLeaf = kLeafBit;
    constexpr uint32_t kNoItem = 0xffffffffu;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_walk[];
    const int E = job.stackEntries;                             // LDS stack entries per SLOT
    uint32_t* const stackA = s_walk + threadIdx.x;              // slot A: entry i at stackA[i*BLOCK]; slot B: entry i at stackA[(E + i)*BLOCK]
    uint32_t* const spillBase = job.overflow ? job.overflow + ((size_t)blockIdx.x*BLOCK + threadIdx.x)*(size_t)job.overflowEntries*2u : nullptr;
    uint32_t* const s_itemA = s_walk + (2*E + 0)*BLOCK + threadIdx.x;
    uint32_t* const s_itemB = s_walk + (2*E + 1)*BLOCK + threadIdx.x;
    float* const s_stopA = reinterpret_cast<float*>(s_walk + (2*E + 2)*BLOCK + threadIdx.x);
    float* const s_stopB = reinterpret_cast<float*>(s_walk + (2*E + 3)*BLOCK + threadIdx.x);
    uint32_t* const s_ctl = s_walk + (2*E + kWalk2LaneRows)*BLOCK;
    WalkF4* const s_top = reinterpret_cast<WalkF4*>(s_ctl + kWalkCtlWords);

    const int lane = (int)__lane_id();
    const uint32_t Kx = job.mixed == 1 ? (uint32_t)job.neePerPath + 1u : job.neePerPath > 0 ? (uint32_t)job.neePerPath : 1u;
    const uint32_t per = Kx;                                    // (one walked primitive)
    const uint32_t total = (*job.frontCount)*per;
    const uint32_t perM = (uint32_t)__builtin_amdgcn_readfirstlane((int)(0xffffffffu/per));

    const uint32_t chunk = (total + gridDim.x - 1u)/gridDim.x;
    const uint32_t bbeg = blockIdx.x*chunk < total ? blockIdx.x*chunk : total;
    const uint32_t end = (bbeg + chunk) < total ? (bbeg + chunk) : total;
    if (threadIdx.x == 0)
        s_ctl[0] = bbeg;
    *s_itemA = kNoItem; *s_itemB = kNoItem;
    *s_stopA = -kFltMax; *s_stopB = -kFltMax;

    const Prim64 prim0 = load_prim(sc.prims, job.prim[0]);
    const DevMesh* mesh0p = sc.meshes + prim0.mesh;
    GlobalF4 mesh0nodes = as_global(mesh0p->nodes), mesh0tris = as_global(mesh0p->tris);
    const uint32_t mesh0root = mesh0p->root;
    const uint32_t top0N = (uint32_t)job.topCount[0];
    if (bbeg < end && top0N > 0)
        for (uint32_t i = threadIdx.x; i < top0N*4u; i += BLOCK)
            s_top[i] = mesh0nodes[i];
    __syncthreads();

    float4 box0a, box0b;
    {
        const float4* bp = reinterpret_cast<const float4*>(sc.primBoxes + job.prim[0]);
        box0a = bp[0]; box0b = bp[1];
    }

    struct Slot
    {
        uint32_t ref;
        int sp;
        V3 o, d, rcp;
        float closestT, hv, hw, hsign;
        int htri;
    };
    Slot A, B;
    A.ref = B.ref = kNoNode;
    A.sp = B.sp = 0;
    A.closestT = B.closestT = kFltMax;
    A.hv = A.hw = A.hsign = B.hv = B.hw = B.hsign = 0.0f;
    A.htri = B.htri = -1;
    A.o = A.d = A.rcp = B.o = B.d = B.rcp = V3(0.0f);
    bool finiteAll = true;
    bool exhausted = bbeg >= end;

    auto write_record = [&](const Slot& s, uint32_t* s_item) {
        const uint32_t item = *s_item;
        if (item != kNoItem)
        {
            float4* out = job.rec + (size_t)item*2;
            out[0] = make_float4(s.closestT, 1.0f - s.hv - s.hw, s.hv, s.hw);
            if (s.closestT < kFltMax)
            {
                const V3 hn = hit_normal(mesh0tris, s.htri, s.hsign);
                out[1] = make_float4(hn.x, hn.y, hn.z, __int_as_float(s.htri));
            }
            *s_item = kNoItem;
        }
    };
    // a slot takes item `my` of the workgroup's range (k_walk's refill, tn_walk.h)
    auto take = [&](Slot& s, uint32_t my, uint32_t* s_item, float* s_stop) {
        write_record(s, s_item);        // (the finished ray's record, next to the next ray's loads: stores count against vmcnt like loads)
        if (my >= end)
            return;
        uint32_t qi = __umulhi(my, perM), k = my - qi*per;
        if (k >= per) { ++qi; k -= per; }
        const uint32_t slot = job.queue[qi];
        const uint32_t recAt = slot*per + k;
        float4 ro, rd;
        float time;
        if (job.neePerPath > 0 && !(job.mixed == 1 && k == (uint32_t)job.neePerPath))
        {
            const float4* np = job.nee + (size_t)(k*2u)*job.neeStride + slot;
            ro = np[0]; rd = np[job.neeStride];
            time = job.mixed ? job.rayO[sidx(slot)].w : job.neeTime[slot];
            *s_stop = shadow_stop(ro.w);
        }
        else
        {
            ro = job.rayO[sidx(slot)]; rd = job.rayD[sidx(slot)];
            time = ro.w;
            *s_stop = -kFltMax;
        }
        const V3 wo(ro.x, ro.y, ro.z), wd(rd.x, rd.y, rd.z);
        const bool noRay = job.mixed == 1 && rd.x == 0.0f && rd.y == 0.0f && rd.z == 0.0f;
        const V3 wrcp = rcp3_cr(wd);
        float tbox;
        bool enters = true;
        if (__float_as_uint(box0b.z) == 0u && ray_sane(wo))
            enters = ray_aabb(wo, wrcp, box0a.x, box0a.y, box0a.z, box0a.w, box0b.x, box0b.y, tbox);
        if (noRay)
        {
        }
        else if (!enters)
            job.rec[(size_t)recAt*2] = make_float4(kFltMax, 0.0f, 0.0f, 0.0f);
        else
        {
            const Xform x = prim_pose(sc, prim0, time);
            pose_inv_ray(prim0, x, wo, wd, s.o, s.d, s.rcp, wrcp, true);
            s.ref = mesh0root;
            s.sp = 0;
            s.closestT = kFltMax;
            s.htri = -1;
            *s_item = recAt;
        }
    };

    for (;;)
    {
        // ---- refill: idle SLOTS take the next items of the workgroup's range (slot A's lanes first, then slot B's) ------------------
        const unsigned long long idleA = __ballot(A.ref == kNoNode), idleB = __ballot(B.ref == kNoNode);
        const int nA = __popcll(idleA), nB = __popcll(idleB);
        if (!exhausted && nA + nB >= job.refillMin)
        {
            uint32_t cur = 0;
            if (lane == 0)
                cur = atomicAdd(&s_ctl[0], (uint32_t)(nA + nB));
            cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur);
            exhausted = cur + (uint32_t)(nA + nB) >= end;
            if (A.ref == kNoNode)
                take(A, cur + __builtin_amdgcn_mbcnt_hi((uint32_t)(idleA >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idleA, 0u)), s_itemA, s_stopA);
            if (B.ref == kNoNode)
                take(B, cur + (uint32_t)nA + __builtin_amdgcn_mbcnt_hi((uint32_t)(idleB >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idleB, 0u)), s_itemB, s_stopB);
            finiteAll = __all((A.ref == kNoNode || (finite_bits(A.rcp.x) && finite_bits(A.rcp.y) && finite_bits(A.rcp.z) &&
                                                    finite_bits(A.o.x) && finite_bits(A.o.y) && finite_bits(A.o.z))) &&
                              (B.ref == kNoNode || (finite_bits(B.rcp.x) && finite_bits(B.rcp.y) && finite_bits(B.rcp.z) &&
                                                    finite_bits(B.o.x) && finite_bits(B.o.y) && finite_bits(B.o.z))));
        }

        if (__ballot(A.ref != kNoNode || B.ref != kNoNode) == 0ull)
        {
            if (exhausted)
                break;
            continue;
        }

        // ---- node phase: per lane the slot that is at an internal node (A first) -----------------------------------------------------
        const bool nodeA = A.ref != kNoNode && !(A.ref & kAtLeaf), nodeB = B.ref != kNoNode && !(B.ref & kAtLeaf);
        bool popA = false, popB = false;
        if (nodeA || nodeB)
        {
            const bool useA = nodeA;
            const uint32_t ref = useA ? A.ref : B.ref;
            const V3 o = useA ? A.o : B.o, rcp = useA ? A.rcp : B.rcp;
            const float closestT = useA ? A.closestT : B.closestT;
            int sp = useA ? A.sp : B.sp;
            Node64 nd;
            if (ref < top0N)
                nd = load_node_from((const WalkF4*)s_top, ref);
            else
                nd = load_node_from(mesh0nodes, ref);
            float tL, tR;
            bool hL, hR;
            if (finiteAll)
            {
                hL = ray_aabb_minmax(o, rcp, nd.lminx, nd.lminy, nd.lminz, nd.lmaxx, nd.lmaxy, nd.lmaxz, tL);
                hR = ray_aabb_minmax(o, rcp, nd.rminx, nd.rminy, nd.rminz, nd.rmaxx, nd.rmaxy, nd.rmaxz, tR);
            }
            else
            {
                tL = tR = 0.0f;
                hL = ray_aabb(o, rcp, nd.lminx, nd.lminy, nd.lminz, nd.lmaxx, nd.lmaxy, nd.lmaxz, tL);
                hR = ray_aabb(o, rcp, nd.rminx, nd.rminy, nd.rminz, nd.rmaxx, nd.rmaxy, nd.rmaxz, tR);
            }
            hL = hL && tL < closestT;
            hR = hR && tR < closestT;
            uint32_t next = ref;
            bool pop = false;
            if (hL && hR)
            {
                const bool leftNear = tL < tR;
                const uint32_t far = leftNear ? nd.right : nd.left;
                if (sp < E)
                    stackA[((useA ? 0 : E) + sp)*BLOCK] = far;
                else
                    spillBase[(useA ? 0 : job.overflowEntries) + (sp - E)] = far;
                ++sp;
                next = leftNear ? nd.left : nd.right;
            }
            else if (hL)
                next = nd.left;
            else if (hR)
                next = nd.right;
            else
                pop = true;
            if (useA) { A.ref = next; A.sp = sp; popA = pop; }
            else      { B.ref = next; B.sp = sp; popB = pop; }
        }

        // ---- triangle phase: per lane the slot that waits at a leaf (A first), once enough lanes have one or nobody has a node to visit ---
        const bool leafA = A.ref != kNoNode && !popA && (A.ref & kAtLeaf), leafB = B.ref != kNoNode && !popB && (B.ref & kAtLeaf);
        const unsigned long long leafMask = __ballot(leafA || leafB);
        const bool moreNodes = (A.ref != kNoNode && !popA && !(A.ref & kAtLeaf)) || (B.ref != kNoNode && !popB && !(B.ref & kAtLeaf));
        if (leafMask != 0ull && (__popcll(leafMask) >= job.leafMin || __ballot(moreNodes) == 0ull))
        {
            if (leafA || leafB)
            {
                const bool useA = leafA;
                const uint32_t idx = (useA ? A.ref : B.ref) & ~kLeafBit;
                const V3 o = useA ? A.o : B.o, d = useA ? A.d : B.d;
                GlobalF4 tp = mesh0tris + (size_t)idx*3;
                const WalkF4 q3 = tp[0], q4 = tp[1], q5 = tp[2];
                float t, u, v, w, sign;
                V3 n;
                const bool hit = ray_tri(o, d, V3(q3.x, q3.y, q3.z), V3(q4.x, q4.y, q4.z), V3(q5.x, q5.y, q5.z), t, u, v, w, sign, n);
                if (useA)
                {
                    if (hit && t > 0.0f && t < A.closestT)
                    {
                        A.closestT = t; A.hv = v; A.hw = w; A.htri = (int)idx; A.hsign = sign;
                        if (t < *s_stopA)
                            A.sp = 0;
                    }
                    popA = true;
                }
                else
                {
                    if (hit && t > 0.0f && t < B.closestT)
                    {
                        B.closestT = t; B.hv = v; B.hw = w; B.htri = (int)idx; B.hsign = sign;
                        if (t < *s_stopB)
                            B.sp = 0;
                    }
                    popB = true;
                }
            }
        }

        // ---- next entry, or the ray is done ----------------------------------------------------------------------------------------------
        if (popA)
        {
            if (A.sp > 0)
            {
                --A.sp;
                A.ref = A.sp < E ? stackA[A.sp*BLOCK] : spillBase[A.sp - E];
            }
            else
                A.ref = kNoNode;
        }
        if (popB)
        {
            if (B.sp > 0)
            {
                --B.sp;
                B.ref = B.sp < E ? stackA[(E + B.sp)*BLOCK] : spillBase[job.overflowEntries + (B.sp - E)];
            }
            else
                B.ref = kNoNode;
        }
    }
    write_record(A, s_itemA);
    write_record(B, s_itemB);
}

} // namespace tn
