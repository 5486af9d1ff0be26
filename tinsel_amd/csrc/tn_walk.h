// tn_walk.h -- k_walk / k_walk_rays: the closest-hit walk of meshes that live in HBM, as its own lean streaming kernel (k_walk: ONE walked
// primitive, its tree as kernel-argument scalars; k_walk_rays, at the end of the file: several, a work item is a ray).
//
// Why a kernel of its own.  The scan kernels (k_extend / k_shadow, tn_kernels.h) carry the whole scene-level
// state of Trace() (render.cpp:17-62) around the mesh walk of IntersectRayMesh (intersection.h:661-749): 128
// VGPRs, a wave as slow as its slowest ray (34 % of the VALU lanes active on the 524,288-triangle config,
// profiles/r01_f_ajax_kernel_stats.md).  k_walk runs ONLY that walk, for the rays whose leaf-box test against such
// a mesh succeeds (the front part of the sorted queues), and parks the mesh-space closest hit (32 B) in HBM; the
// scan kernels then read the record where they would have called ray_mesh.
//
// What bounds the walk on gfx950 (measured, scratch/ubench/gather_bench.hip + the -DTN_WALK_PROF section timers):
// a CU's vector-memory front end retires ~0.35 random 64-B records per clock whatever the load shape (4 x dwordx4
// per lane, quad-cooperative, LDS-DMA: all 205 G records/s chip-wide from L1/L2, 67 G/s once the set spills to the
// Infinity Cache), and the first version of this kernel sat at that rate with the VALUs 37 % busy.  So the design
// spends LDS, which reads 64 B per lane an order of magnitude faster, on the part of the tree every ray visits:
//   * TOP OF THE TREE IN LDS: the host numbers the upper levels of a walked tree breadth-first (convert_bvh,
//     tinsel_hip.hip), one 1024-thread workgroup per CU stages the first `topCount` Node64 records (as many as fit
//     beside the stacks, ~60 KB) and every visit to them is four ds_read_b128 instead of four global loads;
//   * RAY REPLACEMENT: a workgroup owns a contiguous range of work items and an LDS cursor into it; a lane whose ray
//     is finished takes the next item, so lanes stay busy until the range is exhausted (no global atomics: ranges
//     are static, the dispatcher balances workgroups);
//   * the near child stays in a register (only the far child of a both-hit node goes to the LDS stack);
//   * v_min/v_max box tests when no 0*inf can occur in the wave (the reference's `a < b ? a : b` ternaries and the
//     hardware min/max differ only on NaN and on the sign of zero, neither of which can reach a decision).
// Same boxes, same triangles, same visit order per ray as ray_mesh (tn_isect.h): results are bit-identical
// (asserted against the inline walk of the other pipelines in tests/test_gpu_parity.py).
// Tried and measured, not kept: XCD-aware range mapping (each XCD a contiguous eighth of the queue so that its L2 holds
// one patch of the tree: 24.1 -> 29.4 ms; the interleaved ranges already share their working set in TIME), 8 waves per
// SIMD at 64 VGPRs (spills: 24.8 -> 38.6 ms); node and triangle fetches issued together and waited for once per iteration
// (no separate triangle phase: 22.5 -> 24.8 ms -- the triangle arithmetic then runs every iteration for ~6 lanes); triangles
// requested when a lane arrives at a leaf and tested an iteration later with the data in registers (21.1 -> 25.0 ms: the
// extra iteration of waiting costs more than the second round trip it saves); work claimed dynamically instead of static
// 1/256 ranges, to even out the tail (3.3 of 4 waves per SIMD resident on average) -- per wave in chunks of 1024 items
// (21.0 -> 24.8 ms) or per workgroup in chunks of 16384 shared through a 64-bit LDS {cursor, end} (25.4 ms): a CU that stays
// in ONE contiguous part of the queue for the whole launch keeps that part's subtrees in its L1 / its XCD's L2.
// Round 3: several node visits per turn of the loop, a lane that finds no child popping at once instead of at the turn's end (the
// refill check, the leaf ballots and the bookkeeping are a quarter of the wave cycles): 16.7 -> 17.7 ms with the pop moved alone, 17.3 /
// 17.1 / 17.3 with 2 / 3 / 4 visits per turn (profiles/r03_o_ab_walk_node_reps.txt) -- the pop's LDS read inside the node phase and the
// registers it keeps alive cost more than the turns it saves.
#pragma once

#include "tn_isect.h"
#include "tn_layout.h"

namespace tn {

constexpr int kWalkMaxPrims = 7;

struct WalkJob
{
    const uint32_t* queue;          // the positions to walk (k_seg_expand: the front entries of every region), *frontCount of them
    const uint32_t* frontCount;
    const float4* rayO;             // extension rays: origin|time, dir|- by path position (SplitState::rayO / rayD of the bounce: [sidx(position)])
    const float4* rayD;
    const float4* nee;              // shadow rays: SplitState::neeRay [(k*2 + {0: o|dist, 1: wi|nl})*neeStride + q] by NEE position q
    uint32_t neeStride;
    const float* neeTime;           // shadow rays: SplitState::neeTime [q]
    float4* rec;                    // out: [((position*K + k)*numPrims + walked primitive)][2] = {t,u,v,w} {n.xyz, tri};  t == FLT_MAX: no hit
    int neePerPath;                 // 0: extension rays; K > 0: the K shadow rays of every queued slot
    int mixed;                      // paired pipeline (tn_paired.h).  1: K + 1 rays per queued slot -- its K shadow rays (`nee`), then its extension ray (rayO / rayD;
                                    // an all-zero direction: the path has none).  2: its K shadow rays only.  Either way a shadow ray's time is rayO[slot].w
    int numPrims;                   // walked primitives (1..7)
    int prim[kWalkMaxPrims];
    int topCount[kWalkMaxPrims];    // Node64 records of each walked primitive's tree staged into LDS (a prefix: breadth-first order)
    int stackEntries;               // LDS stack entries per lane (the deepest walked tree's need, or fewer: see overflow)
    uint32_t* overflow;             // [lane of the grid][overflowEntries]: stack entries beyond the LDS ones (null: the LDS stack holds the deepest tree)
    int overflowEntries;
    int refillMin;                  // idle lanes that trigger a refill from the workgroup's range
    int leafMin;                    // lanes waiting at a triangle that trigger the triangle phase
    unsigned long long* prof;       // developer-only (-DTN_WALK_PROF): per-section wave cycles and event counts
};

// Developer-only section timer of k_walk (never in the shipped library): s_memtime deltas per wave, summed into job.prof
//   [0] refill cycles [1] node-phase cycles [2] triangle-phase cycles [3] pop/finish cycles [4] loop overhead cycles
//   [5] iterations [6] refills [7] node-phase runs [8] triangle-phase runs [9] lanes in node phases [10] lanes in triangle
//   phases [11] lanes refilled [12] waves [13] total wave cycles [14] node visits served from LDS
#ifdef TN_WALK_PROF
#define TN_WP_DECL unsigned long long wp[15] = { 0 }; long long wpT = clock64(); const long long wpT0 = wpT;
#define TN_WP_TICK(k) { __builtin_amdgcn_sched_barrier(0); const long long _t = clock64(); wp[k] += (unsigned long long)(_t - wpT); wpT = _t; __builtin_amdgcn_sched_barrier(0); }
#define TN_WP_COUNT(k, v) { wp[k] += (unsigned long long)(v); }
#define TN_WP_FLUSH { wp[12] = 1; wp[13] = (unsigned long long)(clock64() - wpT0); if (lane == 0 && job.prof) for (int k = 0; k < 15; ++k) atomicAdd(job.prof + k, wp[k]); }
#else
#define TN_WP_DECL
#define TN_WP_TICK(k)
#define TN_WP_COUNT(k, v)
#define TN_WP_FLUSH
#endif

// The walked meshes live in HBM by definition: say so to the compiler (a pointer read out of the mesh table is a
// generic one to it, and generic loads are flat_load + a wait on both counters).
typedef float WalkF4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) WalkF4* GlobalF4;
TN_D GlobalF4 as_global(const void* p) { return (GlobalF4)(uintptr_t)p; }

template <class Ptr>
TN_D Node64 load_node_from(Ptr nodes, uint32_t idx)
{
    Ptr p = nodes + (size_t)idx*4;
    const WalkF4 a = p[0], b = p[1], c = p[2], d = p[3];
    Node64 n;
    n.lminx = a.x; n.lminy = a.y; n.lminz = a.z; n.lmaxx = a.w;
    n.lmaxy = b.x; n.lmaxz = b.y; n.rminx = b.z; n.rminy = b.w;
    n.rminz = c.x; n.rmaxx = c.y; n.rmaxy = c.z; n.rmaxz = c.w;
    n.left = __float_as_uint(d.x);
    n.right = __float_as_uint(d.y);
    return n;
}

// Walk mode (template parameter MODE of k_walk; launch_walk picks it):
//   kWalkSingle  ONE walked primitive (the host knows): tree, triangles and the staged top are the same for every lane -- kernel-argument
//                scalars instead of five per-lane registers.
// (Round 4's kWalkPairs -- a node over two one-triangle leaves fetched as ONE 128-B record in the triangle phase -- was built in three
// versions, bit-identical, and never paid: 10.8-10.9 ms without, 10.96-11.02 with on the 524k-triangle config, glass 6.5 -> 8.8 ms;
// profiles/r04_a..c_ab_walk_pairs*.md have the numbers, `git log -- tinsel_amd/csrc/tn_walk.h` the code.)
// (Round 5 built two more modes, both bit-identical under the whole GPU suite, neither kept: the WHOLE of a small mesh in LDS -- glass.tin's
// sphere + cube, 126 KB, one workgroup per CU: k_walk 6.54 ms against 6.70, 38 % of the lanes active either way: what a ray costs there is its
// refill and its record, not its node fetches -- and TWO tree levels per 128-B record (a node's child boxes followed by those of its larger
// internal child: a third fewer visits, each of twice the bytes: 14.0 ms against 10.7 on the 524k-triangle config -- the walk is bound by the
// BYTES it pulls through a CU's L1, in 64-B sectors, not by lines or round trips).  profiles/r05_b_ab_glass_lds_mesh.md, r05_c_ab_walk_fat.md.)
constexpr int kWalkSingle = 2;

// The closest hit's normal for its record: n*sign with n = Cross(b - a, c - a) as IntersectRayTriTwoSided forms it (intersection.h:122-124),
// computed again from the triangle where the record is written -- the lane's next refill, whose chain of dependent loads hides the
// 48-B fetch -- instead of carried in three registers from the hit to the end of the walk (same vertices, same operations: same bits).
TN_D V3 hit_normal(GlobalF4 tris, int tri, float sign)
{
    GlobalF4 tp = tris + (size_t)tri*3;
    const WalkF4 ta = tp[0], tb = tp[1], tc = tp[2];
    const V3 a(ta.x, ta.y, ta.z), b(tb.x, tb.y, tb.z), c(tc.x, tc.y, tc.z);
    return cross(b - a, c - a)*sign;
}

// LDS of one workgroup: [stackEntries][BLOCK] stack words, [kWalkLaneRows][BLOCK] per-lane words that are touched once or twice per
// RAY and have no business in a register of a 64-VGPR kernel (row 0: the ray's record index, row 1: a shadow ray's stop distance),
// then 16 control words, then the staged tree tops.
constexpr int kWalkCtlWords = 16;
constexpr int kWalkLaneRows = 2;

template <int BLOCK, int WAVES, int MODE = 0>
__global__ __launch_bounds__(BLOCK, WAVES) void k_walk(DevScene sc, WalkJob job)
{
    constexpr bool SINGLE = (MODE & kWalkSingle) != 0;
    static_assert(SINGLE, "k_walk is instantiated for ONE walked primitive only: several go through k_walk_rays (below)");
    constexpr uint32_t kAtLeaf = kLeafBit;      // refs that wait for the triangle phase
    extern __shared__ __attribute__((aligned(16))) uint32_t s_walk[];
    uint32_t* const stack = s_walk + threadIdx.x;               // this lane's column: entry i at stack[i*BLOCK]
    // a SHORT LDS stack leaves room for a second workgroup per CU (8 waves per SIMD at 64 VGPRs): the rare entries beyond it
    // live in HBM, a column per lane of the grid
    uint32_t* const spill = job.overflow ? job.overflow + ((size_t)blockIdx.x*BLOCK + threadIdx.x)*(size_t)job.overflowEntries : nullptr;
    const int ldsEntries = job.stackEntries;
    uint32_t* const s_item = s_walk + job.stackEntries*BLOCK + threadIdx.x;                 // kNoItem: no finished ray's record waits in this lane's registers
    float* const s_stop = reinterpret_cast<float*>(s_walk + (job.stackEntries + 1)*BLOCK + threadIdx.x);    // shadow rays: an accepted hit closer than this ends the walk
    uint32_t* const s_ctl = s_walk + (job.stackEntries + kWalkLaneRows)*BLOCK;              // [0] the workgroup's cursor
    WalkF4* const s_top = reinterpret_cast<WalkF4*>(s_ctl + kWalkCtlWords);

    const int lane = (int)__lane_id();
    const uint32_t Kx = job.mixed == 1 ? (uint32_t)job.neePerPath + 1u : job.neePerPath > 0 ? (uint32_t)job.neePerPath : 1u;
    const uint32_t Kb = (uint32_t)job.numPrims;
    const uint32_t per = Kx*Kb;                                 // work items per queued slot
    const uint32_t total = (*job.frontCount)*per;
    // my/per and rem/Kb below: both divisors are wave-uniform, so the reciprocals live in SGPRs (the compiler's own expansion kept two
    // VGPR reciprocals across the loop, spilled them, and reloaded them in every refill behind an s_waitcnt vmcnt(0) -- i.e. behind the
    // finished rays' record stores).  q' = mulhi(n, floor((2^32-1)/d)) is q or q - 1 for n < 2^32: one correction.
    const uint32_t perM = (uint32_t)__builtin_amdgcn_readfirstlane((int)(0xffffffffu/per));
    const uint32_t KbM = (uint32_t)__builtin_amdgcn_readfirstlane((int)(0xffffffffu/Kb));

    // static ranges: workgroup b -> the b-th contiguous piece of the items; its waves share it through an LDS cursor
    const uint32_t chunk = (total + gridDim.x - 1u)/gridDim.x;
    const uint32_t bbeg = blockIdx.x*chunk < total ? blockIdx.x*chunk : total;
    const uint32_t end = (bbeg + chunk) < total ? (bbeg + chunk) : total;
    if (threadIdx.x == 0)
        s_ctl[0] = bbeg;
    *s_item = 0xffffffffu;
    *s_stop = -kFltMax;                 // (never, for extension rays)

    // stage the tops of the walked trees (a workgroup with nothing to do skips it)
    if (bbeg < end)
    {
        uint32_t base = 0;
#pragma unroll 1
        for (int kb = 0; kb < job.numPrims; ++kb)
        {
            int primIndex = job.prim[0], n = job.topCount[0];
#pragma unroll
            for (int q = 1; q < kWalkMaxPrims; ++q)
                if (q == kb) { primIndex = job.prim[q]; n = job.topCount[q]; }
            if (n > 0)
            {
                const Prim64 p = load_prim(sc.prims, primIndex);
                GlobalF4 src = as_global(sc.meshes[p.mesh].nodes);
                WalkF4* dst = s_top + (size_t)base*4;
                for (uint32_t i = threadIdx.x; i < (uint32_t)n*4u; i += BLOCK)
                    dst[i] = src[i];
                base += (uint32_t)n;
            }
        }
    }
    __syncthreads();

    // ONE walked primitive (the common case): its record, leaf box and mesh table entry are wave-uniform -- read once, not
    // behind every refill's ray fetch (the refill is a chain of dependent loads: queue -> slot -> ray -> [box, primitive,
    // mesh]; the last three were a third of it)
    const bool single = SINGLE || job.numPrims == 1;
    Prim64 prim0 = load_prim(sc.prims, job.prim[0]);
    float4 box0a, box0b;
    {
        const float4* bp = reinterpret_cast<const float4*>(sc.primBoxes + job.prim[0]);
        box0a = bp[0]; box0b = bp[1];
    }
    const DevMesh* mesh0p = sc.meshes + prim0.mesh;
    GlobalF4 mesh0nodes = as_global(mesh0p->nodes), mesh0tris = as_global(mesh0p->tris);
    const uint32_t mesh0root = mesh0p->root;
    const uint32_t top0N = (uint32_t)job.topCount[0];

    // per-lane walk state
    // (no flags: a lane is ACTIVE iff ref != kNoNode; a finished ray's record is PENDING in its registers iff *s_item != kNoItem)
    constexpr uint32_t kNoItem = 0xffffffffu;
    uint32_t ref = kNoNode;
#define active (ref != kNoNode)
    int sp = 0;
    V3 o, d, rcp;
    float closestT = kFltMax;
    float hv = 0.0f, hw = 0.0f;         // (u = 1 - v - w is recomputed where the record is written: IntersectRayTriTwoSided's own expression)
    int htri = -1;
    float hsign = 0.0f;                 // the hit's `sign` (IntersectRayTriTwoSided's d): its normal n*sign is formed where the record is written
    GlobalF4 mnodes = nullptr;
    GlobalF4 mtris = nullptr;
    uint32_t topBase = 0, topN = 0;     // this lane's tree: refs < topN are staged at s_top[(topBase + ref)*4 ..]
    bool finiteAll = true;              // wave-uniform: every active lane's 1/d is finite
    bool exhausted = bbeg >= end;       // wave-uniform: the workgroup's range has been handed out
    TN_WP_DECL

    for (;;)
    {
        TN_WP_TICK(4)
        TN_WP_COUNT(5, 1)
        // ---- refill: idle lanes take the next items of the workgroup's range ---------------------------------
        const unsigned long long idleMask = __ballot(!active);
        const int nIdle = __popcll(idleMask);
        if (!exhausted && nIdle >= job.refillMin)
        {
            TN_WP_COUNT(6, 1)
            TN_WP_COUNT(11, nIdle)
            uint32_t cur = 0;
            if (lane == 0)
                cur = atomicAdd(&s_ctl[0], (uint32_t)nIdle);        // LDS atomic: one per refill
            cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur);
            exhausted = cur + (uint32_t)nIdle >= end;
            if (!active)
            {
                // A finished ray's record is written HERE, next to the loads of the lane's next ray, not where the ray
                // ends: stores count against vmcnt like loads on gfx950, so a 32-B record on its way to HBM would sit in
                // front of every node fetch the wave waits for (measured: node phases of 1900-2800 cycles, 780 when
                // nothing but LDS reads was outstanding).
                const uint32_t item = *s_item;
                if (item != kNoItem)
                {
                    float4* out = job.rec + (size_t)item*2;
                    out[0] = make_float4(closestT, 1.0f - hv - hw, hv, hw);
                    if (closestT < kFltMax)
                    {
                        const V3 hn = hit_normal(SINGLE ? mesh0tris : mtris, htri, hsign);
                        out[1] = make_float4(hn.x, hn.y, hn.z, __int_as_float(htri));
                    }
                    *s_item = kNoItem;
                }
                // (set bits of the idle mask below this lane: v_mbcnt, no per-lane 64-bit mask kept in registers)
                const uint32_t my = cur + __builtin_amdgcn_mbcnt_hi((uint32_t)(idleMask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idleMask, 0u));
                if (my < end)
                {
                    uint32_t qi = __umulhi(my, perM), rem = my - qi*per;
                    if (rem >= per) { ++qi; rem -= per; }
                    uint32_t k = __umulhi(rem, KbM), kb = rem - k*Kb;
                    if (kb >= Kb) { ++k; kb -= Kb; }
                    const uint32_t slot = job.queue[qi];
                    const uint32_t recAt = slot*per + rem;      // records are indexed by position, like everything the scan kernels read

                    float4 ro, rd;
                    float time;
                    if (job.neePerPath > 0 && !(job.mixed == 1 && k == (uint32_t)job.neePerPath))
                    {
                        const float4* np = job.nee + (size_t)(k*2u)*job.neeStride + slot;
                        ro = np[0]; rd = np[job.neeStride];
                        time = job.mixed ? job.rayO[sidx(slot)].w : job.neeTime[slot];
                        *s_stop = shadow_stop(ro.w);    // the record's .w is the sample's distance (< 0: probe sample)
                    }
                    else
                    {
                        ro = job.rayO[sidx(slot)]; rd = job.rayD[sidx(slot)];
                        time = ro.w;
                        if (job.mixed)
                            *s_stop = -kFltMax;         // (the lane's last ray may have been a shadow ray)
                    }
                    const V3 wo(ro.x, ro.y, ro.z), wd(rd.x, rd.y, rd.z);
                    const bool noRay = job.mixed == 1 && rd.x == 0.0f && rd.y == 0.0f && rd.z == 0.0f;      // (a path without an extension ray)

                    int index = job.prim[0];
                    uint32_t tb = 0, tn = (uint32_t)job.topCount[0], run = (uint32_t)job.topCount[0];
                    if (!SINGLE)
                    {
#pragma unroll
                        for (int q = 1; q < kWalkMaxPrims; ++q)
                        {
                            if ((uint32_t)q == kb)
                            {
                                index = job.prim[q];
                                tb = run;
                                tn = (uint32_t)job.topCount[q];
                            }
                            run += (uint32_t)job.topCount[q];
                        }
                    }

                    // the leaf-box test of the scan (trace_flat / the scene BVH walk): same function, same operands
                    float4 b0 = box0a, b1 = box0b;
                    if (!single)
                    {
                        const float4* bp = reinterpret_cast<const float4*>(sc.primBoxes + index);
                        b0 = bp[0]; b1 = bp[1];
                    }
                    const V3 wrcp = rcp3_cr(wd);
                    float tbox;
                    bool enters = true;         // rays the scan does not box-test (ray_sane) are walked unconditionally
                    if (__float_as_uint(b1.z) == 0u && ray_sane(wo))
                        enters = ray_aabb(wo, wrcp, b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, tbox);

                    if (noRay)
                    {
                    }
                    else if (!enters)
                    {
                        job.rec[(size_t)recAt*2] = make_float4(kFltMax, 0.0f, 0.0f, 0.0f);
                    }
                    else
                    {
                        // PrimitiveIntersect's mesh branch up to IntersectRayMesh (intersection.h:977-990)
                        Prim64 p = prim0;
                        if (!single)
                            p = load_prim(sc.prims, index);
                        const Xform x = prim_pose(sc, p, time);
                        pose_inv_ray(p, x, wo, wd, o, d, rcp, wrcp, true);
                        if (SINGLE)
                            ref = mesh0root;
                        else if (single)
                        {
                            mnodes = mesh0nodes;
                            mtris = mesh0tris;
                            ref = mesh0root;
                        }
                        else
                        {
                            const DevMesh* m = sc.meshes + p.mesh;
                            mnodes = as_global(m->nodes);
                            mtris = as_global(m->tris);
                            ref = m->root;
                        }
                        if (!SINGLE)
                        {
                            topBase = tb;
                            topN = tn;
                        }
                        sp = 0;
                        closestT = kFltMax;
                        htri = -1;
                        *s_item = recAt;
                    }
                }
            }
            finiteAll = __all(!active || (finite_bits(rcp.x) && finite_bits(rcp.y) && finite_bits(rcp.z) &&
                                          finite_bits(o.x) && finite_bits(o.y) && finite_bits(o.z)));
            TN_WP_TICK(0)
        }

        if (__ballot(active) == 0ull)
        {
            if (exhausted)
                break;
            continue;
        }

        bool pop = false;
#ifdef TN_WALK_PROF
        { const unsigned long long nm = __ballot(active && !(ref & kAtLeaf)); if (nm) { TN_WP_COUNT(7, 1) TN_WP_COUNT(9, __popcll(nm)) }
          TN_WP_COUNT(14, __popcll(__ballot(active && !(ref & kAtLeaf) && ref < (SINGLE ? top0N : topN)))) }
        TN_WP_TICK(4)
#endif

        // ---- node phase: lanes at an internal node -------------------------------------------------------------
        if (active && !(ref & kAtLeaf))
        {
            Node64 nd;
            if (ref < (SINGLE ? top0N : topN))
                nd = load_node_from((const WalkF4*)s_top, SINGLE ? ref : topBase + ref);    // 4 x ds_read_b128
            else
                nd = load_node_from(SINGLE ? mesh0nodes : mnodes, ref);                     // 4 x global_load_dwordx4
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
            hL = hL && tL < closestT;       // `tLeft < tmax`, tmax == closestT after every leaf (intersection.h:701-702)
            hR = hR && tR < closestT;
            const uint32_t refL = nd.left, refR = nd.right;

            if (hL && hR)
            {
                // the reference pushes far then near and pops near: the near child continues in a register
                const bool leftNear = tL < tR;
                const uint32_t far = leftNear ? refR : refL;
                if (sp < ldsEntries)
                    stack[sp*BLOCK] = far;
                else
                    spill[sp - ldsEntries] = far;
                ++sp;
                ref = leftNear ? refL : refR;
            }
            else if (hL)
                ref = refL;
            else if (hR)
                ref = refR;
            else
                pop = true;
        }

        TN_WP_TICK(1)
        // ---- triangle phase: once enough lanes wait at a leaf (or nobody has a node to visit) -----------------
        const bool atLeaf = active && !pop && (ref & kAtLeaf);
        const unsigned long long leafMask = __ballot(atLeaf);
        if (leafMask != 0ull && (__popcll(leafMask) >= job.leafMin || __ballot(active && !pop && !atLeaf) == 0ull))
        {
            TN_WP_COUNT(8, 1)
            TN_WP_COUNT(10, __popcll(leafMask))
            if (atLeaf)
            {
                // ONE round trip per phase: every lane requests its leaf's Tri48 before anybody waits
                const uint32_t idx = ref & ~kLeafBit;
                GlobalF4 tp = (SINGLE ? mesh0tris : mtris) + (size_t)idx*3;
                const WalkF4 q3 = tp[0], q4 = tp[1], q5 = tp[2];
                bool any = false;
                {
                    float t, u, v, w, sign;
                    V3 n;
                    if (ray_tri(o, d, V3(q3.x, q3.y, q3.z), V3(q4.x, q4.y, q4.z), V3(q5.x, q5.y, q5.z), t, u, v, w, sign, n))
                    {
                        if (t > 0.0f && t < closestT)
                        {
                            any = true;
                            closestT = t;
                            hv = v; hw = w;
                            htri = (int)idx;
                            hsign = sign;
                        }
                    }
                }
                if (any && closestT < *s_stop)
                    sp = 0;             // shadow ray decided (shadow_stop, tn_isect.h): drop what is left on the stack
                pop = true;
            }
        }

        TN_WP_TICK(2)
        // ---- next entry, or the ray is done ---------------------------------------------------------------------
        if (pop)
        {
            if (sp > 0)
            {
                --sp;
                ref = sp < ldsEntries ? stack[sp*BLOCK] : spill[sp - ldsEntries];
            }
            else
            {
                ref = kNoNode;          // done; the record stays in registers until the lane's next refill (*s_item says which)
            }
        }
        TN_WP_TICK(3)
    }
    if (const uint32_t item = *s_item; item != kNoItem)
    {
        float4* out = job.rec + (size_t)item*2;
        out[0] = make_float4(closestT, 1.0f - hv - hw, hv, hw);
        if (closestT < kFltMax)
        {
            const V3 hn = hit_normal(SINGLE ? mesh0tris : mtris, htri, hsign);
            out[1] = make_float4(hn.x, hn.y, hn.z, __int_as_float(htri));
        }
    }
    TN_WP_FLUSH
#undef active
}

// ---------------------------------------------------------------------------
// k_walk_rays: SEVERAL walked primitives.  The same walk; a work item is a RAY (slot, k): the lane that takes it tests all walked primitives'
// leaf boxes once -- the records are wave-uniform -- and walks the ones the ray enters one after the other (`pend`: a bit per primitive still
// to visit; the ray itself parked in an LDS row as position | k << 27).  Until round 5 an item was a (ray, primitive) pair through k_walk's
// MODE 0: glass fetched every ray twice, the reference's table.tin -- seven walked meshes -- seven times, to find most of the boxes missed
// (table.tin 690 -> 896 Msamples/s, transmission.tin 654 -> 815, glass k_walk 6.67 -> 5.91 ms: profiles/r05_e_ab_walk_by_ray.md).  A kernel of
// its own, not a mode of k_walk: written as one template, the one-primitive kernel of the 524k-triangle config came out 4-8 % slower with an
// identical node phase (profiles/r05_g_ab_walk_single_refill.md) -- its source is left exactly as rounds 3-4 tuned it.
constexpr int kWalkRayRows = 3;         // k_walk's two per-lane LDS rows + the ray
// The walked primitives' records in LDS (round 6): a refill used to be a chain of FIVE dependent global round trips -- queue -> ray -> Prim64 ->
// mesh table entry -> (tree pointers, root) -- and took a third of the kernel's cycles on glass (63 k cycles per refill against 5 k per node
// phase, profiles/r05_z_walk_profile.txt).  The last three are the same few records for every ray: one entry per walked primitive is staged at
// the kernel's start (the Prim64, then the tree's nodes / triangles pointers and its root ref), and a refill reads its primitive's entry from LDS.
constexpr int kWalkPrimWords = 24;      // 16 (Prim64) + 2 (nodes) + 2 (tris) + 1 (root) + 3 (padding: entries stay 16-B aligned)

template <int BLOCK, int WAVES>
__global__ __launch_bounds__(BLOCK, WAVES) void k_walk_rays(DevScene sc, WalkJob job)
{
    constexpr uint32_t kAtLeaf = kLeafBit;      // refs that wait for the triangle phase
    extern __shared__ __attribute__((aligned(16))) uint32_t s_walk[];
    uint32_t* const stack = s_walk + threadIdx.x;               // this lane's column: entry i at stack[i*BLOCK]
    // a SHORT LDS stack leaves room for a second workgroup per CU (8 waves per SIMD at 64 VGPRs): the rare entries beyond it
    // live in HBM, a column per lane of the grid
    uint32_t* const spill = job.overflow ? job.overflow + ((size_t)blockIdx.x*BLOCK + threadIdx.x)*(size_t)job.overflowEntries : nullptr;
    const int ldsEntries = job.stackEntries;
    uint32_t* const s_item = s_walk + job.stackEntries*BLOCK + threadIdx.x;                 // kNoItem: no finished ray's record waits in this lane's registers
    float* const s_stop = reinterpret_cast<float*>(s_walk + (job.stackEntries + 1)*BLOCK + threadIdx.x);    // shadow rays: an accepted hit closer than this ends the walk
    uint32_t* const s_ray = s_walk + (job.stackEntries + 2)*BLOCK + threadIdx.x;            // several walked primitives: the lane's ray (slot | k << 27: the host checks the ranges) while primitives are left
    uint32_t* const s_ctl = s_walk + (job.stackEntries + kWalkRayRows)*BLOCK;               // [0] the workgroup's cursor
    uint32_t* const s_prim = s_ctl + kWalkCtlWords;                                          // [walked primitive][kWalkPrimWords]
    WalkF4* const s_top = reinterpret_cast<WalkF4*>(s_prim + kWalkMaxPrims*kWalkPrimWords);

    const int lane = (int)__lane_id();
    const uint32_t Kx = job.mixed == 1 ? (uint32_t)job.neePerPath + 1u : job.neePerPath > 0 ? (uint32_t)job.neePerPath : 1u;
    const uint32_t Kb = (uint32_t)job.numPrims;
    // A work item is a RAY (slot, k).  With several walked primitives the lane that takes it tests all their leaf boxes once -- the records
    // are wave-uniform -- and walks the ones the ray enters one after the other (`pend`: a bit per primitive still to visit).  (Until round 5
    // an item was a (ray, primitive) pair: glass fetched every ray twice, the reference's table.tin -- seven walked meshes -- seven times,
    // to find five or six of the seven boxes missed.)
    const uint32_t per = Kx;                                    // work items per queued slot
    const uint32_t total = (*job.frontCount)*per;
    // my/per and rem/Kb below: both divisors are wave-uniform, so the reciprocals live in SGPRs (the compiler's own expansion kept two
    // VGPR reciprocals across the loop, spilled them, and reloaded them in every refill behind an s_waitcnt vmcnt(0) -- i.e. behind the
    // finished rays' record stores).  q' = mulhi(n, floor((2^32-1)/d)) is q or q - 1 for n < 2^32: one correction.
    const uint32_t perM = (uint32_t)__builtin_amdgcn_readfirstlane((int)(0xffffffffu/per));

    // static ranges: workgroup b -> the b-th contiguous piece of the items; its waves share it through an LDS cursor
    const uint32_t chunk = (total + gridDim.x - 1u)/gridDim.x;
    const uint32_t bbeg = blockIdx.x*chunk < total ? blockIdx.x*chunk : total;
    const uint32_t end = (bbeg + chunk) < total ? (bbeg + chunk) : total;
    if (threadIdx.x == 0)
        s_ctl[0] = bbeg;
    *s_item = 0xffffffffu;
    *s_stop = -kFltMax;                 // (never, for extension rays)

    // one entry per walked primitive: what a refill needs of it (see kWalkPrimWords)
    if (bbeg < end && threadIdx.x < (uint32_t)job.numPrims)
    {
        int primIndex = job.prim[0];
#pragma unroll
        for (int q = 1; q < kWalkMaxPrims; ++q)
            if ((uint32_t)q == threadIdx.x) primIndex = job.prim[q];
        const float4* pp = reinterpret_cast<const float4*>(sc.prims + primIndex);
        float4* dst = reinterpret_cast<float4*>(s_prim + threadIdx.x*kWalkPrimWords);
        const float4 r0 = pp[0], r1 = pp[1], r2 = pp[2], r3 = pp[3];
        dst[0] = r0; dst[1] = r1; dst[2] = r2; dst[3] = r3;
        const DevMesh* m = sc.meshes + __float_as_uint(r3.z);
        const unsigned long long pn = (unsigned long long)(uintptr_t)m->nodes, pt = (unsigned long long)(uintptr_t)m->tris;
        uint32_t* e = s_prim + threadIdx.x*kWalkPrimWords + 16;
        e[0] = (uint32_t)pn; e[1] = (uint32_t)(pn >> 32); e[2] = (uint32_t)pt; e[3] = (uint32_t)(pt >> 32); e[4] = m->root;
    }
    // stage the tops of the walked trees (a workgroup with nothing to do skips it)
    if (bbeg < end)
    {
        uint32_t base = 0;
#pragma unroll 1
        for (int kb = 0; kb < job.numPrims; ++kb)
        {
            int primIndex = job.prim[0], n = job.topCount[0];
#pragma unroll
            for (int q = 1; q < kWalkMaxPrims; ++q)
                if (q == kb) { primIndex = job.prim[q]; n = job.topCount[q]; }
            if (n > 0)
            {
                const Prim64 p = load_prim(sc.prims, primIndex);
                GlobalF4 src = as_global(sc.meshes[p.mesh].nodes);
                WalkF4* dst = s_top + (size_t)base*4;
                for (uint32_t i = threadIdx.x; i < (uint32_t)n*4u; i += BLOCK)
                    dst[i] = src[i];
                base += (uint32_t)n;
            }
        }
    }
    __syncthreads();

    // per-lane walk state
    // (no flags: a lane is ACTIVE iff ref != kNoNode; a finished ray's record is PENDING in its registers iff *s_item != kNoItem)
    constexpr uint32_t kNoItem = 0xffffffffu;
    uint32_t ref = kNoNode;
#define active (ref != kNoNode)
    int sp = 0;
    V3 o, d, rcp;
    float closestT = kFltMax;
    float hv = 0.0f, hw = 0.0f;         // (u = 1 - v - w is recomputed where the record is written: IntersectRayTriTwoSided's own expression)
    int htri = -1;
    float hsign = 0.0f;                 // the hit's `sign` (IntersectRayTriTwoSided's d): its normal n*sign is formed where the record is written
    GlobalF4 mnodes = nullptr;
    GlobalF4 mtris = nullptr;
    uint32_t topBase = 0, topN = 0;     // this lane's tree: refs < topN are staged at s_top[(topBase + ref)*4 ..]
    uint32_t pend = 0;                  // (several walked primitives) bit kb: this lane's ray enters walked primitive kb's box and has not walked it yet
    bool finiteAll = true;              // wave-uniform: every active lane's 1/d is finite
    bool exhausted = bbeg >= end;       // wave-uniform: the workgroup's range has been handed out
    TN_WP_DECL

    for (;;)
    {
        TN_WP_TICK(4)
        TN_WP_COUNT(5, 1)
        // ---- refill: idle lanes take the next items of the workgroup's range ---------------------------------
        const unsigned long long idleMask = __ballot(!active);
        const int nIdle = __popcll(idleMask);
        // idle lanes whose ray has primitives left go on with it; the others take new items
        const unsigned long long newMask = __ballot(!active && pend == 0u);
        const bool anyCont = newMask != idleMask;
        if ((!exhausted || anyCont) && nIdle >= job.refillMin)
        {
            TN_WP_COUNT(6, 1)
            TN_WP_COUNT(11, nIdle)
            const int nNew = __popcll(newMask);
            uint32_t cur = end;
            if (!exhausted)
            {
                cur = 0;
                if (lane == 0)
                    cur = atomicAdd(&s_ctl[0], (uint32_t)nNew);         // LDS atomic: one per refill
                cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur);
                exhausted = cur + (uint32_t)nNew >= end;
            }
            if (!active)
            {
                // A finished ray's record is written HERE, next to the loads of the lane's next ray, not where the ray
                // ends: stores count against vmcnt like loads on gfx950, so a 32-B record on its way to HBM would sit in
                // front of every node fetch the wave waits for (measured: node phases of 1900-2800 cycles, 780 when
                // nothing but LDS reads was outstanding).
                const uint32_t item = *s_item;
                if (item != kNoItem)
                {
                    float4* out = job.rec + (size_t)item*2;
                    out[0] = make_float4(closestT, 1.0f - hv - hw, hv, hw);
                    if (closestT < kFltMax)
                    {
                        const V3 hn = hit_normal(mtris, htri, hsign);
                        out[1] = make_float4(hn.x, hn.y, hn.z, __int_as_float(htri));
                    }
                    *s_item = kNoItem;
                }
                // the lane's ray: the one it still has primitives for, or the next item of the range
                // (set bits of the mask below this lane: v_mbcnt, no per-lane 64-bit mask kept in registers)
                const bool cont = pend != 0u;
                uint32_t slot = 0, k = 0;
                bool have = cont;
                if (cont)
                {
                    const uint32_t sk = *s_ray;
                    slot = sk & 0x07ffffffu;
                    k = sk >> 27;
                }
                else
                {
                    const uint32_t my = cur + __builtin_amdgcn_mbcnt_hi((uint32_t)(newMask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)newMask, 0u));
                    if (my < end)
                    {
                        uint32_t qi = __umulhi(my, perM);
                        k = my - qi*per;
                        if (k >= per) { ++qi; k -= per; }
                        slot = job.queue[qi];
                        have = true;
                        *s_ray = slot | (k << 27);
                    }
                }
                if (have)
                {
                    float4 ro, rd;
                    float time;
                    if (job.neePerPath > 0 && !(job.mixed == 1 && k == (uint32_t)job.neePerPath))
                    {
                        const float4* np = job.nee + (size_t)(k*2u)*job.neeStride + slot;
                        ro = np[0]; rd = np[job.neeStride];
                        time = job.mixed ? job.rayO[sidx(slot)].w : job.neeTime[slot];
                        *s_stop = shadow_stop(ro.w);    // the record's .w is the sample's distance (< 0: probe sample)
                    }
                    else
                    {
                        ro = job.rayO[sidx(slot)]; rd = job.rayD[sidx(slot)];
                        time = ro.w;
                        if (job.mixed)
                            *s_stop = -kFltMax;         // (the lane's last ray may have been a shadow ray)
                    }
                    const V3 wo(ro.x, ro.y, ro.z), wd(rd.x, rd.y, rd.z);
                    const V3 wrcp = rcp3_cr(wd);
                    const bool noRay = job.mixed == 1 && rd.x == 0.0f && rd.y == 0.0f && rd.z == 0.0f;      // (a path without an extension ray)

                    // which walked primitives does the ray enter?  The leaf-box test of the scan (trace_flat / the scene BVH walk): same
                    // function, same operands; rays the scan does not box-test (ray_sane) are walked unconditionally.  A primitive whose
                    // box the ray misses gets no record: the scan kernels read one only behind the same test.
                    {
                        if (!cont)
                        {
                            const bool sane = ray_sane(wo);
                            pend = 0u;
#pragma unroll
                            for (int q = 0; q < kWalkMaxPrims; ++q)
                            {
                                if (q < job.numPrims)
                                {
                                    const float4* bp = reinterpret_cast<const float4*>(sc.primBoxes + job.prim[q]);     // (wave-uniform)
                                    const float4 b0 = bp[0], b1 = bp[1];
                                    float tbox;
                                    bool in = !noRay;
                                    if (__float_as_uint(b1.z) == 0u && sane && !noRay)
                                        in = ray_aabb(wo, wrcp, b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, tbox);
                                    pend |= in ? (1u << q) : 0u;
                                }
                            }
                        }
                    }

                    if (pend != 0u)
                    {
                        // the primitive to walk now: the lowest one left
                        const uint32_t kb = (uint32_t)__builtin_ctz(pend);
                        pend &= pend - 1u;
                        uint32_t tb = 0, tn = (uint32_t)job.topCount[0], run = (uint32_t)job.topCount[0];
                        {
#pragma unroll
                            for (int q = 1; q < kWalkMaxPrims; ++q)
                            {
                                if ((uint32_t)q == kb)
                                {
                                    tb = run;
                                    tn = (uint32_t)job.topCount[q];
                                }
                                run += (uint32_t)job.topCount[q];
                            }
                        }
                        // PrimitiveIntersect's mesh branch up to IntersectRayMesh (intersection.h:977-990); the primitive's record and its
                        // tree from the entry staged in LDS
                        const uint32_t* const entry = s_prim + kb*kWalkPrimWords;
                        const Prim64 p = load_prim(reinterpret_cast<const Prim64*>(entry), 0);
                        const Xform x = prim_pose(sc, p, time);
                        pose_inv_ray(p, x, wo, wd, o, d, rcp, wrcp, true);
                        {
                            const uint4 tr = *reinterpret_cast<const uint4*>(entry + 16);
                            mnodes = (GlobalF4)(uintptr_t)((unsigned long long)tr.x | ((unsigned long long)tr.y << 32));
                            mtris = (GlobalF4)(uintptr_t)((unsigned long long)tr.z | ((unsigned long long)tr.w << 32));
                            ref = entry[20];
                        }
                        topBase = tb;
                        topN = tn;
                        sp = 0;
                        closestT = kFltMax;
                        htri = -1;
                        *s_item = (slot*per + k)*Kb + kb;       // records are indexed by position, like everything the scan kernels read
                    }
                }
            }
            finiteAll = __all(!active || (finite_bits(rcp.x) && finite_bits(rcp.y) && finite_bits(rcp.z) &&
                                          finite_bits(o.x) && finite_bits(o.y) && finite_bits(o.z)));
            TN_WP_TICK(0)
        }

        if (__ballot(active) == 0ull)
        {
            // nobody walks: done when the range is handed out and no lane has a primitive left; else the idle lanes refill at the next turn
            if (exhausted && __ballot(pend != 0u) == 0ull)
                break;
            continue;
        }

        bool pop = false;
#ifdef TN_WALK_PROF
        { const unsigned long long nm = __ballot(active && !(ref & kAtLeaf)); if (nm) { TN_WP_COUNT(7, 1) TN_WP_COUNT(9, __popcll(nm)) }
          TN_WP_COUNT(14, __popcll(__ballot(active && !(ref & kAtLeaf) && ref < topN))) }
        TN_WP_TICK(4)
#endif

        // ---- node phase: lanes at an internal node -------------------------------------------------------------
        if (active && !(ref & kAtLeaf))
        {
            Node64 nd;
            if (ref < topN)
                nd = load_node_from((const WalkF4*)s_top, topBase + ref);    // 4 x ds_read_b128
            else
                nd = load_node_from(mnodes, ref);                     // 4 x global_load_dwordx4
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
            hL = hL && tL < closestT;       // `tLeft < tmax`, tmax == closestT after every leaf (intersection.h:701-702)
            hR = hR && tR < closestT;
            const uint32_t refL = nd.left, refR = nd.right;

            if (hL && hR)
            {
                // the reference pushes far then near and pops near: the near child continues in a register
                const bool leftNear = tL < tR;
                const uint32_t far = leftNear ? refR : refL;
                if (sp < ldsEntries)
                    stack[sp*BLOCK] = far;
                else
                    spill[sp - ldsEntries] = far;
                ++sp;
                ref = leftNear ? refL : refR;
            }
            else if (hL)
                ref = refL;
            else if (hR)
                ref = refR;
            else
                pop = true;
        }

        TN_WP_TICK(1)
        // ---- triangle phase: once enough lanes wait at a leaf (or nobody has a node to visit) -----------------
        const bool atLeaf = active && !pop && (ref & kAtLeaf);
        const unsigned long long leafMask = __ballot(atLeaf);
        if (leafMask != 0ull && (__popcll(leafMask) >= job.leafMin || __ballot(active && !pop && !atLeaf) == 0ull))
        {
            TN_WP_COUNT(8, 1)
            TN_WP_COUNT(10, __popcll(leafMask))
            if (atLeaf)
            {
                // ONE round trip per phase: every lane requests its leaf's Tri48 before anybody waits
                const uint32_t idx = ref & ~kLeafBit;
                GlobalF4 tp = mtris + (size_t)idx*3;
                const WalkF4 q3 = tp[0], q4 = tp[1], q5 = tp[2];
                bool any = false;
                {
                    float t, u, v, w, sign;
                    V3 n;
                    if (ray_tri(o, d, V3(q3.x, q3.y, q3.z), V3(q4.x, q4.y, q4.z), V3(q5.x, q5.y, q5.z), t, u, v, w, sign, n))
                    {
                        if (t > 0.0f && t < closestT)
                        {
                            any = true;
                            closestT = t;
                            hv = v; hw = w;
                            htri = (int)idx;
                            hsign = sign;
                        }
                    }
                }
                if (any && closestT < *s_stop)
                    sp = 0;             // shadow ray decided (shadow_stop, tn_isect.h): drop what is left on the stack
                pop = true;
            }
        }

        TN_WP_TICK(2)
        // ---- next entry, or the ray is done ---------------------------------------------------------------------
        if (pop)
        {
            if (sp > 0)
            {
                --sp;
                ref = sp < ldsEntries ? stack[sp*BLOCK] : spill[sp - ldsEntries];
            }
            else
            {
                ref = kNoNode;          // done; the record stays in registers until the lane's next refill (*s_item says which)
            }
        }
        TN_WP_TICK(3)
    }
    if (const uint32_t item = *s_item; item != kNoItem)
    {
        float4* out = job.rec + (size_t)item*2;
        out[0] = make_float4(closestT, 1.0f - hv - hw, hv, hw);
        if (closestT < kFltMax)
        {
            const V3 hn = hit_normal(mtris, htri, hsign);
            out[1] = make_float4(hn.x, hn.y, hn.z, __int_as_float(htri));
        }
    }
    TN_WP_FLUSH
#undef active
}


} // namespace tn
