// tn_kernels.h -- the gfx950 kernels.
//
// Streaming (wavefront) pipeline, one batch of B = pixels x passes path slots:
//
//   for bounce in 0..maxDepth-1:
//     k_bounce<FIRST = bounce==0>   one iteration of the oracle's path loop for every path in
//                                   queue[bounce] (bounce 0: camera rays generated in-kernel);
//                                   survivors' 96-B state -> HBM, wave64 ballot compaction into
//                                   queue[bounce+1]
//   k_accumulate                    filter-footprint GATHER into the float4 accumulator (no atomics,
//                                   bit-reproducible, same summation order as render.cpp:401-445)
//
// A/B arms sharing the same per-path arithmetic: the SPLIT pipeline (k_generate, then k_extend /
// k_shade / k_shadow per bounce with hit and NEE records parked in HBM) and k_mega (one lane per
// whole path).  All trace kernels are streaming: a fixed grid whose blocks own contiguous queue ranges and
// append survivors with one atomic per 2048 entries; traversal stacks live in LDS as stack[entry][lane].
#pragma once

#include "tn_integrator.h"
#include "tn_display.h"
#include "tn_walk.h"

namespace tn {

// Minimum waves per SIMD the register allocator must leave room for (2nd __launch_bounds__
// argument).  Measured on cornell 1024^2 (profiles/r01_b): the fused kernels are fastest at 2
// (256 VGPRs, no AGPR spill copies, ~200 B scratch), the trace-only kernels at 4 (128 VGPRs).
#ifndef TN_WAVES_FUSED
#define TN_WAVES_FUSED 2
#endif
// k_bounce alone: the tolerance arm's k_bounce needs ~205 VGPRs and is faster squeezed to 168 (3 waves per SIMD, ~100 B of
// scratch: cornell 3203 -> 3827 Msamples/s); its k_shade is not (veach 1698 -> 1563), and the exact arm's k_bounce (256 VGPRs
// + scratch already) is 1.4-1.8x slower when squeezed.
#ifndef TN_WAVES_BOUNCE
#define TN_WAVES_BOUNCE TN_WAVES_FUSED
#endif
#ifndef TN_WAVES_SHADE
#define TN_WAVES_SHADE TN_WAVES_FUSED
#endif
#ifndef TN_WAVES_TRACE
#define TN_WAVES_TRACE 4
#endif
constexpr int kBlock = 256;
constexpr int kWave = 64;

// ---------------------------------------------------------------------------
// per-batch path state (SoA of 16-B records, one per path slot)

struct PathState
{
    float4* rayO;       // origin.xyz, time
    float4* rayD;       // dir.xyz, bsdfPdf
    float4* thr;        // throughput.xyz, rayEta
    float4* rad;        // radiance.xyz, rayType (int bits)
    float4* absorb;     // rayAbsorption.xyz, -
    float4* rngRaster;  // rng.s1, rng.s2 (bits), rasterX, rasterY     (rasterX < -1e29: slot not owned by this shard)
    float4* hit;        // t, n.xyz
    int32_t* hitPrim;
    float4* nee;        // [slot*neeStride + 4*k + {0..3}] : {o,dist} {wi,nl} {f,bsdfPdf} {absDot,light,-,-}
    float4* neeThr;     // throughput at NEE time
    int32_t neePerPath; // K
};

struct QueueCtl
{
    // all indexed by bounce; zeroed once per batch
    uint32_t* activeCount;  // [maxDepth+1]   entries at the FRONT of queue[bounce]
    uint32_t* activeBack;   // [maxDepth+1]   entries at the BACK (fused pipeline: rays that meet no bounded primitive)
    uint32_t* neeCount;     // [maxDepth]     front of the shadow queue of the bounce
    uint32_t* neeBack;      // [maxDepth]     back of it
    uint32_t* cursorExtend; // [maxDepth]
    uint32_t* cursorShade;  // [maxDepth]
    uint32_t* cursorShadow; // [maxDepth]
    unsigned long long* stats;  // [0]=rays traced [1]=samples [2]=internal visits [3]=tri tests [4]=prim tests [5]=shadow rays
};

struct CameraParams
{
    float r2w[16];      // rasterToWorld, column-major (util.h:45-71)
    float ox, oy, oz;   // cameraToWorld.GetCol(3)
    float shutterStart, shutterEnd;
};

struct FrameParams
{
    int width, height;
    int passBase;           // first pass of this batch (index into passSeeds)
    int numPasses;          // passes in this batch
    int accBegin, accEnd;   // the batch passes [accBegin, accEnd) the accumulate kernels add (all of them, or one call's worth: look-ahead)
    int maxDepth;
    int shardRank, shardWorld, shardTile;
    int shardTilesX, shardOwnedTiles;   // tiles per frame row; tiles this shard owns (t % world == rank)
    uint32_t shardPerPass;              // path slots per pass of this shard (owned tiles x tile^2; W*H for one shard)
    uint32_t genCount;                  // camera paths the generation kernels enumerate per batch (gen_slot)
    uint32_t queueCapacity;             // entries per ray queue (= path slots of the batch)
    int rrStart;                        // > 0: Russian roulette from this bounce on (opt-in, not the reference's behaviour)
    int filterType;
    float filterWidth, filterFalloff, filterOffset;
    float clampLen;
};

// ---------------------------------------------------------------------------
// wave-level helpers
//
// Single-address atomics retire at ~88 M/s on this chip (MI355X_MICROARCH.md, "dequeue" row): one
// atomic per 64 rays caps a kernel at ~5.6 Grays/s per counter, one per 128 still costs ~0.4 ms per
// 4 Mi rays (profiles/r01_a, r01_b).  So the queue is cut STATICALLY into contiguous per-block
// ranges (blocks are handed to CUs dynamically by the dispatcher, which is all the load balancing
// a 2048-block grid needs) and a block appends its survivors with ONE atomic per queue per
// kMaxItems x 256 entries, after a wave64-ballot + LDS scan.

constexpr int kMaxItems = 8;
constexpr int kStatShards = 2048;       // stats[kStatShards][8]
constexpr int kStatWords = 8;
constexpr int kScanWords = 16;           // LDS words behind the traversal stacks used by block_append

TN_D int lane_id() { return (int)__lane_id(); }

// rounds of kBlock entries this block must make over a queue of `count` entries
TN_D uint32_t block_rounds(uint32_t count)
{
    return (count + gridDim.x*kBlock - 1u)/(gridDim.x*kBlock);
}

// Appends this block's survivors of up to kMaxItems rounds.  bits: thread-private mask, bit i =
// the entry this thread handled in round i survives; slot(i) returns the value to append for it.
// One atomic per block.  MUST be reached by every thread of the block (it synchronises).
// `last` != 0: entries are placed from index `last` DOWNWARDS (queue[last - position]) -- used to fill one array from
// both ends.
template <class SlotFn>
TN_D void block_append(uint32_t bits, uint32_t* counter, uint32_t* __restrict__ queue, uint32_t* s_scan, SlotFn slot, uint32_t last = 0u)
{
    const int lane = lane_id();
    const int wave = (int)threadIdx.x/kWave;
    unsigned long long masks[kMaxItems];
    uint32_t waveTotal = 0;
#pragma unroll
    for (int i = 0; i < kMaxItems; ++i)
    {
        masks[i] = __ballot((bits >> i) & 1u);
        waveTotal += (uint32_t)__popcll(masks[i]);
    }
    if (lane == 0)
        s_scan[wave] = waveTotal;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        const uint32_t total = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
        s_scan[4] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    uint32_t base = s_scan[4];
    for (int w = 0; w < wave; ++w)
        base += s_scan[w];
#pragma unroll
    for (int i = 0; i < kMaxItems; ++i)
    {
        if ((bits >> i) & 1u)
        {
            const uint32_t pos = base + (uint32_t)__popcll(masks[i] & ((1ull << lane) - 1ull));
            queue[last ? last - pos : pos] = slot(i);
        }
        base += (uint32_t)__popcll(masks[i]);
    }
    __syncthreads();        // s_scan is reused by the next call
}

// A queue filled from both ends: entries [0, front) and [capacity - back, capacity).  Item `idx` of the front + back
// items, front ones first.
TN_D uint32_t two_ended(uint32_t idx, uint32_t front, uint32_t back, uint32_t capacity)
{
    return idx < front ? idx : capacity - back + (idx - front);
}

// Both ends of a two-ended queue in one go (same synchronisation cost as one block_append): `front` bits are placed
// upwards from counterFront's cursor, `back` bits downwards from `last`.  s_scan needs 10 words.
template <class SlotFn>
TN_D void block_append2(uint32_t front, uint32_t back, uint32_t* counterFront, uint32_t* counterBack, uint32_t* __restrict__ queue,
                        uint32_t* s_scan, SlotFn slot, uint32_t last)
{
    const int lane = lane_id();
    const int wave = (int)threadIdx.x/kWave;
    unsigned long long mf[kMaxItems], mb[kMaxItems];
    uint32_t totalF = 0, totalB = 0;
#pragma unroll
    for (int i = 0; i < kMaxItems; ++i)
    {
        mf[i] = __ballot((front >> i) & 1u);
        mb[i] = __ballot((back >> i) & 1u);
        totalF += (uint32_t)__popcll(mf[i]);
        totalB += (uint32_t)__popcll(mb[i]);
    }
    if (lane == 0)
    {
        s_scan[wave] = totalF;
        s_scan[5 + wave] = totalB;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        const uint32_t tf = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
        const uint32_t tb = s_scan[5] + s_scan[6] + s_scan[7] + s_scan[8];
        s_scan[4] = tf ? atomicAdd(counterFront, tf) : 0u;
        s_scan[9] = tb ? atomicAdd(counterBack, tb) : 0u;
    }
    __syncthreads();
    uint32_t baseF = s_scan[4], baseB = s_scan[9];
    for (int w = 0; w < wave; ++w)
    {
        baseF += s_scan[w];
        baseB += s_scan[5 + w];
    }
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int i = 0; i < kMaxItems; ++i)
    {
        if ((front >> i) & 1u)
            queue[baseF + (uint32_t)__popcll(mf[i] & below)] = slot(i);
        if ((back >> i) & 1u)
            queue[last - (baseB + (uint32_t)__popcll(mb[i] & below))] = slot(i);
        baseF += (uint32_t)__popcll(mf[i]);
        baseB += (uint32_t)__popcll(mb[i]);
    }
    __syncthreads();        // s_scan is reused by the next call
}

// statistics: wave reduction, then one atomic per wave into this block's shard (distinct addresses)
TN_D void wave_add_stat(unsigned long long* stats, int word, uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off);
    if (lane_id() == 0 && v)
        atomicAdd(stats + (size_t)(blockIdx.x % kStatShards)*kStatWords + word, (unsigned long long)v);
}

// Stages the scene arena into LDS (cooperative 16-B copies) and re-points the scene at the LDS copy.
//   SceneT<true>  (host guarantees the arena holds EVERYTHING incl. every mesh and fits): pointers are
//                 derived unconditionally from the LDS base, so every scene access compiles to ds_read.
//   SceneT<false> generic pointers: staged only when arenaLdsBytes != 0, reached through flat loads, and
//                 large meshes stay in HBM.  New pointers are derived FROM the LDS base (base + offset
//                 inside the arena), never from the old global pointers: the back-end assumes
//                 kernel-argument pointers are global, and global + delta would be issued as a global
//                 load of an LDS aperture address.
// MUST be reached by every thread of the block.
template <bool LDS, bool WONLY>
TN_D void stage_scene_lds(SceneT<LDS, WONLY>& sc, const DevScene& in, uint32_t* ldsWords)
{
    static_cast<DevScene&>(sc) = in;
    unsigned char* lds = reinterpret_cast<unsigned char*>(ldsWords);
    sc.ldsBase = lds;
    sc.walkRec = nullptr;
    sc.walkItem = 0u;
    if (!LDS && in.arenaLdsBytes == 0)
        return;

    const float4* src = reinterpret_cast<const float4*>(in.arena);
    float4* dst = reinterpret_cast<float4*>(lds);
    const uint32_t n16 = (LDS ? in.arenaBytes : in.arenaLdsBytes)/16u;
    for (uint32_t i = threadIdx.x; i < n16; i += kBlock)
        dst[i] = src[i];
    __syncthreads();

    const unsigned char* g0 = in.arena;
    auto rebase = [&](const void* p) -> const unsigned char* {
        return lds + (reinterpret_cast<const unsigned char*>(p) - g0);
    };
    if (!LDS)
    {
        DevMesh* lm = reinterpret_cast<DevMesh*>(lds + (reinterpret_cast<const unsigned char*>(in.meshes) - g0));
        for (int i = threadIdx.x; i < in.numMeshes; i += kBlock)
        {
            if (lm[i].inArena)
            {
                lm[i].nodes = reinterpret_cast<const Node64*>(lds + lm[i].offNodes);
                lm[i].tris = reinterpret_cast<const Tri48*>(lds + lm[i].offTris);
                lm[i].normals = reinterpret_cast<const float*>(lds + lm[i].offNormals);
                lm[i].cdf = reinterpret_cast<const float*>(lds + lm[i].offCdf);
            }
        }
        __syncthreads();
    }

    sc.nodes = reinterpret_cast<const Node64*>(rebase(in.nodes));
    sc.prims = reinterpret_cast<const Prim64*>(rebase(in.prims));
    sc.mats = reinterpret_cast<const Mat128*>(rebase(in.mats));
    sc.moving = reinterpret_cast<const Moving64*>(rebase(in.moving));
    sc.meshes = reinterpret_cast<const DevMesh*>(rebase(in.meshes));
    sc.lights = reinterpret_cast<const int32_t*>(rebase(in.lights));
    sc.primBoxes = reinterpret_cast<const PrimBox*>(rebase(in.primBoxes));
}

TN_D bool pixel_owned(const FrameParams& fp, int i, int j)
{
    if (fp.shardWorld <= 1)
        return true;
    const int tilesX = (fp.width + fp.shardTile - 1)/fp.shardTile;
    const int t = (j/fp.shardTile)*tilesX + (i/fp.shardTile);
    return (t % fp.shardWorld) == fp.shardRank;
}

// Path slots.  One shard: slot = pass*W*H + j*W + i.  Several: slots are RANK-LOCAL -- the shard's own tiles one after
// the other, pass by pass (slot = pass*perPass + k*T*T + (j%T)*T + i%T for the shard's k-th tile) -- so a rank's state
// arrays hold exactly the paths it traces whatever the number of ranks, the generation kernels' lanes are all busy and
// consecutive slots are consecutive pixels of a tile.  Tiles that stick out of the frame are padded to full size; the
// padding slots are never generated, written or read.
TN_D bool slot_pixel(const FrameParams& fp, uint32_t slot, int& s, int& i, int& j)
{
    if (fp.shardWorld <= 1)
    {
        const uint32_t npix = (uint32_t)(fp.width*fp.height);
        const uint32_t ss = slot/npix;
        const uint32_t pix = slot - ss*npix;
        const uint32_t jj = pix/(uint32_t)fp.width;
        s = (int)ss; j = (int)jj; i = (int)(pix - jj*(uint32_t)fp.width);
        return true;
    }
    const uint32_t T = (uint32_t)fp.shardTile;
    const uint32_t ss = slot/fp.shardPerPass;
    const uint32_t o = slot - ss*fp.shardPerPass;
    const uint32_t k = o/(T*T);
    const uint32_t within = o - k*T*T;
    const uint32_t t = (uint32_t)fp.shardRank + k*(uint32_t)fp.shardWorld;
    const uint32_t ty = t/(uint32_t)fp.shardTilesX, tx = t - ty*(uint32_t)fp.shardTilesX;
    const uint32_t wy = within/T, wx = within - wy*T;
    s = (int)ss; i = (int)(tx*T + wx); j = (int)(ty*T + wy);
    return i < fp.width && j < fp.height;
}

// slot of the path of pass `s` (in the batch) generated at pixel (i, j); several shards: the pixel must be owned
TN_D uint32_t slot_of(const FrameParams& fp, int s, int i, int j)
{
    if (fp.shardWorld <= 1)
        return (uint32_t)s*(uint32_t)(fp.width*fp.height) + (uint32_t)j*(uint32_t)fp.width + (uint32_t)i;
    const uint32_t T = (uint32_t)fp.shardTile;
    const uint32_t ty = (uint32_t)j/T, tx = (uint32_t)i/T;
    const uint32_t k = (ty*(uint32_t)fp.shardTilesX + tx)/(uint32_t)fp.shardWorld;
    return (uint32_t)s*fp.shardPerPass + k*T*T + ((uint32_t)j - ty*T)*T + ((uint32_t)i - tx*T);
}

// The idx-th camera path this shard generates in a batch -> its slot (= idx); false for tile padding.
TN_D bool gen_slot(const FrameParams& fp, uint32_t idx, uint32_t& slot)
{
    slot = idx;
    if (fp.shardWorld <= 1)
        return true;
    int s, i, j;
    return slot_pixel(fp, idx, s, i, j);
}

// CameraSampler::GenerateRay (util.h:73-79) with TransformPoint(Mat44, Vec3) (maths.h:917-924)
TN_D void generate_ray(const CameraParams& c, float rx, float ry, V3& o, V3& d)
{
    const float vz = 0.0f;
    V3 p;
    p.x = c.r2w[0]*rx + c.r2w[4]*ry + c.r2w[8]*vz + c.r2w[12];
    p.y = c.r2w[1]*rx + c.r2w[5]*ry + c.r2w[9]*vz + c.r2w[13];
    p.z = c.r2w[2]*rx + c.r2w[6]*ry + c.r2w[10]*vz + c.r2w[14];
    o = V3(c.ox, c.oy, c.oz);
    d = normalize(p - o);
}

// The camera sample of one path: seed contract + draw order of render.cpp:476-484
TN_D void camera_sample(const CameraParams& cam, const FrameParams& fp, int i, int j, uint32_t passSeed,
                        Rng& rng, float& rx, float& ry, float& time, V3& o, V3& d)
{
    rng = Rng::seeded((uint32_t)i + (uint32_t)j*(uint32_t)fp.width + passSeed);
    float x = rng.randf();
    float y = rng.randf();
    float t = rng.randf();
    time = lerpf(cam.shutterStart, cam.shutterEnd, t);
    rx = x + i;
    ry = y + j;
    generate_ray(cam, rx, ry, o, d);
}

// ---------------------------------------------------------------------------
// path-state load/store

TN_D void load_path(const PathState& ps, uint32_t slot, PathRegs& p, float& rasterX, float& rasterY, bool hasMedia)
{
    const float4 ro = ps.rayO[slot], rd = ps.rayD[slot], th = ps.thr[slot], ra = ps.rad[slot];
    const float4 ab = hasMedia ? ps.absorb[slot] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const float4 rr = ps.rngRaster[slot];
    p.o = V3(ro.x, ro.y, ro.z); p.time = ro.w;
    p.d = V3(rd.x, rd.y, rd.z); p.bsdfPdf = rd.w;
    p.thr = V3(th.x, th.y, th.z); p.eta = th.w;
    p.rad = V3(ra.x, ra.y, ra.z); p.rayType = __float_as_int(ra.w);
    p.absorption = V3(ab.x, ab.y, ab.z);
    p.rng.s1 = __float_as_uint(rr.x); p.rng.s2 = __float_as_uint(rr.y);
    rasterX = rr.z; rasterY = rr.w;
}

TN_D void store_path(const PathState& ps, uint32_t slot, const PathRegs& p, float rasterX, float rasterY, bool hasMedia)
{
    ps.rayO[slot] = make_float4(p.o.x, p.o.y, p.o.z, p.time);
    ps.rayD[slot] = make_float4(p.d.x, p.d.y, p.d.z, p.bsdfPdf);
    ps.thr[slot] = make_float4(p.thr.x, p.thr.y, p.thr.z, p.eta);
    ps.rad[slot] = make_float4(p.rad.x, p.rad.y, p.rad.z, __int_as_float(p.rayType));
    if (hasMedia)
        ps.absorb[slot] = make_float4(p.absorption.x, p.absorption.y, p.absorption.z, 0.0f);
    ps.rngRaster[slot] = make_float4(__uint_as_float(p.rng.s1), __uint_as_float(p.rng.s2), rasterX, rasterY);
}

// Slot -> (pass, pixel); generates the camera sample.  Returns false for tile padding.
TN_D bool begin_path(const CameraParams& cam, const FrameParams& fp, const uint32_t* __restrict__ passSeeds, uint32_t slot,
                     PathRegs& p, float& rx, float& ry)
{
    int s, i, j;
    if (!slot_pixel(fp, slot, s, i, j))
        return false;
    Rng rng;
    float time;
    V3 o, d;
    camera_sample(cam, fp, i, j, passSeeds[fp.passBase + s], rng, rx, ry, time, o, d);
    path_begin(p, o, d, time, rng);
    return true;
}

// ---------------------------------------------------------------------------
// k_bounce: the streaming pipeline's per-bounce kernel (the product path).
//
// One launch per bounce.  Each lane takes ONE live path from queue[bounce] (bounce 0: straight
// from the camera), runs one iteration of the oracle's loop (render.cpp:250-385: closest hit,
// emission/MIS, every NEE shadow ray, BSDF sample) and either retires the path or writes its
// 96-B state back and appends it to queue[bounce+1].  Lanes are therefore always full at the
// start of a bounce, and a path costs one state read + one state write per bounce.

// Developer-only section timer (-DTN_PROFILE_SECTIONS, never in the shipped library): per-wave s_memtime
// deltas of the k_bounce sections, summed into the stats words 2..7 instead of the traversal counters.
#ifdef TN_PROFILE_SECTIONS
#define TN_PROF_DECL uint32_t prof[6] = { 0, 0, 0, 0, 0, 0 }; long long tprev = clock64();
#define TN_TICK(k) { const long long _t = clock64(); prof[k] += (uint32_t)(_t - tprev); tprev = _t; }
#define TN_PROF_FLUSH if (lane_id() == 0) { for (int k = 0; k < 6; ++k) atomicAdd(q.stats + (size_t)(blockIdx.x % kStatShards)*kStatWords + 2 + k, (unsigned long long)prof[k]); } if (true) return;
#else
#define TN_TICK(k)
#ifdef TN_PROFILE_TRACE
#define TN_PROF_DECL TraceCounters ctrN = { 0, 0, 0 };
#define TN_CTR_NEE ctrN
#define TN_PROF_FLUSH if (lane_id() == 0) { for (int k = 0; k < 6; ++k) atomicAdd(q.stats + (size_t)(blockIdx.x % kStatShards)*kStatWords + 2 + k, (unsigned long long)(TN_PROFILE_TRACE == 2 ? ctrN.cyc[k] : ctr.cyc[k])); } if (true) return;
#else
#define TN_PROF_DECL
#define TN_PROF_FLUSH
#endif
#endif
#ifndef TN_CTR_NEE
#define TN_CTR_NEE ctr
#endif

template <bool COUNT, bool FIRST, bool LDS>
__global__ __launch_bounds__(kBlock, TN_WAVES_BOUNCE) void k_bounce(DevScene scIn, PathState ps, QueueCtl q, const uint32_t* __restrict__ queueIn,
                                                   uint32_t* __restrict__ queueOut, int bounce, int stackEntries, CameraParams cam,
                                                   FrameParams fp, const uint32_t* __restrict__ passSeeds)
{
    extern __shared__ uint32_t s_stack[];      // [stackEntries][kBlock], sized at launch
    LdsStack<kBlock> st = { s_stack + threadIdx.x };

    uint32_t* s_scan = s_stack + stackEntries*kBlock;
    SceneT<LDS> sc;
    stage_scene_lds(sc, scIn, s_scan + kScanWords);

    const uint32_t frontCount = FIRST ? 0u : q.activeCount[bounce], backCount = FIRST ? 0u : q.activeBack[bounce];
    const uint32_t count = FIRST ? fp.genCount : frontCount + backCount;
    const uint32_t rounds = block_rounds(count);
    const uint32_t first = blockIdx.x*rounds*kBlock;       // this block's contiguous range
    uint32_t rays = 0, shadowRays = 0, samples = 0;
    TraceCounters ctr = { 0, 0, 0 };
    TN_PROF_DECL

    for (uint32_t r0 = 0; r0 < rounds; r0 += kMaxItems)
    {
        const uint32_t base = first + r0*kBlock;
        const uint32_t groups = (rounds - r0) < (uint32_t)kMaxItems ? (rounds - r0) : (uint32_t)kMaxItems;

        uint32_t keep = 0, keepBack = 0;
        for (uint32_t g = 0; g < groups; ++g)
        {
            const uint32_t idx = base + g*kBlock + threadIdx.x;
            if (idx >= count)
                continue;
            uint32_t slot;
            if (FIRST)
            {
                if (!gen_slot(fp, idx, slot))
                    continue;
            }
            else
                slot = queueIn[two_ended(idx, frontCount, backCount, fp.queueCapacity)];

            TN_TICK(4)
            PathRegs p;
            float rx, ry;
            if (FIRST)
            {
                if (!begin_path(cam, fp, passSeeds, slot, p, rx, ry))
                {
                    ps.rad[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    ps.rngRaster[slot] = make_float4(0.0f, 0.0f, -1e30f, -1e30f);
                    continue;
                }
                samples++;
            }
            else
            {
                load_path(ps, slot, p, rx, ry, sc.hasMedia != 0);
            }

            TN_TICK(0)
            float t;
            V3 n;
            const int prim = trace<SceneT<LDS>, LdsStack<kBlock>, COUNT>(sc, st, p.o, p.d, p.time, t, n, ctr);
            rays++;
            TN_TICK(1)

            bool alive = false;
            if (prim < 0)
            {
                on_miss(sc, p, bounce);
            }
            else
            {
                const Mat mat = load_mat(sc.mats, prim);
                HitCtx h;
                on_hit_begin(p, mat, t, n, bounce, h);

                if (sc.totalLightSamples > 0)
                {
                    const V3 thrAtNee = p.thr;
                    int li = 0, sInLight = 0;
                    V3 sum = nee_sum(sc, [&](int k) -> V3 {
                        NeeRec r;
                        // the 28-register material record is re-read per sample instead of living across the shadow trace
                        const Mat matK = load_mat(sc.mats, prim);
                        if (sc.probe.valid && k == 0)
                        {
                            nee_prepare_probe(sc, matK, h, p.rng, r);
                        }
                        else
                        {
                            // NEE rays arrive in order: walk (light, sample) along with k
                            while (sInLight >= sc.mats[sc.lights[li]].lightSamples) { ++li; sInLight = 0; }
                            nee_prepare_light(sc, matK, h, p.time, sc.lights[li], p.rng, r);
                            ++sInLight;
                        }
                        TN_TICK(2)
                        float ts;
                        V3 nn;
                        const int hp = trace<SceneT<LDS>, LdsStack<kBlock>, COUNT>(sc, st, r.o, r.wi, p.time, ts, nn, TN_CTR_NEE);
                        TN_TICK(3)
                        rays++;
                        shadowRays++;
                        if (r.dist < 0.0f)
                            return (hp < 0) ? r.f : V3(0.0f);
                        return nee_resolve_light(sc, r, hp, ts);
                    });
                    p.rad = p.rad + thrAtNee*sum;
                }

                // the last iteration's BSDF sample is never used by the oracle's loop (render.cpp:250)
                TN_TICK(2)
                if (bounce + 1 < fp.maxDepth)
                {
                    // the material is read again rather than kept in 28 registers across the shadow traces
                    const Mat matAgain = load_mat(sc.mats, prim);
                    alive = (bsdf_step(p, matAgain, h) == kContinue);
                    if (alive && fp.rrStart > 0 && bounce + 1 >= fp.rrStart)
                        alive = roulette_survives(p);
                }
            }

            TN_TICK(5)
            if (alive)
            {
                store_path(ps, slot, p, rx, ry, sc.hasMedia != 0);
                // next bounce's queue, sorted: rays that meet a bounded primitive's box in front, plane-only rays at the back
                if (!sc.sortQueues || ray_meets_bounded_prim(sc, p.o, p.d))
                    keep |= 1u << g;
                else
                    keepBack |= 1u << g;
            }
            else
            {
                ps.rad[slot] = make_float4(p.rad.x, p.rad.y, p.rad.z, 0.0f);
                if (FIRST)
                    ps.rngRaster[slot] = make_float4(0.0f, 0.0f, rx, ry);
            }
        }

        auto slotOf = [&](int i) -> uint32_t {
            const uint32_t idx = base + (uint32_t)i*kBlock + threadIdx.x;
            uint32_t slot = 0;
            if (FIRST)
                (void)gen_slot(fp, idx, slot);
            else
                slot = queueIn[two_ended(idx, frontCount, backCount, fp.queueCapacity)];
            return slot;
        };
        if (sc.sortQueues)
            block_append2(keep, keepBack, q.activeCount + bounce + 1, q.activeBack + bounce + 1, queueOut, s_scan, slotOf, fp.queueCapacity - 1u);
        else
            block_append(keep, q.activeCount + bounce + 1, queueOut, s_scan, slotOf);
    }

    wave_add_stat(q.stats, 0, rays);
    wave_add_stat(q.stats, 1, samples);
#if !defined(TN_PROFILE_SECTIONS) && !defined(TN_PROFILE_TRACE)
    wave_add_stat(q.stats, 5, shadowRays);
#endif
    TN_PROF_FLUSH
    if (COUNT)
    {
        wave_add_stat(q.stats, 2, ctr.internal);
        wave_add_stat(q.stats, 3, ctr.tris);
        wave_add_stat(q.stats, 4, ctr.prims);
    }
}

// ===========================================================================
// The SPLIT variant of the pipeline (TINSEL_PIPELINE_WAVEFRONT_SPLIT): the same bounce cut into
// three kernels with hit / NEE records parked in HBM in between.  Kept as an A/B arm: it trades
// ~3x the state traffic for smaller kernels (k_extend/k_shadow 132-136 VGPRs vs k_bounce's).

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
    const V3 rcp(1.0f/d.x, 1.0f/d.y, 1.0f/d.z);
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
// k_generate

__global__ __launch_bounds__(kBlock, 4) void k_generate(PathState ps, QueueCtl q, uint32_t* queue0, CameraParams cam, FrameParams fp,
                                                     const uint32_t* __restrict__ passSeeds, const PrimBox* __restrict__ primBoxes, BinPrims bp)
{
    __shared__ uint32_t s_scan[kScanWords];
    const uint32_t count = fp.genCount;
    const uint32_t rounds = block_rounds(count);
    const uint32_t first = blockIdx.x*rounds*kBlock;
    uint32_t samples = 0;

    for (uint32_t r0 = 0; r0 < rounds; r0 += kMaxItems)
    {
        const uint32_t base = first + r0*kBlock;
        const uint32_t groups = (rounds - r0) < (uint32_t)kMaxItems ? (rounds - r0) : (uint32_t)kMaxItems;
        uint32_t keep = 0, keepBack = 0;
        for (uint32_t g = 0; g < groups; ++g)
        {
            const uint32_t idx = base + g*kBlock + threadIdx.x;
            uint32_t slot;
            if (idx >= count || !gen_slot(fp, idx, slot))
                continue;
            PathRegs p;
            float rx, ry;
            if (begin_path(cam, fp, passSeeds, slot, p, rx, ry))
            {
                store_path(ps, slot, p, rx, ry, true);
                // sorted like every later queue: camera rays that enter a mesh in HBM in front (k_walk takes those)
                if (bp.count == 0 || ray_enters_big_mesh(primBoxes, bp, p.o, p.d)) keep |= 1u << g; else keepBack |= 1u << g;
                samples++;
            }
            else
            {
                ps.rad[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                ps.rngRaster[slot] = make_float4(0.0f, 0.0f, -1e30f, -1e30f);
            }
        }
        auto slotOf = [&](int i) -> uint32_t {
            uint32_t slot = 0;
            (void)gen_slot(fp, base + (uint32_t)i*kBlock + threadIdx.x, slot);
            return slot;
        };
        if (bp.count)
            block_append2(keep, keepBack, q.activeCount + 0, q.activeBack + 0, queue0, s_scan, slotOf, fp.queueCapacity - 1u);
        else
            block_append(keep, q.activeCount + 0, queue0, s_scan, slotOf);
    }
    wave_add_stat(q.stats, 1, samples);
}

// ---------------------------------------------------------------------------
// k_extend: closest hit for every queued path

// WONLY: every mesh of the scene is walked by k_walk: the kernel is the flat scan + record reads, built for more waves
#ifndef TN_WAVES_SCAN
#define TN_WAVES_SCAN 6
#endif
template <bool COUNT, bool LDS, bool WONLY = false>
__global__ __launch_bounds__(kBlock, WONLY ? TN_WAVES_SCAN : TN_WAVES_TRACE) void k_extend(DevScene scIn, PathState ps, QueueCtl q, const uint32_t* __restrict__ queue, int bounce, int stackEntries,
                                                                  uint32_t queueCapacity, const float4* __restrict__ walkRec, uint32_t walkPrims)
{
    extern __shared__ uint32_t s_stack[];      // [stackEntries][kBlock], sized at launch
    LdsStack<kBlock> st = { s_stack + threadIdx.x };
    SceneT<LDS, WONLY> sc;
    stage_scene_lds(sc, scIn, s_stack + stackEntries*kBlock + kScanWords);

    const uint32_t frontCount = q.activeCount[bounce], backCount = q.activeBack[bounce];
    const uint32_t count = frontCount + backCount;
    const uint32_t rounds = block_rounds(count);
    const uint32_t first = blockIdx.x*rounds*kBlock;
    uint32_t rays = 0;
    TraceCounters ctr = { 0, 0, 0 };
    sc.walkRec = walkRec;           // k_walk's records of the front rays (null: meshes are walked inline)

    {
        for (uint32_t g = 0; g < rounds; ++g)
        {
            const uint32_t idx = first + g*kBlock + threadIdx.x;
            if (idx >= count)
                continue;
            const uint32_t slot = queue[two_ended(idx, frontCount, backCount, queueCapacity)];
            const float4 ro = ps.rayO[slot];
            const float4 rd = ps.rayD[slot];
            sc.walkItem = idx*walkPrims;        // only front rays (idx < frontCount) ever reach a walked primitive

            float t;
            V3 n;
            const int prim = trace<SceneT<LDS, WONLY>, LdsStack<kBlock>, COUNT>(sc, st, V3(ro.x, ro.y, ro.z), V3(rd.x, rd.y, rd.z), ro.w, t, n, ctr);

            ps.hit[slot] = make_float4(t, n.x, n.y, n.z);
            ps.hitPrim[slot] = prim;
            rays++;
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

// ---------------------------------------------------------------------------
// k_shade

TN_D void store_nee(const PathState& ps, uint32_t slot, int k, const NeeRec& r)
{
    float4* dst = ps.nee + ((size_t)slot*ps.neePerPath + k)*4;
    dst[0] = make_float4(r.o.x, r.o.y, r.o.z, r.dist);
    dst[1] = make_float4(r.wi.x, r.wi.y, r.wi.z, r.nl);
    dst[2] = make_float4(r.f.x, r.f.y, r.f.z, r.bsdfPdf);
    dst[3] = make_float4(r.absDot, __int_as_float(r.light), 0.0f, 0.0f);
}

TN_D NeeRec load_nee(const PathState& ps, uint32_t slot, int k)
{
    const float4* src = ps.nee + ((size_t)slot*ps.neePerPath + k)*4;
    const float4 a = src[0], b = src[1], c = src[2], d = src[3];
    NeeRec r;
    r.o = V3(a.x, a.y, a.z); r.dist = a.w;
    r.wi = V3(b.x, b.y, b.z); r.nl = b.w;
    r.f = V3(c.x, c.y, c.z); r.bsdfPdf = c.w;
    r.absDot = d.x; r.light = __float_as_int(d.y);
    return r;
}

template <bool LDS>
__global__ __launch_bounds__(kBlock, TN_WAVES_SHADE) void k_shade(DevScene scIn, PathState ps, QueueCtl q, const uint32_t* __restrict__ queue,
                                                  uint32_t* __restrict__ queueNext, uint32_t* __restrict__ queueNee, int bounce, int maxDepth, int rrStart,
                                                  uint32_t queueCapacity, BinPrims bp)
{
    __shared__ uint32_t s_scan[kScanWords];
    extern __shared__ uint32_t s_arena[];
    SceneT<LDS> sc;
    stage_scene_lds(sc, scIn, s_arena);
    const uint32_t frontCount = q.activeCount[bounce], backCount = q.activeBack[bounce];
    const uint32_t count = frontCount + backCount;
    const uint32_t rounds = block_rounds(count);
    const uint32_t first = blockIdx.x*rounds*kBlock;

    for (uint32_t r0 = 0; r0 < rounds; r0 += kMaxItems)
    {
        const uint32_t base = first + r0*kBlock;
        const uint32_t groups = (rounds - r0) < (uint32_t)kMaxItems ? (rounds - r0) : (uint32_t)kMaxItems;

        uint32_t keepNee = 0, keepNext = 0, keepNeeBack = 0, keepNextBack = 0;
        for (uint32_t g = 0; g < groups; ++g)
        {
            const uint32_t idx = base + g*kBlock + threadIdx.x;
            if (idx >= count)
                continue;
            const uint32_t slot = queue[two_ended(idx, frontCount, backCount, queueCapacity)];

            PathRegs p;
            float rx, ry;
            load_path(ps, slot, p, rx, ry, sc.hasMedia != 0);

            const int prim = ps.hitPrim[slot];
            if (prim < 0)
            {
                on_miss(sc, p, bounce);
                ps.rad[slot] = make_float4(p.rad.x, p.rad.y, p.rad.z, __int_as_float(p.rayType));
                continue;
            }

            const float4 hh = ps.hit[slot];
            const Mat mat = load_mat(sc.mats, prim);

            HitCtx h;
            on_hit_begin(p, mat, hh.x, V3(hh.y, hh.z, hh.w), bounce, h);

            // SampleLights, part 1 (render.cpp:107-170): consume the RNG, emit shadow-ray records
            int k = 0;
            bool neeInMesh = bp.count == 0;         // no big mesh: everything goes to the front
            if (sc.probe.valid)
            {
                NeeRec r;
                nee_prepare_probe(sc, mat, h, p.rng, r);
                store_nee(ps, slot, k++, r);
                neeInMesh = neeInMesh || ray_enters_big_mesh(sc.primBoxes, bp, r.o, r.wi);
            }
            for (int li = 0; li < sc.numLights; ++li)
            {
                const int light = sc.lights[li];
                const int ns = sc.mats[light].lightSamples;
                for (int s = 0; s < ns; ++s)
                {
                    NeeRec r;
                    nee_prepare_light(sc, mat, h, p.time, light, p.rng, r);
                    store_nee(ps, slot, k++, r);
                    neeInMesh = neeInMesh || ray_enters_big_mesh(sc.primBoxes, bp, r.o, r.wi);
                }
            }
            if (k > 0)
            {
                ps.neeThr[slot] = make_float4(p.thr.x, p.thr.y, p.thr.z, 0.0f);
                if (neeInMesh) keepNee |= 1u << g; else keepNeeBack |= 1u << g;
            }

            // the last iteration's BSDF sample is never used by the oracle's loop (render.cpp:250)
            int res = kTerminate;
            if (bounce + 1 < maxDepth)
                res = bsdf_step(p, mat, h);
            if (res == kContinue && rrStart > 0 && bounce + 1 >= rrStart && !roulette_survives(p))
                res = kTerminate;

            if (res == kContinue)
            {
                store_path(ps, slot, p, rx, ry, sc.hasMedia != 0);
                if (bp.count == 0 || ray_enters_big_mesh(sc.primBoxes, bp, p.o, p.d)) keepNext |= 1u << g; else keepNextBack |= 1u << g;
            }
            else
            {
                ps.rad[slot] = make_float4(p.rad.x, p.rad.y, p.rad.z, __int_as_float(p.rayType));
            }
        }

        auto slotOf = [&](int i) -> uint32_t { return queue[two_ended(base + (uint32_t)i*kBlock + threadIdx.x, frontCount, backCount, queueCapacity)]; };
        if (bp.count)
        {
            block_append2(keepNee, keepNeeBack, q.neeCount + bounce, q.neeBack + bounce, queueNee, s_scan, slotOf, queueCapacity - 1u);
            block_append2(keepNext, keepNextBack, q.activeCount + bounce + 1, q.activeBack + bounce + 1, queueNext, s_scan, slotOf, queueCapacity - 1u);
        }
        else
        {
            block_append(keepNee, q.neeCount + bounce, queueNee, s_scan, slotOf);
            block_append(keepNext, q.activeCount + bounce + 1, queueNext, s_scan, slotOf);
        }
    }
}

// ---------------------------------------------------------------------------
// k_shadow: SampleLights, part 2 (render.cpp:118-139, 171-224): one thread per path resolves
// its K shadow rays in the oracle's order, then totalRadiance += pathThroughput*sum (render.cpp:314)

template <bool COUNT, bool LDS, bool WONLY = false>
__global__ __launch_bounds__(kBlock, WONLY ? TN_WAVES_SCAN : TN_WAVES_TRACE) void k_shadow(DevScene scIn, PathState ps, QueueCtl q, const uint32_t* __restrict__ queueNee, int bounce, int stackEntries,
                                                                  uint32_t queueCapacity, const float4* __restrict__ walkRec, uint32_t walkPrims)
{
    extern __shared__ uint32_t s_stack[];      // [stackEntries][kBlock], sized at launch
    LdsStack<kBlock> st = { s_stack + threadIdx.x };
    SceneT<LDS, WONLY> sc;
    stage_scene_lds(sc, scIn, s_stack + stackEntries*kBlock + kScanWords);

    const uint32_t frontCount = q.neeCount[bounce], backCount = q.neeBack[bounce];
    const uint32_t count = frontCount + backCount;
    const uint32_t rounds = block_rounds(count);
    const uint32_t first = blockIdx.x*rounds*kBlock;
    uint32_t rays = 0;
    TraceCounters ctr = { 0, 0, 0 };
    sc.walkRec = walkRec;

    {
        for (uint32_t g = 0; g < rounds; ++g)
        {
            const uint32_t idx = first + g*kBlock + threadIdx.x;
            if (idx >= count)
                continue;
            const uint32_t slot = queueNee[two_ended(idx, frontCount, backCount, queueCapacity)];
            const float time = ps.rayO[slot].w;     // rayTime never changes along a path

            V3 sum = nee_sum(sc, [&](int k) -> V3 {
                const NeeRec r = load_nee(ps, slot, k);
                float t;
                V3 n;
                sc.walkItem = (idx*(uint32_t)ps.neePerPath + (uint32_t)k)*walkPrims;
                const int hp = trace<SceneT<LDS, WONLY>, LdsStack<kBlock>, COUNT>(sc, st, r.o, r.wi, time, t, n, ctr);
                rays++;
                if (r.dist < 0.0f)
                    return (hp < 0) ? r.f : V3(0.0f);       // probe sample: contributes iff unoccluded
                return nee_resolve_light(sc, r, hp, t);
            });

            const float4 nt = ps.neeThr[slot];
            float4 ra = ps.rad[slot];
            V3 rad = V3(ra.x, ra.y, ra.z) + V3(nt.x, nt.y, nt.z)*sum;
            ps.rad[slot] = make_float4(rad.x, rad.y, rad.z, ra.w);
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

// ---------------------------------------------------------------------------
// k_mega: the A/B arm -- one lane walks one whole path (render.cpp:230-388), same pieces.

template <bool COUNT, bool LDS>
__global__ __launch_bounds__(kBlock, TN_WAVES_FUSED) void k_mega(DevScene scIn, PathState ps, QueueCtl q, CameraParams cam, FrameParams fp,
                                                 const uint32_t* __restrict__ passSeeds, int stackEntries)
{
    extern __shared__ uint32_t s_stack[];      // [stackEntries][kBlock], sized at launch
    LdsStack<kBlock> st = { s_stack + threadIdx.x };
    SceneT<LDS> sc;
    stage_scene_lds(sc, scIn, s_stack + stackEntries*kBlock + kScanWords);

    const uint32_t idx = blockIdx.x*kBlock + threadIdx.x;
    const uint32_t slot = idx;
    int s = 0, i = 0, j = 0;
    const bool live = idx < fp.genCount && slot_pixel(fp, slot, s, i, j);
    uint32_t rays = 0, shadowRays = 0, samples = 0;
    TraceCounters ctr = { 0, 0, 0 };

    if (live)
    {
        {
            Rng rng;
            float rx, ry, time;
            V3 o, d;
            camera_sample(cam, fp, i, j, passSeeds[fp.passBase + s], rng, rx, ry, time, o, d);
            samples = 1;

            PathRegs p;
            path_begin(p, o, d, time, rng);

            for (int bounce = 0; bounce < fp.maxDepth; ++bounce)
            {
                float t;
                V3 n;
                const int prim = trace<SceneT<LDS>, LdsStack<kBlock>, COUNT>(sc, st, p.o, p.d, p.time, t, n, ctr);
                rays++;

                if (prim < 0)
                {
                    on_miss(sc, p, bounce);
                    break;
                }

                const Mat mat = load_mat(sc.mats, prim);
                HitCtx h;
                on_hit_begin(p, mat, t, n, bounce, h);

                // SampleLights: RNG draws first (all lights, in order), then the traces.  The oracle
                // interleaves draw/trace per sample; the traces consume no random numbers, so the
                // stream is identical.  To keep registers bounded the draws are replayed per sample.
                {
                    const V3 thrAtNee = p.thr;
                    V3 sum = nee_sum(sc, [&](int k) -> V3 {
                        NeeRec r;
                        if (sc.probe.valid && k == 0)
                        {
                            nee_prepare_probe(sc, mat, h, p.rng, r);
                        }
                        else
                        {
                            // locate light of NEE ray k
                            int kk = k - (sc.probe.valid ? 1 : 0);
                            int li = 0;
                            for (;; ++li)
                            {
                                const int ns = sc.mats[sc.lights[li]].lightSamples;
                                if (kk < ns)
                                    break;
                                kk -= ns;
                            }
                            nee_prepare_light(sc, mat, h, p.time, sc.lights[li], p.rng, r);
                        }
                        float ts;
                        V3 nn;
                        const int hp = trace<SceneT<LDS>, LdsStack<kBlock>, COUNT>(sc, st, r.o, r.wi, p.time, ts, nn, ctr);
                        rays++;
                        shadowRays++;
                        if (r.dist < 0.0f)
                            return (hp < 0) ? r.f : V3(0.0f);
                        return nee_resolve_light(sc, r, hp, ts);
                    });
                    p.rad = p.rad + thrAtNee*sum;
                }

                if (bounce + 1 >= fp.maxDepth)
                    break;
                if (bsdf_step(p, mat, h) != kContinue)
                    break;
                if (fp.rrStart > 0 && bounce + 1 >= fp.rrStart && !roulette_survives(p))
                    break;
            }

            ps.rad[slot] = make_float4(p.rad.x, p.rad.y, p.rad.z, 0.0f);
            ps.rngRaster[slot] = make_float4(0.0f, 0.0f, rx, ry);
        }
    }

    wave_add_stat(q.stats, 0, rays);
    wave_add_stat(q.stats, 1, samples);
    wave_add_stat(q.stats, 5, shadowRays);
    if (COUNT)
    {
        wave_add_stat(q.stats, 2, ctr.internal);
        wave_add_stat(q.stats, 3, ctr.tris);
        wave_add_stat(q.stats, 4, ctr.prims);
    }
}

// ---------------------------------------------------------------------------
// k_accumulate: CpuRenderer::AddSample (render.cpp:401-445) as a gather.
// Pixel (px,py) visits the paths generated at pixels (i,j) in raster order, pass by pass, and
// adds the ones whose splat footprint [int(x-fw), int(x+fw)] x [int(y-fw), int(y+fw)] covers it --
// exactly the adds, in exactly the order, the serial oracle performs on that pixel.

template <class Tab>
TN_D float filter_gauss_tab(float x, float falloff, float offset, const Tab& tab)     // same, expf table passed in
{
    return maxT(0.0f, float(m_expf_tab(-falloff*x*x, tab)) - offset);
}

TN_D float filter_gauss(float x, float falloff, float offset)      // Filter::Gaussian (render.h:29-32)
{
    return maxT(0.0f, float(m_expf(-falloff*x*x)) - offset);
}

__global__ __launch_bounds__(kBlock, 4) void k_accumulate(PathState ps, FrameParams fp, float4* __restrict__ accum)
{
    const int npix = fp.width*fp.height;
    const int pix = blockIdx.x*kBlock + threadIdx.x;
    if (pix >= npix)
        return;
    const int py = pix/fp.width;
    const int px = pix - py*fp.width;

    const float fw = fp.filterWidth;
    // generating pixels (i,j) that can reach (px,py): i in [px-1-floor(fw), px+ceil(fw)]
    const int reachLo = 1 + (int)floorf(fw);
    const int reachHi = (int)ceilf(fw);
    const int i0 = maxI(0, px - reachLo), i1 = minI(fp.width - 1, px + reachHi);
    const int j0 = maxI(0, py - reachLo), j1 = minI(fp.height - 1, py + reachHi);

    float4 acc = accum[pix];

    for (int s = fp.accBegin; s < fp.accEnd; ++s)
    {
        for (int j = j0; j <= j1; ++j)
        {
            for (int i = i0; i <= i1; ++i)
            {
                if (!pixel_owned(fp, i, j))
                    continue;       // path not generated by this shard
                const size_t slot = slot_of(fp, s, i, j);
                const float4 rr = ps.rngRaster[slot];
                const float rx = rr.z, ry = rr.w;

                const int startX = maxI(0, int(rx - fw));
                const int startY = maxI(0, int(ry - fw));
                const int endX = minI(int(rx + fw), fp.width - 1);
                const int endY = minI(int(ry + fw), fp.height - 1);
                if (px < startX || px > endX || py < startY || py > endY)
                    continue;

                const float4 ra = ps.rad[slot];
                const V3 c = clamp_length(V3(ra.x, ra.y, ra.z), fp.clampLen);

                if (fp.filterType == 0)
                {
                    acc.x += c.x; acc.y += c.y; acc.z += c.z; acc.w += 1.0f;
                }
                else
                {
                    const float w = filter_gauss(px - rx, fp.filterFalloff, fp.filterOffset)*filter_gauss(py - ry, fp.filterFalloff, fp.filterOffset);
                    acc.x += c.x*w; acc.y += c.y*w; acc.z += c.z*w; acc.w += w;
                }
            }
        }
    }

    accum[pix] = acc;
}

// k_accumulate_tiled: the same gather, one 16x16 pixel tile per block.  Per pass the block stages the
// (16 + halo)^2 candidate paths of its tile into LDS once -- raster position and the ALREADY CLAMPED sample
// (ClampLength is per path, render.cpp:412/431, not per covered pixel) -- instead of every pixel re-reading
// its 16 candidates from L2.  HBM traffic: one 16-B radiance record per path.  Same adds, same order; used when the footprint halo fits (filter width <= 2).

constexpr int kAccTile = 16;
constexpr int kAccMaxHalo = 5;      // reachLo + reachHi
constexpr int kAccSide = kAccTile + kAccMaxHalo;
constexpr int kAccEntries = kAccSide*kAccSide;
constexpr int kAccMaxFoot = 5;      // widest footprint (pixels per axis) for filter widths <= 2

__global__ __launch_bounds__(kBlock, 4) void k_accumulate_tiled(PathState ps, FrameParams fp, float4* __restrict__ accum,
                                                                const uint32_t* __restrict__ passSeeds, const int* __restrict__ tileList)
{
    // per candidate path of the tile: clamped sample, footprint [startX, startX+nX) x [startY, startY+nY)
    // and the separable Gaussian weights of its footprint columns / rows (each shared by up to 5 pixels)
    __shared__ float4 s_c[kAccEntries];                 // rgb, .w = bits(startX | nX << 16)
    __shared__ uint32_t s_y[kAccEntries];               // startY | nY << 16
    __shared__ float s_wx[kAccMaxFoot][kAccEntries];
    __shared__ float s_wy[kAccMaxFoot][kAccEntries];
    __shared__ unsigned long long s_exp[32];            // expf's table: six data-dependent reads per staged path
    if (threadIdx.x < 32)
        s_exp[threadIdx.x] = kExp2fTab[threadIdx.x];
    __syncthreads();

    // Sharded renders launch one block per tile that has candidate paths of THIS shard (tileList, built on the host:
    // ownership depends on the pixel only); the other tiles have nothing to add in any pass, and with N shards they
    // are most of the frame while the pass loop below is N x longer.
    const int tilesX = (fp.width + kAccTile - 1)/kAccTile;
    const int tile = tileList ? tileList[blockIdx.x] : (int)blockIdx.x;
    const int tx = tile % tilesX, ty = tile/tilesX;
    const int lx = threadIdx.x % kAccTile, ly = threadIdx.x/kAccTile;
    const int px = tx*kAccTile + lx, py = ty*kAccTile + ly;
    const bool inside = px < fp.width && py < fp.height;

    const float fw = fp.filterWidth;
    const int reachLo = 1 + (int)floorf(fw);
    const int reachHi = (int)ceilf(fw);
    const int side = kAccTile + reachLo + reachHi;
    const int ox = tx*kAccTile - reachLo, oy = ty*kAccTile - reachLo;     // frame coordinates of LDS entry (0,0)
    const int npix = fp.width*fp.height;
    const bool gauss = fp.filterType != 0;

    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (inside)
        acc = accum[py*fp.width + px];

    // this pixel's candidate window, in LDS coordinates (clipped to the frame like the reference's loops)
    const int i0 = maxI(0, px - reachLo) - ox, i1 = minI(fp.width - 1, px + reachHi) - ox;
    const int j0 = maxI(0, py - reachLo) - oy, j1 = minI(fp.height - 1, py + reachHi) - oy;

    // The (at most two) candidate entries this thread stages every pass: which path, where in LDS, whether the path
    // is this shard's.  The radiance of the NEXT pass is requested before the current pass is processed.
    int entLe[2], entGx[2], entGy[2];
    bool entLive[2];
    float4 nextRa[2];
    for (int k = 0; k < 2; ++k)
    {
        const int e = threadIdx.x + k*kBlock;
        const int ex = e % side, ey = e/side;
        entGx[k] = ox + ex; entGy[k] = oy + ey;
        entLe[k] = ey*kAccSide + ex;
        entLive[k] = e < side*side && entGx[k] >= 0 && entGy[k] >= 0 && entGx[k] < fp.width && entGy[k] < fp.height &&
                     pixel_owned(fp, entGx[k], entGy[k]);
        nextRa[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (entLive[k] && fp.accBegin < fp.accEnd)
            nextRa[k] = ps.rad[slot_of(fp, fp.accBegin, entGx[k], entGy[k])];
    }

    for (int s = fp.accBegin; s < fp.accEnd; ++s)
    {
        float4 curRa[2] = { nextRa[0], nextRa[1] };
        if (s + 1 < fp.accEnd)
            for (int k = 0; k < 2; ++k)
                if (entLive[k])
                    nextRa[k] = ps.rad[slot_of(fp, s + 1, entGx[k], entGy[k])];

        for (int k = 0; k < 2; ++k)
        {
            if (threadIdx.x + k*kBlock >= side*side)
                continue;
            const int gx = entGx[k], gy = entGy[k], le = entLe[k];
            float4 c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);     // nX == 0: covers nothing
            uint32_t ym = 0;
            if (entLive[k])
            {
                // the raster position is the first two draws of the path's own stream (camera_sample):
                // two LCG steps are cheaper than reading it back from the 16-B rngRaster record
                Rng rng = Rng::seeded((uint32_t)gx + (uint32_t)gy*(uint32_t)fp.width + passSeeds[fp.passBase + s]);
                const float x = rng.randf();
                const float y = rng.randf();
                const float rx = x + gx, ry = y + gy;
                const float4 ra = curRa[k];
                const V3 cl = clamp_length(V3(ra.x, ra.y, ra.z), fp.clampLen);

                const int startX = maxI(0, int(rx - fw));
                const int startY = maxI(0, int(ry - fw));
                const int endX = minI(int(rx + fw), fp.width - 1);
                const int endY = minI(int(ry + fw), fp.height - 1);
                const int nX = maxI(0, endX - startX + 1), nY = maxI(0, endY - startY + 1);
                c = make_float4(cl.x, cl.y, cl.z, __uint_as_float((uint32_t)startX | (uint32_t)nX << 16));
                ym = (uint32_t)startY | (uint32_t)nY << 16;
                if (gauss)
                {
                    for (int kk = 0; kk < kAccMaxFoot; ++kk)
                    {
                        if (kk < nX)
                            s_wx[kk][le] = filter_gauss_tab((startX + kk) - rx, fp.filterFalloff, fp.filterOffset, s_exp);
                        if (kk < nY)
                            s_wy[kk][le] = filter_gauss_tab((startY + kk) - ry, fp.filterFalloff, fp.filterOffset, s_exp);
                    }
                }
            }
            s_c[le] = c;
            s_y[le] = ym;
        }
        __syncthreads();

        if (inside)
        {
            for (int j = j0; j <= j1; ++j)
            {
                for (int i = i0; i <= i1; ++i)
                {
                    const int le = j*kAccSide + i;
                    const float4 c = s_c[le];
                    const uint32_t xm = __float_as_uint(c.w), ym = s_y[le];
                    const uint32_t kx = (uint32_t)(px - (int)(xm & 0xffffu)), ky = (uint32_t)(py - (int)(ym & 0xffffu));
                    if (kx >= (xm >> 16) || ky >= (ym >> 16))
                        continue;
                    if (!gauss)
                    {
                        acc.x += c.x; acc.y += c.y; acc.z += c.z; acc.w += 1.0f;
                    }
                    else
                    {
                        const float w = s_wx[kx][le]*s_wy[ky][le];
                        acc.x += c.x*w; acc.y += c.y*w; acc.z += c.z*w; acc.w += w;
                    }
                }
            }
        }
        __syncthreads();
    }

    if (inside)
        accum[py*fp.width + px] = acc;
}

// ---------------------------------------------------------------------------
// k_pass_seeds: passSeed[s] = the (first + s + 1)-th output of Random(1).Rand() (render.cu:1050-1052, 1099), continued on
// the device from the generator state the host keeps for the next pass: one thread, a few thousand integer steps at
// most, and the call that needs the seeds neither copies from host memory nor waits for anything.
__global__ void k_pass_seeds(uint32_t s1, uint32_t s2, int n, uint32_t* __restrict__ out)
{
    if (blockIdx.x != 0 || threadIdx.x != 0)
        return;
    Rng r;
    r.s1 = s1; r.s2 = s2;
    for (int i = 0; i < n; ++i)
        out[i] = r.rand();
}

// ---------------------------------------------------------------------------
// k_normals: eNormals mode of the CPU renderer (render.cpp:494-515): x=i, y=j, time 1, overwrite.

template <bool LDS>
__global__ __launch_bounds__(kBlock, TN_WAVES_TRACE) void k_normals(DevScene scIn, CameraParams cam, FrameParams fp, float4* __restrict__ accum, int stackEntries)
{
    extern __shared__ uint32_t s_stack[];      // [stackEntries][kBlock], sized at launch
    LdsStack<kBlock> st = { s_stack + threadIdx.x };
    SceneT<LDS> sc;
    stage_scene_lds(sc, scIn, s_stack + stackEntries*kBlock + kScanWords);

    const int npix = fp.width*fp.height;
    const int pix = blockIdx.x*kBlock + threadIdx.x;
    if (pix >= npix)
        return;
    const int j = pix/fp.width;
    const int i = pix - j*fp.width;

    V3 o, d;
    generate_ray(cam, float(i), float(j), o, d);

    float t;
    V3 n;
    TraceCounters ctr = { 0, 0, 0 };
    const int prim = trace<SceneT<LDS>, LdsStack<kBlock>, false>(sc, st, o, d, 1.0f, t, n, ctr);
    if (prim >= 0)
    {
        n = n*0.5f + V3(0.5f);
        accum[pix] = make_float4(n.x, n.y, n.z, 1.0f);
    }
    else
    {
        accum[pix] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
}

// ---------------------------------------------------------------------------
// k_leaf: test hook -- the device restatements of the reference's leaf functions evaluated on caller
// arrays, so tests can table them against the reference's own inline functions.

enum LeafOp : int
{
    kLeafRandom = 0,            // seeds -> 4 x Rand(), 4 x Randf() (as float bits)              out stride 8
    kLeafCameraRay = 1,         // in: rasterX, rasterY -> origin(3), dir(3)                       out stride 6
    kLeafBsdfEval = 2,          // in: n(3) V(3) L(3) etaI etaO -> f(3), pdf                       out stride 4
    kLeafBsdfSample = 3,        // in: n(3) V(3) etaI etaO + seed -> L(3), pdf, type, s1, s2       out stride 7
    kLeafPrimIntersect = 4,     // in: origin(3) dir(3) time -> hit, t, n(3)                       out stride 5
    kLeafPrimSample = 5,        // in: time + seed -> pos(3), normal(3), s1, s2                    out stride 8
    kLeafProbe = 6,             // seed -> dir(3), color(3), pdf, ProbePdf(dir), Sky::Eval(dir)(3)  out stride 11
    kLeafLibm = 7,              // in: x, y -> sinf(x), cosf(x), expf(-x), acosf(y), atan2f(y, x - 3)  out stride 5
    kLeafDisplay = 8,           // in: x, y -> powf(x, 2.2f), powf(x, 1/2.2f), expf(y), tonemap_channel(x)  out stride 4
};

__global__ __launch_bounds__(kBlock) void k_leaf(DevScene scIn, int op, int index, int n, const float* __restrict__ in, int inStride,
                                                 const uint32_t* __restrict__ seeds, float* __restrict__ out, int outStride,
                                                 CameraParams cam, int stackEntries)
{
    extern __shared__ uint32_t s_stack[];
    LdsStack<kBlock> st = { s_stack + threadIdx.x };
    SceneT<false> sc;
    stage_scene_lds(sc, scIn, s_stack + stackEntries*kBlock + kScanWords);

    const int i = blockIdx.x*kBlock + threadIdx.x;
    if (i >= n)
        return;
    const float* r = in ? in + (size_t)i*inStride : nullptr;
    float* o = out + (size_t)i*outStride;
    Rng rng = Rng::seeded(seeds ? seeds[i] : 0u);

    if (op == kLeafRandom)
    {
        Rng a = rng, b = rng;
        for (int k = 0; k < 4; ++k)
        {
            o[k] = __uint_as_float(a.rand());
            o[4 + k] = b.randf();
        }
    }
    else if (op == kLeafCameraRay)
    {
        V3 ro, rd;
        generate_ray(cam, r[0], r[1], ro, rd);
        o[0] = ro.x; o[1] = ro.y; o[2] = ro.z; o[3] = rd.x; o[4] = rd.y; o[5] = rd.z;
    }
    else if (op == kLeafBsdfEval)
    {
        const Mat mat = load_mat(sc.mats, index);
        V3 N(r[0], r[1], r[2]), V(r[3], r[4], r[5]), L(r[6], r[7], r[8]);
        V3 f = bsdf_eval(mat, r[9], r[10], N, V, L);
        o[0] = f.x; o[1] = f.y; o[2] = f.z;
        o[3] = bsdf_pdf(mat, r[9], r[10], N, V, L);
    }
    else if (op == kLeafBsdfSample)
    {
        const Mat mat = load_mat(sc.mats, index);
        V3 N(r[0], r[1], r[2]), V(r[3], r[4], r[5]);
        V3 u, v;
        basis_from_vector(N, u, v);
        V3 L(0.0f);
        float pdf = 0.0f;
        int type = kReflected;
        bsdf_sample(mat, r[6], r[7], u, v, N, V, L, pdf, type, rng);
        o[0] = L.x; o[1] = L.y; o[2] = L.z; o[3] = pdf; o[4] = __int_as_float(type);
        o[5] = __uint_as_float(rng.s1); o[6] = __uint_as_float(rng.s2);
    }
    else if (op == kLeafPrimIntersect)
    {
        float t = 0.0f;
        V3 nrm(0.0f);
        TraceCounters ctr = { 0, 0, 0 };
        const bool hit = prim_intersect<SceneT<false>, LdsStack<kBlock>, false>(sc, index, st, 0, V3(r[0], r[1], r[2]), V3(r[3], r[4], r[5]), r[6], t, nrm, ctr);
        o[0] = hit ? 1.0f : 0.0f;
        o[1] = hit ? t : 0.0f;
        o[2] = hit ? nrm.x : 0.0f; o[3] = hit ? nrm.y : 0.0f; o[4] = hit ? nrm.z : 0.0f;
    }
    else if (op == kLeafPrimSample)
    {
        V3 pos, nrm;
        primitive_sample(sc, index, r[0], pos, nrm, rng);
        o[0] = pos.x; o[1] = pos.y; o[2] = pos.z; o[3] = nrm.x; o[4] = nrm.y; o[5] = nrm.z;
        o[6] = __uint_as_float(rng.s1); o[7] = __uint_as_float(rng.s2);
    }
    else if (op == kLeafLibm)
    {
        float sn, cs;
        m_sincosf(r[0], sn, cs);
        o[0] = sn; o[1] = cs; o[2] = m_expf(-r[0]);
        o[3] = m_acosf(r[1]); o[4] = m_atan2f(r[1], r[0] - 3.0f);
    }
    else if (op == kLeafDisplay)
    {
        o[0] = m_powf(r[0], 2.2f); o[1] = m_powf(r[0], 1.0f/2.2f); o[2] = m_expf(r[1]); o[3] = tonemap_channel(r[0]);
    }
    else if (op == kLeafProbe)
    {
        V3 dir(0.0f), color(0.0f);
        float pdf = 0.0f;
        if (sc.probe.valid)
            probe_sample(sc.probe, dir, color, pdf, rng);
        else
        {
            float u1 = rng.randf();
            float u2 = rng.randf();
            dir = uniform_sample_sphere(u1, u2);
        }
        const float pdf2 = sc.probe.valid ? probe_pdf(sc.probe, dir) : 0.0f;
        const V3 e = sky_eval(sc, dir);
        o[0] = dir.x; o[1] = dir.y; o[2] = dir.z; o[3] = color.x; o[4] = color.y; o[5] = color.z;
        o[6] = pdf; o[7] = pdf2; o[8] = e.x; o[9] = e.y; o[10] = e.z;
    }
}

} // namespace tn
