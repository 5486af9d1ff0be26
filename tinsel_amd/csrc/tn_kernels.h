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
// k_bounce: FOUR waves per SIMD (128 VGPRs; 4-9 registers of loop invariants spilled in the prologue).  History: the parity arm's kernel needed
// ~250 registers until the SLP vectoriser went (tinsel_amd/build.py); the third wave then paid several times over -- the kernel waits on
// dependent fp32 / fp64 chains, not on issue slots: cornell 2989 -> 3805 Msamples/s -- and so does the fourth since round 5 put the kernel's
// wave-uniform bookkeeping (region index, base, length, pool pointer: what hangs off threadIdx.x/64) into scalar registers and took slot_pixel's
// reciprocals from the host: 162 -> 134 VGPRs at three waves, and at four cornell 4455 -> 5267, veach 4K 2923 -> 3489, gloss 11 708 -> 13 115,
// env_loft 5016 -> 5912, cfg1 2930 -> 3236, features 1025 -> 1070 -- where only three workgroups' LDS fit a CU too
// (profiles/r05_a_ab_waves4.md, r05_b_ab_bounce_waves.md).
constexpr int kBounceWaves = 4;
// k_shade: four since the end of round 4 -- with the libm coefficients out of its registers (K64, tn_math.h) the staged-arena variant needs
// 133 VGPRs and fits 128 without a byte of scratch (glass k_shade 8.9-9.1 -> 8.7-8.9 ms, motionblur 6.3 -> 5.4, many_spheres +1.7 %:
// profiles/r04_v_ab_k64.md; at three waves it had been 168 VGPRs + 108 B)
#ifndef TN_WAVES_SHADE
#define TN_WAVES_SHADE 4
#endif
#ifndef TN_WAVES_LIGHTS
#define TN_WAVES_LIGHTS 4
#endif
// the lean k_extend carries the light sampling in its tail and needs 123 VGPRs for it
#ifndef TN_WAVES_SCAN_EXTEND
#define TN_WAVES_SCAN_EXTEND 4
#endif
// (without the SLP vectoriser -- tinsel_amd/build.py -- the trace kernels need 77-100 VGPRs: at 5 waves glass's k_extend 6.5 -> 5.7
// ms, many_spheres' 10.4 -> 9.5; at 6 it spills, 11.6)
#ifndef TN_WAVES_TRACE
#define TN_WAVES_TRACE 5
#endif
constexpr int kBlock = 256;
constexpr int kWave = 64;

// ---------------------------------------------------------------------------
// What every pipeline hands to the accumulate kernels: the radiance of the batch's finished paths, by path slot
// (slot <-> (pass, pixel): slot_pixel / slot_of below).  The path state of the wavefront pipelines is SplitState (below).

struct PathState
{
    float4* rad;        // radiance.xyz, -
};

struct QueueCtl
{
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
    uint32_t npixM, widthM;         // floor((2^32 - 1)/(width*height)), floor((2^32 - 1)/width): slot_pixel's divisions as a multiply-high + one correction
    uint32_t perPassM, tileSqM, tileM, tilesXM;     // the same for shardPerPass, shardTile^2, shardTile, shardTilesX (several shards)
    int passBase;           // first pass of this batch (index into passSeeds)
    int numPasses;          // passes in this batch
    int accBegin, accEnd;   // the batch passes [accBegin, accEnd) the accumulate kernels add (all of them, or one call's worth: look-ahead)
    int maxDepth;
    int shardRank, shardWorld, shardTile;
    int shardTilesX, shardOwnedTiles;   // tiles per frame row; tiles this shard owns (t % world == rank)
    uint32_t shardPerPass;              // path slots per pass of this shard (owned tiles x tile^2; W*H for one shard)
    uint32_t genCount;                  // camera paths the generation kernels enumerate per batch (gen_slot)
    int rrStart;                        // > 0: Russian roulette from this bounce on (opt-in, not the reference's behaviour)
    int repack;                         // k_bounce: paths that hit a surface close ranks (per-wave LDS pool) before the shading half
    int share;                          // k_bounce, bounces > 0: the workgroup's four regions form one stream dealt to its waves (host: >= 3 light samples, or short regions)
    int filterType;
    float filterWidth, filterFalloff, filterOffset;
    float clampLen;
};

// ---------------------------------------------------------------------------
// wave-level helpers
//
// Single-address atomics retire at ~88 M/s on this chip (MI355X_MICROARCH.md, "dequeue" row): one atomic per 64 rays
// caps a kernel at ~5.6 Grays/s per counter (round 1's first queues were atomic-bound, profiles/r01_a, r01_b).  The
// wavefront pipelines now append without any: positions come from a wave64 ballot inside a region the wave owns
// (RegionAppend, below); what is left is one atomic per wave per counter for the statistics.

constexpr int kStatShards = 2048;       // stats[kStatShards][8]
constexpr int kStatWords = 8;
constexpr int kScanWords = 16;           // LDS words kept between the traversal stacks and the staged arena

TN_D int lane_id() { return (int)__lane_id(); }

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
template <bool LDS, bool WONLY, int DEFER, bool MIXED>
TN_D void stage_scene_lds(SceneT<LDS, WONLY, DEFER, MIXED>& sc, const DevScene& in, uint32_t* ldsWords, uint32_t blockSize = kBlock)
{
    static_cast<DevScene&>(sc) = in;
    unsigned char* lds = reinterpret_cast<unsigned char*>(ldsWords);
    sc.ldsBase = lds;
    sc.walkRec = nullptr;
    sc.walkItem = 0u;
    sc.kPrims = (ConstF4)(uintptr_t)in.prims;
    sc.kBoxes = (ConstF4)(uintptr_t)in.primBoxes;
    sc.kPlaneEq = (ConstF4)(uintptr_t)in.planeEq;
    sc.kPlaneIdx = (ConstF4)(uintptr_t)in.planeIdx;
    if (!LDS && in.arenaLdsBytes == 0)
        return;

    const float4* src = reinterpret_cast<const float4*>(in.arena);
    float4* dst = reinterpret_cast<float4*>(lds);
    const uint32_t n16 = (LDS ? in.arenaBytes : in.arenaLdsBytes)/16u;
    for (uint32_t i = threadIdx.x; i < n16; i += blockSize)
        dst[i] = src[i];
    __syncthreads();

    const unsigned char* g0 = in.arena;
    auto rebase = [&](const void* p) -> const unsigned char* {
        return lds + (reinterpret_cast<const unsigned char*>(p) - g0);
    };
    if (!LDS)
    {
        DevMesh* lm = reinterpret_cast<DevMesh*>(lds + (reinterpret_cast<const unsigned char*>(in.meshes) - g0));
        for (int i = threadIdx.x; i < in.numMeshes; i += (int)blockSize)
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
// n / d and n % d with the host's m = floor((2^32 - 1)/d): mulhi(n, m) is the quotient or one less (n < 2^32, d >= 1)
TN_D void div_magic(uint32_t n, uint32_t d, uint32_t m, uint32_t& q, uint32_t& rem)
{
    q = __umulhi(n, m);
    rem = n - q*d;
    if (rem >= d) { ++q; rem -= d; }
}

TN_D bool slot_pixel(const FrameParams& fp, uint32_t slot, int& s, int& i, int& j)
{
    if (fp.shardWorld <= 1)
    {
        // slot/npix and pix/width with the host's reciprocals (FrameParams::npixM / widthM): q' = mulhi(n, floor((2^32 - 1)/d)) is q or q - 1
        // for n < 2^32.  The compiler's own expansion computes a float reciprocal of each (wave-uniform) divisor on the VALU, hoists it out
        // of every loop and keeps it in a VGPR for the whole kernel -- four of k_bounce's spilled registers at four waves per SIMD.
        uint32_t ss, pix, jj, ii;
        div_magic(slot, (uint32_t)(fp.width*fp.height), fp.npixM, ss, pix);
        div_magic(pix, (uint32_t)fp.width, fp.widthM, jj, ii);
        s = (int)ss; j = (int)jj; i = (int)ii;
        return true;
    }
    const uint32_t T = (uint32_t)fp.shardTile;
    uint32_t ss, o, k, within, ty, tx, wy, wx;
    div_magic(slot, fp.shardPerPass, fp.perPassM, ss, o);
    div_magic(o, T*T, fp.tileSqM, k, within);
    const uint32_t t = (uint32_t)fp.shardRank + k*(uint32_t)fp.shardWorld;
    div_magic(t, (uint32_t)fp.shardTilesX, fp.tilesXM, ty, tx);
    div_magic(within, T, fp.tileM, wy, wx);
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
// Path state of the wavefront pipelines: DENSE.  What bounds the split pipeline's kernels is HBM traffic, and the L2 fetches
// 128-B lines: a 16-B record read through a queue of sparse (or sorted) slots costs a whole line -- measured on glass,
// maxDepth 12, the light-sampling kernel went from 25 ps per path at bounce 0 (4.8 TB/s) to 157 ps at bounce 11, 6.6 % of the
// slots alive.  So nothing is indexed by a path's slot: the batch is cut into one REGION of `regionLen` positions per wave of the
// grid, and the live paths of a region are PACKED at its two ends -- in front the paths whose ray enters the box of a
// mesh in HBM (k_walk's work, and waves of the scan kernels that are all-mesh or no-mesh; fused kernel, open scenes: the
// rays that meet a bounded primitive's box), at the back all others.  The kernel that ends a bounce (k_shade, k_bounce)
// reads a path's state at its position in buffer `bounce & 1` and writes the survivor to its new position in the other
// buffer; positions come from a wave64 ballot, so there is no queue, no atomic and no barrier, and every load and
// store of every kernel is a run of consecutive 16-B records.  A path carries its slot (the pixel/pass it belongs to)
// to write its radiance where the accumulate kernels look for it.  A path never leaves its region, so that write stays
// local too.
struct SplitState
{
    float4* rayO[2];    // [bounce & 1][sidx(position)]: origin.xyz, time
    float4* rayD[2];    // dir.xyz, bsdfPdf
    float4* thr[2];     // throughput.xyz, rayEta
    float4* rad[2];     // radiance.xyz, rayType (int bits)
    float4* rngId[2];   // rng.s1, rng.s2, path slot (bits), the medium the ray travels in (PathRegs::medium: a primitive index as bits, -1 = none)
    float4* hit;        // [hidx(position)] this bounce's closest hit: t, n.xyz
    int32_t* hitPrim;   // [hidx1(position)]
    uint32_t* pathNee;  // [hidx1(position)] NEE position q of the path's shadow rays of this bounce
    float4* neeRay;     // [(k*2 + {0, 1})*capacity + q] = {o, dist} {wi, nl}: lanes are consecutive q        (k_lights -> k_walk, k_shadow, k_shade)
    float4* neeSky;     // [q] the probe sample's {skyColor, skyPdf}                                           (k_lights -> k_shade)
    float* neeTime;     // [q] rayTime of the path                                                              (k_lights -> k_walk, k_shadow)
    float2* neeRes;     // [k*capacity + q] = {primitive whose emission arrives (int bits; < 0: nothing does), t}  (k_shadow -> k_shade)
    float4* radOut;     // [slot] radiance of finished paths (PathState::rad: what the accumulate kernels read)
    uint32_t* segFront; // [bounce][region] paths packed at the front of the region when the bounce starts
    uint32_t* segBack;  // [bounce][region] ... at its back
    uint32_t* neeFront; // [bounce][region] the same for the paths that have shadow rays, by NEE position
    uint32_t* neeBack;
    uint32_t numRegions, regionLen;     // regionLen is a multiple of 64
    // k_bounce over all bounces only: the regions [bigRegions, numRegions) are SHORT ones (shortLen positions each, a multiple of 64; they
    // follow the long ones in the position space).  Workgroups are dispatched in index order, so the short regions are what the chip works
    // on when the launch runs out: its tail is a short region's time, not a long one's.  bigRegions == numRegions: all alike.
    uint32_t bigRegions, shortLen;
    uint32_t capacity;  // positions per array
    int32_t neePerPath; // K
};

TN_D uint32_t wave_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// this wave's index in its workgroup, as a SCALAR (threadIdx.x/64 is wave-uniform, but only readfirstlane tells the compiler: what hangs off it --
// region index, base, length, pool pointer -- then lives in SGPRs, or their spill lanes, instead of a VGPR each)
TN_D uint32_t wave_in_block() { return wave_uniform(threadIdx.x/kWave); }
// set bits of a wave mask below this lane: v_mbcnt_lo / _hi (no per-lane 64-bit mask kept in two registers)
TN_D uint32_t bits_below(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }

// the i-th live entry of a region packed at both ends
TN_D uint32_t region_pos(uint32_t base, uint32_t len, uint32_t nFront, uint32_t i)
{
    return i < nFront ? base + i : base + len - 1u - (i - nFront);
}

// a region's first position and its length: the regions [bigRegions, numRegions) are the short ones at the end of the position space
// (SplitState::bigRegions); wave-uniform
TN_D uint32_t region_len(const SplitState& ss, uint32_t r) { return r < ss.bigRegions ? ss.regionLen : ss.shortLen; }
TN_D uint32_t region_base(const SplitState& ss, uint32_t r)
{
    return r < ss.bigRegions ? r*ss.regionLen : ss.bigRegions*ss.regionLen + (r - ss.bigRegions)*ss.shortLen;
}

// Appends to the two ends of a region, one wave at a time.  push() must be reached by every lane that is still in the
// caller's loop (lanes with nothing to append pass keep = false).
struct RegionAppend
{
    uint32_t base, len, nFront, nBack;      // wave-uniform
    TN_D uint32_t push(bool keep, bool front)
    {
        const unsigned long long fm = __ballot(keep && front), bm = __ballot(keep && !front);
        const uint32_t pos = front ? base + nFront + bits_below(fm)
                                   : base + len - 1u - (nBack + bits_below(bm));
        nFront += (uint32_t)__popcll(fm);
        nBack += (uint32_t)__popcll(bm);
        return pos;
    }
};

constexpr uint32_t kRegionsPerBlock = kBlock/kWave;

// rayAbsorption of a path whose state says which medium it is in: the material's own vector (what on_hit_begin copied when the path
// entered, render.cpp:262-263), or 0.  Scenes without an absorbing material never look.
TN_D V3 medium_absorption(const DevScene& sc, int medium, bool hasMedia)
{
    if (!hasMedia || medium < 0)
        return V3(0.0f);
    const float4 c = reinterpret_cast<const float4*>(sc.mats + medium)[2];
    return V3(c.x, c.y, c.z);
}

// one of the two buffers of the path state: five arrays by position
struct StateBuf { float4 *rayO, *rayD, *thr, *rad, *rngId; };
TN_D StateBuf state_buf(const SplitState& ss, int buf) { StateBuf b = { ss.rayO[buf], ss.rayD[buf], ss.thr[buf], ss.rad[buf], ss.rngId[buf] }; return b; }

TN_D void load_state(const DevScene& sc, const StateBuf& sb, uint32_t pos, PathRegs& p, uint32_t& slot, bool hasMedia)
{
    const uint32_t at = sidx(pos);
    const float4 ro = sb.rayO[at], rd = sb.rayD[at], th = sb.thr[at], ra = sb.rad[at];
    const float4 rr = sb.rngId[at];
    p.o = V3(ro.x, ro.y, ro.z); p.time = ro.w;
    p.d = V3(rd.x, rd.y, rd.z); p.bsdfPdf = rd.w;
    p.thr = V3(th.x, th.y, th.z); p.eta = th.w;
    p.rad = V3(ra.x, ra.y, ra.z); p.rayType = __float_as_int(ra.w);
    p.rng.s1 = __float_as_uint(rr.x); p.rng.s2 = __float_as_uint(rr.y);
    slot = __float_as_uint(rr.z);
    p.medium = __float_as_int(rr.w);
    p.absorption = medium_absorption(sc, p.medium, hasMedia);
}

TN_D void load_state(const DevScene& sc, const SplitState& ss, int buf, uint32_t pos, PathRegs& p, uint32_t& slot, bool hasMedia)
{
    load_state(sc, state_buf(ss, buf), pos, p, slot, hasMedia);
}

TN_D void store_state(const StateBuf& sb, uint32_t pos, const PathRegs& p, uint32_t slot)
{
    const uint32_t at = sidx(pos);
    sb.rayO[at] = make_float4(p.o.x, p.o.y, p.o.z, p.time);
    sb.rayD[at] = make_float4(p.d.x, p.d.y, p.d.z, p.bsdfPdf);
    sb.thr[at] = make_float4(p.thr.x, p.thr.y, p.thr.z, p.eta);
    sb.rad[at] = make_float4(p.rad.x, p.rad.y, p.rad.z, __int_as_float(p.rayType));
    sb.rngId[at] = make_float4(__uint_as_float(p.rng.s1), __uint_as_float(p.rng.s2), __uint_as_float(slot), __int_as_float(p.medium));
}

TN_D void store_state(const SplitState& ss, int buf, uint32_t pos, const PathRegs& p, uint32_t slot)
{
    store_state(state_buf(ss, buf), pos, p, slot);
}

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

// ---------------------------------------------------------------------------
// k_bounce's shading pool.  A lane runs one iteration of the oracle's loop for its path, and a lane whose ray left the scene
// used to idle through its wave-mates' shading half (shadow traces, light and BSDF terms, the BSDF step: two thirds of a
// round's time; on cornell a fifth to a quarter of the rays of bounces 1..3 leave through the open front).  So between the
// closest-hit trace and the shading half the wave closes ranks through LDS: every wave owns a pool of up to 63 paths that
// have hit something.  After a round's traces either the lanes whose path is finished PULL a waiting path each (when pool +
// this round's hits fill the wave: the shading half runs with 64 lanes), or the round's hits are PUSHED and the shading half
// is skipped this round.  A region thus runs ceil(hits/64) shading rounds instead of one per trace round, no barrier, no
// atomic; a path's arithmetic does not know which lane runs it, so no result changes.  Layout: pool[field][entry], one
// dword per field, consecutive lanes on consecutive entries.
constexpr int kPoolFields = 25;         // (28 until round 5: the medium's absorption vector is looked up again from the medium's index, like load_state does)
constexpr int kPoolWordsPerWave = kPoolFields*kWave;
constexpr int kPoolWords = kPoolWordsPerWave*(kBlock/kWave);     // per workgroup: 25 KB

TN_D void pool_store(uint32_t* pool, uint32_t e, const PathRegs& p, uint32_t slot, int prim, float t, V3 n)
{
    uint32_t* q = pool + e;
    q[0*kWave] = __float_as_uint(p.o.x); q[1*kWave] = __float_as_uint(p.o.y); q[2*kWave] = __float_as_uint(p.o.z);
    q[3*kWave] = __float_as_uint(p.d.x); q[4*kWave] = __float_as_uint(p.d.y); q[5*kWave] = __float_as_uint(p.d.z);
    q[6*kWave] = __float_as_uint(p.time);
    q[7*kWave] = __float_as_uint(p.thr.x); q[8*kWave] = __float_as_uint(p.thr.y); q[9*kWave] = __float_as_uint(p.thr.z);
    q[10*kWave] = __float_as_uint(p.rad.x); q[11*kWave] = __float_as_uint(p.rad.y); q[12*kWave] = __float_as_uint(p.rad.z);
    q[13*kWave] = p.rng.s1; q[14*kWave] = p.rng.s2;
    q[15*kWave] = __float_as_uint(p.eta);
    q[16*kWave] = __float_as_uint(p.bsdfPdf);
    q[17*kWave] = (uint32_t)p.rayType;
    q[18*kWave] = slot;
    q[19*kWave] = (uint32_t)prim;
    q[20*kWave] = __float_as_uint(t);
    q[21*kWave] = __float_as_uint(n.x); q[22*kWave] = __float_as_uint(n.y); q[23*kWave] = __float_as_uint(n.z);
    q[24*kWave] = (uint32_t)p.medium;
}

TN_D void pool_load(const uint32_t* pool, uint32_t e, PathRegs& p, uint32_t& slot, int& prim, float& t, V3& n)
{
    const uint32_t* q = pool + e;
    p.o = V3(__uint_as_float(q[0*kWave]), __uint_as_float(q[1*kWave]), __uint_as_float(q[2*kWave]));
    p.d = V3(__uint_as_float(q[3*kWave]), __uint_as_float(q[4*kWave]), __uint_as_float(q[5*kWave]));
    p.time = __uint_as_float(q[6*kWave]);
    p.thr = V3(__uint_as_float(q[7*kWave]), __uint_as_float(q[8*kWave]), __uint_as_float(q[9*kWave]));
    p.rad = V3(__uint_as_float(q[10*kWave]), __uint_as_float(q[11*kWave]), __uint_as_float(q[12*kWave]));
    p.rng.s1 = q[13*kWave]; p.rng.s2 = q[14*kWave];
    p.eta = __uint_as_float(q[15*kWave]);
    p.bsdfPdf = __uint_as_float(q[16*kWave]);
    p.rayType = (int)q[17*kWave];
    slot = q[18*kWave];
    prim = (int)q[19*kWave];
    t = __uint_as_float(q[20*kWave]);
    n = V3(__uint_as_float(q[21*kWave]), __uint_as_float(q[22*kWave]), __uint_as_float(q[23*kWave]));
    p.medium = (int)q[24*kWave];
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

// The launch covers the bounces [bounceBegin, bounceEnd).  A path never leaves its region and a region belongs to one wave (one
// workgroup where its waves share): nothing a bounce reads was written outside the workgroup, so ONE launch can take its regions
// through ALL the bounces of a batch -- no launch boundary and no tail between bounces (what a 1 M-path batch spends most of its
// time in), no k_region_order launches; the dispatcher balances the workgroups over whole paths instead of over bounces.  Between
// two bounces a workgroup-scope fence (and a barrier where waves share regions) orders the state stores before their loads.
// k_bounce reads three groups of its by-value arguments from the kernel-argument segment WHERE THEY ARE USED, through a pointer the
// compiler cannot see through (so it cannot hoist the scalar loads back to the top): the camera (21 words, bounce 0 only), the sky (probe
// tables, horizon, zenith: 20 words, only for a ray that left the scene or a probe sample) and the path state's pointers (22 words, a
// dozen instructions at each end of a round).  As plain arguments they sat in SGPRs -- or in the VGPR lanes SGPRs spill to, and the
// VGPRs those displace in scratch -- through every bounce: 340 -> 131 v_readlane, scratch 268 -> 216 B in cornell's variant; cornell
// 4297 -> 4404 Msamples/s at 20 passes, veach 4K 2813 -> 2902, gloss 11 062 -> 11 838, env_loft 5537 -> 5753, a 1 M-path batch 2782 -> 2881
// (profiles/r04_r_ab_late_kernargs.md; -DTN_LATE_CAMERA=0 -DTN_LATE_SKY=0 -DTN_LATE_STATE=0: the plain arm)
// k_bounce's kernel arguments: ONE struct, passed by value as the kernel's only parameter -- so the kernel-argument segment IS this struct
// and the offsetof() of the late reads below cannot drift from what the launch lays out (the struct used to mirror a parameter
// list by hand).
struct BounceKernargs { DevScene scIn; SplitState ss; QueueCtl q; int bounceBegin, bounceEnd, stackEntries; CameraParams cam; FrameParams fp; const uint32_t* passSeeds; };
template <bool COUNT, bool LDS, bool DEFER>
__global__ __launch_bounds__(kBlock, kBounceWaves) void k_bounce(BounceKernargs ka)
{
    const DevScene& scIn = ka.scIn;
    const SplitState& ss = ka.ss;
    const QueueCtl& q = ka.q;
    const int bounceBegin = ka.bounceBegin, bounceEnd = ka.bounceEnd, stackEntries = ka.stackEntries;
    const CameraParams& cam = ka.cam;
    const FrameParams& fp = ka.fp;
    const uint32_t* __restrict__ const passSeeds = ka.passSeeds;
    extern __shared__ uint32_t s_stack[];      // [stackEntries][kBlock], sized at launch
    LdsStack<kBlock> st = { s_stack + threadIdx.x };

    // LDS: [stackEntries][kBlock] stack words, kScanWords, (fp.repack) the waves' shading pools, the staged arena
    const bool repack = fp.repack != 0;
    const uint32_t wave = wave_in_block();
    uint32_t* const pool = s_stack + stackEntries*kBlock + kScanWords + wave*kPoolWordsPerWave;
    SceneT<LDS, false, DEFER ? 1 : 0> sc;
    stage_scene_lds(sc, scIn, s_stack + stackEntries*kBlock + kScanWords + (repack ? kPoolWords : 0));
    // the path state's pointers (ten of them and the radiance array: 22 SGPRs that a round needs for a dozen instructions at its start
    // and its end) from the kernel-argument segment where they are used: `ssIn(buf)` what load_state reads of buffer `buf`, `ssOut(buf)`
    // what store_state writes, `radOutNow()` the radiance array of finished paths
    typedef const __attribute__((address_space(4))) SplitState* StatePtr;
    auto state_args = [&]() {
        StatePtr sp = (StatePtr)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(BounceKernargs, ss));
        asm volatile("" : "+s"(sp));
        return sp;
    };
    auto ssBuf = [&](int buf) {
        StatePtr sp = state_args();
        StateBuf b = { sp->rayO[buf], sp->rayD[buf], sp->thr[buf], sp->rad[buf], sp->rngId[buf] };
        return b;
    };
    auto radOutNow = [&]() { return state_args()->radOut; };
#define TN_SS_BUF(buf) ssBuf(buf)
#define TN_RAD_OUT radOutNow()
    // the sky (probe tables, horizon, zenith: 20 words that only a ray that LEFT the scene or a probe sample reads) from the kernel-argument
    // segment where it is needed, like the camera below: on_miss / nee_sample_probe read nothing else of the scene
    auto late_sky = [&]() {
        typedef const __attribute__((address_space(4))) DevScene* ScenePtr;
        ScenePtr sp = (ScenePtr)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(BounceKernargs, scIn));
        asm volatile("" : "+s"(sp));
        DevScene s;
        s.probe.data = sp->probe.data; s.probe.pdfX = sp->probe.pdfX; s.probe.cdfX = sp->probe.cdfX; s.probe.pdfY = sp->probe.pdfY; s.probe.cdfY = sp->probe.cdfY;
        s.probe.width = sp->probe.width; s.probe.height = sp->probe.height; s.probe.valid = sp->probe.valid; s.probe.alias = sp->probe.alias;
        for (int c = 0; c < 3; ++c)
        {
            s.horizon[c] = sp->horizon[c];
            s.zenith[c] = sp->zenith[c];
        }
        return s;
    };

    const uint32_t lane = __lane_id();
    const bool hasMedia = sc.hasMedia != 0;
    uint32_t rays = 0, shadowRays = 0, samples = 0;
    TraceCounters ctr = { 0, 0, 0 };
    TN_PROF_DECL

    // a wave takes a region: bounce 0 generates its camera paths, the others read what the previous bounce packed there
    // (one workgroup per group of four regions, in index order)
    for (uint32_t b = blockIdx.x; b < ss.numRegions/kRegionsPerBlock; b += gridDim.x)
    {
        const uint32_t r0 = b*kRegionsPerBlock;
        const uint32_t r = r0 + wave;                         // the region this wave generates / appends to
        const uint32_t rLen = region_len(ss, r);           // (the same for the four regions of a group)
        const uint32_t base = region_base(ss, r);
      for (int bounce = bounceBegin; bounce < bounceEnd; ++bounce)
      {
        const bool FIRST = bounce == 0;
        const int cur = bounce & 1, nxt = cur ^ 1;
        if (bounce > bounceBegin)
        {
            // this workgroup's stores of the previous bounce (path state, region counts) before this bounce's loads
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (fp.share)
                __syncthreads();        // its waves read each other's regions (`share` below); wave-uniform for the whole grid
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        uint32_t nFront = 0, n;
        // bounces > 0 of scenes with several shadow rays per bounce: the live entries of the workgroup's four regions form ONE
        // stream, dealt to its waves round by round.  A workgroup holds its LDS and its wave slots until its last wave ends;
        // where a round is long (veach: 4 shadow traces, features: 9) its waves drift apart unless they share (veach 1409 ->
        // 1457 Msamples/s, features 685 -> 712); where rounds are short the dealing costs more than it gives (env_loft, gloss -2 %)
        const bool share = !FIRST && fp.share != 0;
        uint32_t gF[kRegionsPerBlock], gStart[kRegionsPerBlock];
        if (FIRST)
        {
            const uint32_t end = (base + rLen) < fp.genCount ? (base + rLen) : fp.genCount;
            n = base < end ? end - base : 0u;
        }
        else if (share)
        {
            uint32_t run = 0;
#pragma unroll
            for (uint32_t k = 0; k < kRegionsPerBlock; ++k)
            {
                gF[k] = wave_uniform(ss.segFront[(size_t)bounce*ss.numRegions + r0 + k]);
                gStart[k] = run;
                run += gF[k] + wave_uniform(ss.segBack[(size_t)bounce*ss.numRegions + r0 + k]);
            }
            n = run;
        }
        else
        {
            nFront = wave_uniform(ss.segFront[(size_t)bounce*ss.numRegions + r]);
            n = nFront + wave_uniform(ss.segBack[(size_t)bounce*ss.numRegions + r]);
        }
        RegionAppend out = { base, rLen, 0u, 0u };

        uint32_t poolCount = 0;         // wave-uniform: paths that hit a surface and wait for the shading half
        for (uint32_t j0 = share ? wave*kWave : 0u; ; j0 += share ? kBlock : kWave)
        {
            // the region's rounds are done: what still waits in the pool is shaded, then the region ends
            const bool flush = j0 >= n;
            if (flush && poolCount == 0u)
                break;
            const uint32_t j = j0 + lane;
            bool have = false, alive = false, front = true;
            PathRegs p;
            uint32_t slot = 0;
            int prim = -1;
            float t = 0.0f;
            V3 n3;

            TN_TICK(4)
            if (j < n)
            {
                if (FIRST)
                {
                    if (gen_slot(fp, base + j, slot))
                    {
                        float rx, ry;
                        // the camera (21 words, read by bounce 0 only) is fetched from the kernel-argument segment HERE, by scalar loads the
                        // compiler may not hoist: as a by-value argument it sat in SGPRs (or their spill lanes) through every bounce
                        typedef const __attribute__((address_space(4))) CameraParams* CamPtr;
                        CamPtr camp = (CamPtr)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(BounceKernargs, cam));
                        asm volatile("" : "+s"(camp));
                        CameraParams camNow;
                        for (int w = 0; w < 16; ++w)
                            camNow.r2w[w] = camp->r2w[w];
                        camNow.ox = camp->ox; camNow.oy = camp->oy; camNow.oz = camp->oz;
                        camNow.shutterStart = camp->shutterStart; camNow.shutterEnd = camp->shutterEnd;
                        have = begin_path(camNow, fp, passSeeds, slot, p, rx, ry);
                        if (!have)
                            TN_RAD_OUT[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        else
                            samples++;
                    }
                }
                else
                {
                    uint32_t pos;
                    if (share)
                    {
                        uint32_t k = 0, f = gF[0], s0 = 0;
#pragma unroll
                        for (uint32_t q2 = 1; q2 < kRegionsPerBlock; ++q2)
                            if (j >= gStart[q2]) { k = q2; f = gF[q2]; s0 = gStart[q2]; }
                        pos = region_pos(base + (k - wave)*rLen, rLen, f, j - s0);     // region r0 + k of this group
                    }
                    else
                        pos = region_pos(base, rLen, nFront, j);
                    load_state(sc, TN_SS_BUF(cur), pos, p, slot, hasMedia);
                    have = true;
                }
            }

            // ---- the closest-hit trace; a ray that leaves the scene ends its path here ------------------------------------
            if (have)
            {
                TN_TICK(0)
                prim = trace<SceneT<LDS, false, DEFER ? 1 : 0>, LdsStack<kBlock>, COUNT>(sc, st, p.o, p.d, p.time, t, n3, ctr);
                rays++;
                TN_TICK(1)
                if (prim < 0)
                {
                    on_miss(late_sky(), p, bounce);
                    TN_RAD_OUT[slot] = make_float4(p.rad.x, p.rad.y, p.rad.z, 0.0f);
                    have = false;
                }
            }

            // ---- close ranks (see the pool's comment above) -------------------------------------------------------------
            if (repack)
            {
                const unsigned long long live = __ballot(have);
                const uint32_t nLive = (uint32_t)__popcll(live);
                if (flush || poolCount + nLive >= (uint32_t)kWave)
                {
                    const uint32_t take = (poolCount < (uint32_t)kWave - nLive) ? poolCount : (uint32_t)kWave - nLive;
                    const uint32_t rank = bits_below(~live);
                    if (!have && rank < take)
                    {
                        pool_load(pool, poolCount - 1u - rank, p, slot, prim, t, n3);
                        p.absorption = medium_absorption(sc, p.medium, hasMedia);
                        have = true;
                    }
                    poolCount -= take;
                }
                else
                {
                    if (have)
                        pool_store(pool, poolCount + bits_below(live), p, slot, prim, t, n3);
                    poolCount += nLive;
                    have = false;           // waits in the pool
                }
            }

            // ---- the shading half: emission, light sampling with its shadow traces, the BSDF step ----------------------------
            if (have)
            {
                {
                    const V3 n = n3;
                    const Mat mat = load_mat(sc.mats, prim);
                    HitCtx h;
                    on_hit_begin(p, mat, t, n, bounce, h, prim);

                    if (sc.totalLightSamples > 0)
                    {
                        const V3 thrAtNee = p.thr;
                        LightCursor lights;
                        V3 sum = nee_sum(sc, [&](int k) -> V3 {
                            NeeGeo g;
                            V3 skyColor;
                            float skyPdf = 0.0f;
                            int light = -1;
                            if (sc.probe.valid && k == 0)
                                nee_sample_probe(late_sky(), h.p, h.n, p.rng, g, skyColor, skyPdf);
                            else
                            {
                                light = lights.next(sc);
                                nee_sample_light(sc, h.p, h.n, p.time, light, p.rng, g);
                            }
                            TN_TICK(2)
                            float ts;
                            V3 nn;
                            const int hp = trace<SceneT<LDS, false, DEFER ? 1 : 0>, LdsStack<kBlock>, COUNT>(sc, st, g.o, g.wi, p.time, ts, nn, TN_CTR_NEE);
                            TN_TICK(3)
                            rays++;
                            shadowRays++;
                            // the BSDF terms only for the samples that arrive; the 28-register material record is re-read here
                            // instead of living across the shadow trace
                            if (light < 0)
                                return (hp < 0) ? nee_contrib_probe(load_mat(sc.mats, prim), h, g.wi, skyColor, skyPdf) : V3(0.0f);
                            if (!nee_light_reached(g, hp, ts))
                                return V3(0.0f);
                            return nee_contrib_light(sc, load_mat(sc.mats, prim), h, g.wi, g.nl, light, hp, ts);
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
                    // the next bounce, sorted: rays that meet a bounded primitive's box in front, plane-only rays at the back
                    front = !sc.sortQueues || ray_meets_bounded_prim(sc, p.o, p.d);
                else
                    TN_RAD_OUT[slot] = make_float4(p.rad.x, p.rad.y, p.rad.z, 0.0f);
            }
            const uint32_t np = out.push(alive, front);
            if (alive)
                store_state(TN_SS_BUF(nxt), np, p, slot);
            if (flush)
                break;
        }
        if (lane == 0)
        {
            ss.segFront[(size_t)(bounce + 1)*ss.numRegions + r] = out.nFront;
            ss.segBack[(size_t)(bounce + 1)*ss.numRegions + r] = out.nBack;
        }
      }
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
#undef TN_SS_BUF
#undef TN_RAD_OUT

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

// WONLY: every mesh of the scene is walked by k_walk: the kernel is the flat scan + record reads, built for more waves
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
template <bool COUNT, bool LDS, bool WONLY = false, bool MIXED = false, bool LIGHTS = WONLY>
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
template <bool COUNT, bool LDS, bool WONLY = false, bool MIXED = false>
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
#include "tn_swalk.h"
namespace tn {

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
                on_hit_begin(p, mat, t, n, bounce, h, prim);

                // SampleLights (render.cpp:103-227): draw, trace, and the BSDF terms for the samples that arrive
                {
                    const V3 thrAtNee = p.thr;
                    LightCursor lights;
                    V3 sum = nee_sum(sc, [&](int k) -> V3 {
                        NeeGeo g;
                        V3 skyColor;
                        float skyPdf = 0.0f;
                        int light = -1;
                        if (sc.probe.valid && k == 0)
                            nee_sample_probe(sc, h.p, h.n, p.rng, g, skyColor, skyPdf);
                        else
                        {
                            light = lights.next(sc);
                            nee_sample_light(sc, h.p, h.n, p.time, light, p.rng, g);
                        }
                        float ts;
                        V3 nn;
                        const int hp = trace<SceneT<LDS>, LdsStack<kBlock>, COUNT>(sc, st, g.o, g.wi, p.time, ts, nn, ctr);
                        rays++;
                        shadowRays++;
                        if (light < 0)
                            return (hp < 0) ? nee_contrib_probe(mat, h, g.wi, skyColor, skyPdf) : V3(0.0f);
                        if (!nee_light_reached(g, hp, ts))
                            return V3(0.0f);
                        return nee_contrib_light(sc, mat, h, g.wi, g.nl, light, hp, ts);
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

__global__ __launch_bounds__(kBlock, 4) void k_accumulate(PathState ps, FrameParams fp, float4* __restrict__ accum, const uint32_t* __restrict__ passSeeds)
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
                // the raster position is the first two draws of the path's own stream (camera_sample)
                Rng rng = Rng::seeded((uint32_t)i + (uint32_t)j*(uint32_t)fp.width + passSeeds[fp.passBase + s]);
                const float x = rng.randf();
                const float y = rng.randf();
                const float rx = x + i, ry = y + j;

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

// SPAN = the candidate window's edge (reachLo + reachHi + 1: 3 for the default filter width 0.75, 4 for cornell's 1.0) as a compile-time
// constant: the gather loop is unrolled over the SPAN x SPAN window with no bounds -- candidates outside the frame are staged as
// "covers nothing", so clipping the window changes nothing -- in the same raster order; 0 = the window's bounds at run time.
// THREADS = kBlock, or 2*kBlock for frames of few tiles (one wave per SIMD or less: a pass is then as long as one thread's chain, and
// the second half of the workgroup -- no pixels of its own -- takes the second staging round off it).
template <int SPAN, int THREADS = kBlock>
__global__ __launch_bounds__(THREADS, THREADS == kBlock ? 4 : 2) void k_accumulate_tiled(PathState ps, FrameParams fp, float4* __restrict__ accum,
                                                                const uint32_t* __restrict__ passSeeds, const int* __restrict__ tileList)
{
    // LDS sized by what the window can hold: for the compile-time windows (filter widths up to 1) an edge of 16 + SPAN - 1 entries and at most
    // three footprint columns / rows per path (int(r + fw) - int(r - fw) + 1 <= 3) -- 14-16 KB a workgroup instead of 26.7, so the registers
    // (six waves per SIMD) and not the LDS (five workgroups per CU) set the occupancy of a kernel that waits half of its cycles
    constexpr int kSide = SPAN > 0 ? kAccTile + SPAN - 1 : kAccSide;
    constexpr int kEntries = kSide*kSide;
    constexpr int kFoot = SPAN > 0 ? 3 : kAccMaxFoot;
    constexpr int kEnt = (kEntries + THREADS - 1)/THREADS;          // candidate entries a thread stages per pass
    // per candidate path of the tile: clamped sample, footprint [startX, startX+nX) x [startY, startY+nY)
    // and the separable Gaussian weights of its footprint columns / rows (each shared by up to 5 pixels)
    __shared__ float4 s_c[kEntries];                 // rgb, .w = bits(startX | nX << 16)
    __shared__ uint32_t s_y[kEntries];               // startY | nY << 16
    __shared__ float s_wx[kFoot][kEntries];
    __shared__ float s_wy[kFoot][kEntries];
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

    const float fw = fp.filterWidth;
    const int reachLo = 1 + (int)floorf(fw);
    const int reachHi = (int)ceilf(fw);
    const int side = kAccTile + reachLo + reachHi;
    const int ox = tx*kAccTile - reachLo, oy = ty*kAccTile - reachLo;     // frame coordinates of LDS entry (0,0)
    const bool gauss = fp.filterType != 0;

    // The (at most two) candidate entries this thread stages every pass: which path, where in LDS, whether the path
    // is this shard's.  The radiance of the NEXT pass is requested before the current pass is processed.
    int entLe[kEnt], entGx[kEnt], entGy[kEnt];
    bool entLive[kEnt], entStage[kEnt];
    float4 nextRa[kEnt];
    auto set_entry = [&](int k, int e) {
        const int ex = e % side, ey = e/side;
        entGx[k] = ox + ex; entGy[k] = oy + ey;
        entLe[k] = ey*kSide + ex;
        entLive[k] = e < side*side && entGx[k] >= 0 && entGy[k] >= 0 && entGx[k] < fp.width && entGy[k] < fp.height &&
                     pixel_owned(fp, entGx[k], entGy[k]);
    };
#pragma unroll
    for (int k = 0; k < kEnt; ++k)
    {
        set_entry(k, threadIdx.x + k*THREADS);
        entStage[k] = threadIdx.x + k*THREADS < side*side;       // (entries outside the frame are staged as "covers nothing", every pass)
    }

    // Which pixel is this thread's, which entries does it stage?  Pixel by pixel, row by row (a wave = 4 rows of the tile) and entry
    // t, t + THREADS -- unless the tile is one of a shard's HALO tiles: its candidate window reaches an owned shard tile by a pixel or
    // two, so only a strip of its entries (or a corner) is this shard's and only a strip of its pixels has any candidate of this shard,
    // in every pass (ownership is a function of the pixel).  Spread over the workgroup as above that is a lane or two of EVERY wave
    // staging and gathering: a halo tile cost 0.7 of an inner one, and with 8 shards of 64-pixel tiles 20 of a shard tile's 36
    // accumulate tiles are halo while the pass loop is 8 x as long (profiles/r05_n_shard_tile.md).  So for a shard:
    //   - the shard's own entries are handed out DENSELY (the t-th live entry to thread t): a strip is staged by one wave, and
    //     the entries that are not the shard's are marked "covers nothing" once, here;
    //   - pixels without a candidate of this shard are left alone altogether -- no gather, no load, no store -- and the tile's
    //     pixels are dealt to the threads row by row or column by column, whichever leaves fewer waves with a pixel to do.
    // A pixel's adds are its own thread's, in pass and raster order, whichever thread that is.
    int lx = threadIdx.x % kAccTile, ly = (threadIdx.x/kAccTile) % kAccTile;
    bool mine = true;
    if (tileList)
    {
        __shared__ int s_count[3];                      // waves with a pixel to do (by rows, by columns); live entries
        int* s_list = reinterpret_cast<int*>(&s_wx[0][0]);      // (the weights are written by the passes: free until then)
        if (threadIdx.x < 3)
            s_count[threadIdx.x] = 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kEnt; ++k)
        {
            const int e = threadIdx.x + k*THREADS;
            if (e < side*side)
            {
                s_y[entLe[k]] = entLive[k] ? 1u : 0u;
                s_c[entLe[k]] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            const unsigned long long live = __ballot(entLive[k]);
            int base = 0;
            if ((threadIdx.x & 63) == 0 && live != 0ull)
                base = atomicAdd(&s_count[2], __popcll(live));
            base = __builtin_amdgcn_readfirstlane(base);
            if (entLive[k])
                s_list[base + (int)bits_below(live)] = e;
        }
        __syncthreads();
        auto window_live = [&](int wx, int wy) {
            const int qx = tx*kAccTile + wx, qy = ty*kAccTile + wy;
            if (qx >= fp.width || qy >= fp.height)
                return false;
            const int a0 = maxI(0, qx - reachLo) - ox, a1 = minI(fp.width - 1, qx + reachHi) - ox;
            const int b0 = maxI(0, qy - reachLo) - oy, b1 = minI(fp.height - 1, qy + reachHi) - oy;
            uint32_t any = 0u;
            for (int j = b0; j <= b1; ++j)
                for (int i = a0; i <= a1; ++i)
                    any |= s_y[j*kSide + i];
            return any != 0u;
        };
        const int cx = ly, cy = lx;                     // the same thread, column by column
        const bool byRow = threadIdx.x < kBlock && window_live(lx, ly);
        const bool byCol = threadIdx.x < kBlock && window_live(cx, cy);
        const bool waveRow = __ballot(byRow) != 0ull, waveCol = __ballot(byCol) != 0ull;
        if ((threadIdx.x & 63) == 0)
        {
            if (waveRow) atomicAdd(&s_count[0], 1);
            if (waveCol) atomicAdd(&s_count[1], 1);
        }
        __syncthreads();
        const bool columns = s_count[1] < s_count[0];
        if (columns) { lx = cx; ly = cy; }
        mine = columns ? byCol : byRow;
        const int nLive = s_count[2];
#pragma unroll
        for (int k = 0; k < kEnt; ++k)
        {
            const int t = threadIdx.x + k*THREADS;
            entStage[k] = t < nLive;
            entLive[k] = false;
            if (entStage[k])
                set_entry(k, s_list[t]);
        }
        __syncthreads();                                // (the flags and the list are staged over by the first pass)
    }
#pragma unroll
    for (int k = 0; k < kEnt; ++k)
    {
        nextRa[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (entLive[k] && fp.accBegin < fp.accEnd)
            nextRa[k] = ps.rad[slot_of(fp, fp.accBegin, entGx[k], entGy[k])];
    }
    const int px = tx*kAccTile + lx, py = ty*kAccTile + ly;
    const bool inside = threadIdx.x < kBlock && px < fp.width && py < fp.height && mine;

    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (inside)
        acc = accum[py*fp.width + px];

    // this pixel's candidate window, in LDS coordinates (clipped to the frame like the reference's loops)
    const int i0 = maxI(0, px - reachLo) - ox, i1 = minI(fp.width - 1, px + reachHi) - ox;
    const int j0 = maxI(0, py - reachLo) - oy, j1 = minI(fp.height - 1, py + reachHi) - oy;

    for (int s = fp.accBegin; s < fp.accEnd; ++s)
    {
        float4 curRa[kEnt];
#pragma unroll
        for (int k = 0; k < kEnt; ++k)
            curRa[k] = nextRa[k];
        if (s + 1 < fp.accEnd)
#pragma unroll
            for (int k = 0; k < kEnt; ++k)
                if (entLive[k])
                    nextRa[k] = ps.rad[slot_of(fp, s + 1, entGx[k], entGy[k])];

#pragma unroll
        for (int k = 0; k < kEnt; ++k)
        {
            if (!entStage[k])
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
                    for (int kk = 0; kk < kFoot; ++kk)
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

        auto add = [&](int le) {
            const float4 c = s_c[le];
            const uint32_t xm = __float_as_uint(c.w), ym = s_y[le];
            const uint32_t kx = (uint32_t)(px - (int)(xm & 0xffffu)), ky = (uint32_t)(py - (int)(ym & 0xffffu));
            if (kx >= (xm >> 16) || ky >= (ym >> 16))
                return;
            if (!gauss)
            {
                acc.x += c.x; acc.y += c.y; acc.z += c.z; acc.w += 1.0f;
            }
            else
            {
                const float w = s_wx[kx][le]*s_wy[ky][le];
                acc.x += c.x*w; acc.y += c.y*w; acc.z += c.z*w; acc.w += w;
            }
        };
        if (inside)
        {
            if (SPAN > 0)
            {
                // the window of pixel (lx, ly) starts at LDS entry (lx, ly): px - reachLo - ox == lx
#pragma unroll
                for (int dj = 0; dj < SPAN; ++dj)
#pragma unroll
                    for (int di = 0; di < SPAN; ++di)
                        add((ly + dj)*kSide + lx + di);
            }
            else
            {
                for (int j = j0; j <= j1; ++j)
                    for (int i = i0; i <= i1; ++i)
                        add(j*kSide + i);
            }
        }
        __syncthreads();
    }

    if (inside)
        accum[py*fp.width + px] = acc;
}

// k_accumulate_piped: the same adds for launches of FEW tiles, where a tile's pass loop -- stage the pass's candidates, barrier, gather,
// barrier, one pass after the other -- is what the launch lasts: a small frame (a wave per SIMD or less), or a shard of N, whose pass
// loop is N x as long over 1/N of the tiles (8 shards of cornell 1024^2: 160 passes x 3.5 us whatever else was changed; calls o-p).
// Ten waves per tile: waves 4-9 stage pass s + 1 into one half of a double buffer while waves 0-3 gather pass s from the other; one
// barrier per pass, and a pass lasts as long as the longer of the two instead of their sum.  A pixel's adds are still one thread's, in
// pass and raster order.  Entries that cover nothing in any pass (outside the frame, another shard's) are marked so once, in both
// halves; a shard's halo tiles are handled as in k_accumulate_tiled (own entries dense, pixels without a candidate left alone, rows or
// columns).
constexpr int kAccPipeStagers = 384;        // >= 19 x 19 entries (filter widths up to 1): one entry per staging thread
constexpr int kAccPipeThreads = kBlock + kAccPipeStagers;

template <int SPAN>
__global__ __launch_bounds__(kAccPipeThreads, 2) void k_accumulate_piped(PathState ps, FrameParams fp, float4* __restrict__ accum,
                                                                     const uint32_t* __restrict__ passSeeds, const int* __restrict__ tileList)
{
    constexpr int kEnt = (kAccEntries + kAccPipeStagers - 1)/kAccPipeStagers;
    // footprint columns / rows a path can have: int(r + fw) - int(r - fw) + 1 <= 3 for the filter widths of SPAN 3 and 4 (fw <= 1)
    constexpr int kFoot = SPAN > 0 ? 3 : kAccMaxFoot;
    __shared__ float4 s_c[2][kAccEntries];              // rgb, .w = bits(startX | nX << 16)
    __shared__ uint32_t s_y[2][kAccEntries];            // startY | nY << 16
    __shared__ float s_wx[2][kFoot][kAccEntries];
    __shared__ float s_wy[2][kFoot][kAccEntries];
    __shared__ unsigned long long s_exp[32];
    __shared__ int s_count[3];                          // waves with a pixel to do (by rows, by columns); live entries
    if (threadIdx.x < 32)
        s_exp[threadIdx.x] = kExp2fTab[threadIdx.x];
    if (threadIdx.x < 3)
        s_count[threadIdx.x] = 0;

    const int tilesX = (fp.width + kAccTile - 1)/kAccTile;
    const int tile = tileList ? tileList[blockIdx.x] : (int)blockIdx.x;
    const int tx = tile % tilesX, ty = tile/tilesX;
    const float fw = fp.filterWidth;
    const int reachLo = 1 + (int)floorf(fw);
    const int reachHi = (int)ceilf(fw);
    const int side = kAccTile + reachLo + reachHi;
    const int ox = tx*kAccTile - reachLo, oy = ty*kAccTile - reachLo;     // frame coordinates of LDS entry (0,0)
    const bool gauss = fp.filterType != 0;
    const bool stager = threadIdx.x >= kBlock;
    const int sid = (int)threadIdx.x - kBlock;          // stagers: 0 .. kAccPipeStagers - 1

    // every thread marks entries "cover nothing" in both halves and flags the live ones; the live entries are listed densely
    int* s_list = reinterpret_cast<int*>(&s_wx[1][0][0]);      // (free until the second pass is staged)
    for (int e = threadIdx.x; e < kAccEntries; e += kAccPipeThreads)
    {
        s_c[0][e] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        s_c[1][e] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        s_y[0][e] = 0u;
        s_y[1][e] = 0u;
    }
    __syncthreads();
    for (int e0 = 0; e0 < side*side; e0 += kAccPipeThreads)
    {
        const int e = e0 + (int)threadIdx.x;
        const int ex = e % side, ey = e/side;
        const int gx = ox + ex, gy = oy + ey;
        const bool live = e < side*side && gx >= 0 && gy >= 0 && gx < fp.width && gy < fp.height && pixel_owned(fp, gx, gy);
        const unsigned long long m = __ballot(live);
        int base = 0;
        if ((threadIdx.x & 63) == 0 && m != 0ull)
            base = atomicAdd(&s_count[2], __popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        if (live)
        {
            s_list[base + (int)bits_below(m)] = e;
            s_y[1][ey*kAccSide + ex] = 1u;              // (the flag: read below, staged over by the second pass)
        }
    }
    __syncthreads();
    const int nLive = s_count[2];

    // gatherers: which pixel (k_accumulate_tiled: rows or columns, whichever leaves fewer waves with a pixel that has a candidate)
    int lx = threadIdx.x % kAccTile, ly = (threadIdx.x/kAccTile) % kAccTile;
    bool mine = false;
    {
        auto window_live = [&](int wx, int wy) {
            const int qx = tx*kAccTile + wx, qy = ty*kAccTile + wy;
            if (qx >= fp.width || qy >= fp.height)
                return false;
            const int a0 = maxI(0, qx - reachLo) - ox, a1 = minI(fp.width - 1, qx + reachHi) - ox;
            const int b0 = maxI(0, qy - reachLo) - oy, b1 = minI(fp.height - 1, qy + reachHi) - oy;
            uint32_t any = 0u;
            for (int j = b0; j <= b1; ++j)
                for (int i = a0; i <= a1; ++i)
                    any |= s_y[1][j*kAccSide + i];
            return any != 0u;
        };
        const int cx = ly, cy = lx;
        const bool byRow = !stager && window_live(lx, ly);
        const bool byCol = !stager && window_live(cx, cy);
        const bool waveRow = __ballot(byRow) != 0ull, waveCol = __ballot(byCol) != 0ull;
        if ((threadIdx.x & 63) == 0)
        {
            if (waveRow) atomicAdd(&s_count[0], 1);
            if (waveCol) atomicAdd(&s_count[1], 1);
        }
        __syncthreads();
        const bool columns = s_count[1] < s_count[0];
        if (columns) { lx = cx; ly = cy; }
        mine = columns ? byCol : byRow;
    }
    // stagers: which entries
    int entLe[kEnt], entGx[kEnt], entGy[kEnt];
    bool entLive[kEnt];
    float4 nextRa[kEnt];
#pragma unroll
    for (int k = 0; k < kEnt; ++k)
    {
        const int t = sid + k*kAccPipeStagers;
        entLive[k] = stager && t < nLive;
        entLe[k] = 0; entGx[k] = 0; entGy[k] = 0;
        if (entLive[k])
        {
            const int e = s_list[t];
            const int ex = e % side, ey = e/side;
            entGx[k] = ox + ex; entGy[k] = oy + ey;
            entLe[k] = ey*kAccSide + ex;
        }
        nextRa[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (entLive[k] && fp.accBegin < fp.accEnd)
            nextRa[k] = ps.rad[slot_of(fp, fp.accBegin, entGx[k], entGy[k])];
    }
    __syncthreads();                                    // (flags and list read by everyone)
    for (int e = threadIdx.x; e < kAccEntries; e += kAccPipeThreads)
        s_y[1][e] = 0u;                                 // the flags go; the barrier of the first staging orders this before any write of half 1

    const int px = tx*kAccTile + lx, py = ty*kAccTile + ly;
    const bool inside = !stager && mine;                // (window_live: inside the frame)
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (inside)
        acc = accum[py*fp.width + px];
    const int i0 = maxI(0, px - reachLo) - ox, i1 = minI(fp.width - 1, px + reachHi) - ox;
    const int j0 = maxI(0, py - reachLo) - oy, j1 = minI(fp.height - 1, py + reachHi) - oy;

    // stage pass s into half h (stagers)
    auto stage = [&](int s, int h) {
        float4 curRa[kEnt];
#pragma unroll
        for (int k = 0; k < kEnt; ++k)
            curRa[k] = nextRa[k];
        if (s + 1 < fp.accEnd)
#pragma unroll
            for (int k = 0; k < kEnt; ++k)
                if (entLive[k])
                    nextRa[k] = ps.rad[slot_of(fp, s + 1, entGx[k], entGy[k])];
#pragma unroll
        for (int k = 0; k < kEnt; ++k)
        {
            if (!entLive[k])
                continue;
            const int gx = entGx[k], gy = entGy[k], le = entLe[k];
            // (k_accumulate_tiled's staging, expression for expression)
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
            if (gauss)
            {
                for (int kk = 0; kk < kFoot; ++kk)
                {
                    if (kk < nX)
                        s_wx[h][kk][le] = filter_gauss_tab((startX + kk) - rx, fp.filterFalloff, fp.filterOffset, s_exp);
                    if (kk < nY)
                        s_wy[h][kk][le] = filter_gauss_tab((startY + kk) - ry, fp.filterFalloff, fp.filterOffset, s_exp);
                }
            }
            s_c[h][le] = make_float4(cl.x, cl.y, cl.z, __uint_as_float((uint32_t)startX | (uint32_t)nX << 16));
            s_y[h][le] = (uint32_t)startY | (uint32_t)nY << 16;
        }
    };

    __syncthreads();
    if (stager && fp.accBegin < fp.accEnd)
        stage(fp.accBegin, 0);
    __syncthreads();

    for (int s = fp.accBegin; s < fp.accEnd; ++s)
    {
        const int h = (s - fp.accBegin) & 1;
        if (stager)
        {
            if (s + 1 < fp.accEnd)
                stage(s + 1, h ^ 1);
        }
        else if (inside)
        {
            auto add = [&](int le) {
                const float4 c = s_c[h][le];
                const uint32_t xm = __float_as_uint(c.w), ym = s_y[h][le];
                const uint32_t kx = (uint32_t)(px - (int)(xm & 0xffffu)), ky = (uint32_t)(py - (int)(ym & 0xffffu));
                if (kx >= (xm >> 16) || ky >= (ym >> 16))
                    return;
                if (!gauss)
                {
                    acc.x += c.x; acc.y += c.y; acc.z += c.z; acc.w += 1.0f;
                }
                else
                {
                    const float w = s_wx[h][kx][le]*s_wy[h][ky][le];
                    acc.x += c.x*w; acc.y += c.y*w; acc.z += c.z*w; acc.w += w;
                }
            };
            if (SPAN > 0)
            {
#pragma unroll
                for (int dj = 0; dj < SPAN; ++dj)
#pragma unroll
                    for (int di = 0; di < SPAN; ++di)
                        add((ly + dj)*kAccSide + lx + di);
            }
            else
            {
                for (int j = j0; j <= j1; ++j)
                    for (int i = i0; i <= i1; ++i)
                        add(j*kAccSide + i);
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
// k_normals: eNormals mode of the CPU renderer (render.cpp:494-515): x=i, y=j, time 1, overwrite.  The mode has no weights
// to divide by (main.cpp:256 shows the buffer as it is), so a shard writes ZERO at the pixels it does not own: the sum over
// the ranks of a group (one reduce, like the path-traced sum) is then the image, not N times it.

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
    if (!pixel_owned(fp, i, j))
    {
        accum[pix] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        return;
    }

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
