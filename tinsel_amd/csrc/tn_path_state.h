// tn_path_state.h -- what every pipeline's kernels share: wave budgets, the frame / camera parameter blocks, wave-level helpers, the scene
// arena's staging into LDS, path slots <-> (pass, pixel), the camera sample, and the DENSE path state of the wavefront pipelines
// (SplitState: regions packed at both ends, RegionAppend, load_state / store_state).
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
template <bool LDS, int WONLY, int DEFER, bool MIXED>
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
    sc.lights = reinterpret_cast<const LightRec*>(rebase(in.lights));
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
    // the PAIRED pipeline's hand-over records (tn_paired.h), by position like the state and double-buffered with it: a path's K pending light samples
    float4* pairThr[2];     // [bounce & 1][position] throughput when the samples were drawn (xyz), sample 0's |dot(wi, n)|
    float4* pairRay[2];     // [(k*2 + {0, 1})*capacity + position] = {o, dist} {wi, nl}: the shadow rays (k_walk's `nee` in mixed mode)
    float4* pairPend[2];    // [(k*2)*capacity + position] = {f.xyz, bsdfPdf} (probe sample: its contribution if unoccluded); [(k*2 + 1)*capacity + position].x = |dot(wi, n)| of sample k >= 1
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

} // namespace tn
