// tn_kernels.h -- the gfx950 kernels.
//
// Streaming (wavefront) pipeline, one batch of B = pixels x passes path slots:
//
//   k_generate                      camera rays + path state  -> HBM, bounce-0 queue
//   for bounce in 0..maxDepth-1:
//     k_extend   (queue[bounce])    closest-hit traversal           reads ray, writes hit
//     k_shade    (queue[bounce])    emission/MIS, NEE sample records, BSDF sample -> next ray;
//                                   wave64 ballot compaction into queue[bounce+1] and the NEE queue
//     k_shadow   (neeQueue[bounce]) NEE visibility rays, resolves direct light into the path radiance
//   k_accumulate                    filter-footprint GATHER into the float4 accumulator (no atomics,
//                                   bit-reproducible, same summation order as render.cpp:401-445)
//
// All trace kernels are persistent: a fixed grid whose waves pull 64-entry chunks from the
// queue with one atomic per wave; traversal stacks live in LDS as stack[entry][lane].
#pragma once

#include "tn_integrator.h"

namespace tn {

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
    uint32_t* activeCount;  // [maxDepth+1]
    uint32_t* neeCount;     // [maxDepth]
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
    int maxDepth;
    int shardRank, shardWorld, shardTile;
    int filterType;
    float filterWidth, filterFalloff, filterOffset;
    float clampLen;
};

// ---------------------------------------------------------------------------
// wave-level helpers

TN_D int lane_id() { return (int)__lane_id(); }

// Order-preserving-within-wave compaction: one atomic per wave.  Must be called by all lanes.
TN_D uint32_t wave_enqueue(bool pred, uint32_t* counter)
{
    const unsigned long long mask = __ballot(pred);
    uint32_t base = 0;
    if (mask)
    {
        const int leader = __ffsll((long long)mask) - 1;
        const int lane = lane_id();
        if (lane == leader)
            base = atomicAdd(counter, (uint32_t)__popcll(mask));
        base = __shfl(base, leader);
        const uint32_t prefix = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        return base + prefix;
    }
    return 0;
}

// One wave grabs the next 64 queue entries.
TN_D uint32_t wave_fetch(uint32_t* cursor)
{
    uint32_t base = 0;
    if (lane_id() == 0)
        base = atomicAdd(cursor, (uint32_t)kWave);
    return __shfl(base, 0);
}

TN_D void wave_add_stat(unsigned long long* dst, uint32_t v)
{
    // wave reduction, one atomic per wave
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off);
    if (lane_id() == 0 && v)
        atomicAdd(dst, (unsigned long long)v);
}

TN_D bool pixel_owned(const FrameParams& fp, int i, int j)
{
    if (fp.shardWorld <= 1)
        return true;
    const int tilesX = (fp.width + fp.shardTile - 1)/fp.shardTile;
    const int t = (j/fp.shardTile)*tilesX + (i/fp.shardTile);
    return (t % fp.shardWorld) == fp.shardRank;
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
// k_generate

__global__ __launch_bounds__(kBlock) void k_generate(PathState ps, QueueCtl q, uint32_t* queue0, CameraParams cam, FrameParams fp,
                                                     const uint32_t* __restrict__ passSeeds)
{
    const int npix = fp.width*fp.height;
    const int total = npix*fp.numPasses;
    const int slot = blockIdx.x*kBlock + threadIdx.x;
    bool live = false;

    if (slot < total)
    {
        const int s = slot/npix;
        const int pix = slot - s*npix;
        const int j = pix/fp.width;
        const int i = pix - j*fp.width;

        if (pixel_owned(fp, i, j))
        {
            Rng rng;
            float rx, ry, time;
            V3 o, d;
            camera_sample(cam, fp, i, j, passSeeds[fp.passBase + s], rng, rx, ry, time, o, d);

            ps.rayO[slot] = make_float4(o.x, o.y, o.z, time);
            ps.rayD[slot] = make_float4(d.x, d.y, d.z, 1.0f);
            ps.thr[slot] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
            ps.rad[slot] = make_float4(0.0f, 0.0f, 0.0f, __int_as_float((int)kReflected));
            ps.absorb[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            ps.rngRaster[slot] = make_float4(__uint_as_float(rng.s1), __uint_as_float(rng.s2), rx, ry);
            live = true;
        }
        else
        {
            ps.rad[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            ps.rngRaster[slot] = make_float4(0.0f, 0.0f, -1e30f, -1e30f);
        }
    }

    const uint32_t at = wave_enqueue(live, q.activeCount + 0);
    if (live)
        queue0[at] = (uint32_t)slot;

    uint32_t one = live ? 1u : 0u;
    wave_add_stat(q.stats + 1, one);
}

// ---------------------------------------------------------------------------
// k_extend: closest hit for every queued path

template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_extend(DevScene sc, PathState ps, QueueCtl q, const uint32_t* __restrict__ queue, int bounce)
{
    extern __shared__ uint32_t s_stack[];      // [stackEntries][kBlock], sized at launch
    LdsStack<kBlock> st = { s_stack + threadIdx.x };

    const uint32_t count = q.activeCount[bounce];
    uint32_t rays = 0;
    TraceCounters ctr = { 0, 0, 0 };

    for (;;)
    {
        const uint32_t base = wave_fetch(q.cursorExtend + bounce);
        if (base >= count)
            break;
        const uint32_t idx = base + lane_id();
        if (idx < count)
        {
            const uint32_t slot = queue[idx];
            const float4 ro = ps.rayO[slot];
            const float4 rd = ps.rayD[slot];

            float t;
            V3 n;
            const int prim = trace<LdsStack<kBlock>, COUNT>(sc, st, V3(ro.x, ro.y, ro.z), V3(rd.x, rd.y, rd.z), ro.w, t, n, ctr);

            ps.hit[slot] = make_float4(t, n.x, n.y, n.z);
            ps.hitPrim[slot] = prim;
            rays++;
        }
    }

    wave_add_stat(q.stats + 0, rays);
    if (COUNT)
    {
        wave_add_stat(q.stats + 2, ctr.internal);
        wave_add_stat(q.stats + 3, ctr.tris);
        wave_add_stat(q.stats + 4, ctr.prims);
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

__global__ __launch_bounds__(kBlock) void k_shade(DevScene sc, PathState ps, QueueCtl q, const uint32_t* __restrict__ queue,
                                                  uint32_t* __restrict__ queueNext, uint32_t* __restrict__ queueNee, int bounce, int maxDepth)
{
    const uint32_t count = q.activeCount[bounce];

    for (;;)
    {
        const uint32_t base = wave_fetch(q.cursorShade + bounce);
        if (base >= count)
            break;
        const uint32_t idx = base + lane_id();

        bool wantNee = false;
        bool wantNext = false;
        uint32_t slot = 0;

        if (idx < count)
        {
            slot = queue[idx];

            PathRegs p;
            {
                const float4 ro = ps.rayO[slot], rd = ps.rayD[slot], th = ps.thr[slot], ra = ps.rad[slot];
                const float4 ab = ps.absorb[slot], rr = ps.rngRaster[slot];
                p.o = V3(ro.x, ro.y, ro.z); p.time = ro.w;
                p.d = V3(rd.x, rd.y, rd.z); p.bsdfPdf = rd.w;
                p.thr = V3(th.x, th.y, th.z); p.eta = th.w;
                p.rad = V3(ra.x, ra.y, ra.z); p.rayType = __float_as_int(ra.w);
                p.absorption = V3(ab.x, ab.y, ab.z);
                p.rng.s1 = __float_as_uint(rr.x); p.rng.s2 = __float_as_uint(rr.y);
            }

            const int prim = ps.hitPrim[slot];
            if (prim < 0)
            {
                on_miss(sc, p, bounce);
                ps.rad[slot] = make_float4(p.rad.x, p.rad.y, p.rad.z, __int_as_float(p.rayType));
            }
            else
            {
                const float4 hh = ps.hit[slot];
                const Mat mat = load_mat(sc.mats, prim);

                HitCtx h;
                on_hit_begin(p, mat, hh.x, V3(hh.y, hh.z, hh.w), bounce, h);

                // SampleLights, part 1 (render.cpp:107-170): consume the RNG, emit shadow-ray records
                int k = 0;
                if (sc.probe.valid)
                {
                    NeeRec r;
                    nee_prepare_probe(sc, mat, h, p.rng, r);
                    store_nee(ps, slot, k++, r);
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
                    }
                }
                if (k > 0)
                {
                    ps.neeThr[slot] = make_float4(p.thr.x, p.thr.y, p.thr.z, 0.0f);
                    wantNee = true;
                }

                // the last iteration's BSDF sample is never used by the oracle's loop (render.cpp:250)
                int res = kTerminate;
                if (bounce + 1 < maxDepth)
                    res = bsdf_step(p, mat, h);
                wantNext = (res == kContinue);

                ps.rad[slot] = make_float4(p.rad.x, p.rad.y, p.rad.z, __int_as_float(p.rayType));
                if (wantNext)
                {
                    const float4 rr = ps.rngRaster[slot];
                    ps.rayO[slot] = make_float4(p.o.x, p.o.y, p.o.z, p.time);
                    ps.rayD[slot] = make_float4(p.d.x, p.d.y, p.d.z, p.bsdfPdf);
                    ps.thr[slot] = make_float4(p.thr.x, p.thr.y, p.thr.z, p.eta);
                    ps.absorb[slot] = make_float4(p.absorption.x, p.absorption.y, p.absorption.z, 0.0f);
                    ps.rngRaster[slot] = make_float4(__uint_as_float(p.rng.s1), __uint_as_float(p.rng.s2), rr.z, rr.w);
                }
            }
        }

        const uint32_t atNee = wave_enqueue(wantNee, q.neeCount + bounce);
        if (wantNee)
            queueNee[atNee] = slot;
        const uint32_t atNext = wave_enqueue(wantNext, q.activeCount + bounce + 1);
        if (wantNext)
            queueNext[atNext] = slot;
    }
}

// ---------------------------------------------------------------------------
// k_shadow: SampleLights, part 2 (render.cpp:118-139, 171-224): one thread per path resolves
// its K shadow rays in the oracle's order, then totalRadiance += pathThroughput*sum (render.cpp:314)

template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_shadow(DevScene sc, PathState ps, QueueCtl q, const uint32_t* __restrict__ queueNee, int bounce)
{
    extern __shared__ uint32_t s_stack[];      // [stackEntries][kBlock], sized at launch
    LdsStack<kBlock> st = { s_stack + threadIdx.x };

    const uint32_t count = q.neeCount[bounce];
    uint32_t rays = 0;
    TraceCounters ctr = { 0, 0, 0 };

    for (;;)
    {
        const uint32_t base = wave_fetch(q.cursorShadow + bounce);
        if (base >= count)
            break;
        const uint32_t idx = base + lane_id();
        if (idx < count)
        {
            const uint32_t slot = queueNee[idx];
            const float time = ps.rayO[slot].w;     // rayTime never changes along a path

            V3 sum = nee_sum(sc, [&](int k) -> V3 {
                const NeeRec r = load_nee(ps, slot, k);
                float t;
                V3 n;
                const int hp = trace<LdsStack<kBlock>, COUNT>(sc, st, r.o, r.wi, time, t, n, ctr);
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

    wave_add_stat(q.stats + 0, rays);
    wave_add_stat(q.stats + 5, rays);
    if (COUNT)
    {
        wave_add_stat(q.stats + 2, ctr.internal);
        wave_add_stat(q.stats + 3, ctr.tris);
        wave_add_stat(q.stats + 4, ctr.prims);
    }
}

// ---------------------------------------------------------------------------
// k_mega: the A/B arm -- one lane walks one whole path (render.cpp:230-388), same pieces.

template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_mega(DevScene sc, PathState ps, QueueCtl q, CameraParams cam, FrameParams fp,
                                                 const uint32_t* __restrict__ passSeeds)
{
    extern __shared__ uint32_t s_stack[];      // [stackEntries][kBlock], sized at launch
    LdsStack<kBlock> st = { s_stack + threadIdx.x };

    const int npix = fp.width*fp.height;
    const int total = npix*fp.numPasses;
    const int slot = blockIdx.x*kBlock + threadIdx.x;
    uint32_t rays = 0, shadowRays = 0, samples = 0;
    TraceCounters ctr = { 0, 0, 0 };

    if (slot < total)
    {
        const int s = slot/npix;
        const int pix = slot - s*npix;
        const int j = pix/fp.width;
        const int i = pix - j*fp.width;

        if (!pixel_owned(fp, i, j))
        {
            ps.rad[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            ps.rngRaster[slot] = make_float4(0.0f, 0.0f, -1e30f, -1e30f);
        }
        else
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
                const int prim = trace<LdsStack<kBlock>, COUNT>(sc, st, p.o, p.d, p.time, t, n, ctr);
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
                        const int hp = trace<LdsStack<kBlock>, COUNT>(sc, st, r.o, r.wi, p.time, ts, nn, ctr);
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
            }

            ps.rad[slot] = make_float4(p.rad.x, p.rad.y, p.rad.z, 0.0f);
            ps.rngRaster[slot] = make_float4(0.0f, 0.0f, rx, ry);
        }
    }

    wave_add_stat(q.stats + 0, rays);
    wave_add_stat(q.stats + 1, samples);
    wave_add_stat(q.stats + 5, shadowRays);
    if (COUNT)
    {
        wave_add_stat(q.stats + 2, ctr.internal);
        wave_add_stat(q.stats + 3, ctr.tris);
        wave_add_stat(q.stats + 4, ctr.prims);
    }
}

// ---------------------------------------------------------------------------
// k_accumulate: CpuRenderer::AddSample (render.cpp:401-445) as a gather.
// Pixel (px,py) visits the paths generated at pixels (i,j) in raster order, pass by pass, and
// adds the ones whose splat footprint [int(x-fw), int(x+fw)] x [int(y-fw), int(y+fw)] covers it --
// exactly the adds, in exactly the order, the serial oracle performs on that pixel.

TN_D float filter_gauss(float x, float falloff, float offset)      // Filter::Gaussian (render.h:29-32)
{
    return maxT(0.0f, float(m_expf(-falloff*x*x)) - offset);
}

__global__ __launch_bounds__(kBlock) void k_accumulate(PathState ps, FrameParams fp, float4* __restrict__ accum)
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

    for (int s = 0; s < fp.numPasses; ++s)
    {
        const size_t passBase = (size_t)s*npix;
        for (int j = j0; j <= j1; ++j)
        {
            for (int i = i0; i <= i1; ++i)
            {
                const size_t slot = passBase + (size_t)j*fp.width + i;
                const float4 rr = ps.rngRaster[slot];
                const float rx = rr.z, ry = rr.w;
                if (rx < -1e29f)
                    continue;       // path not generated by this shard

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

// ---------------------------------------------------------------------------
// k_normals: eNormals mode of the CPU renderer (render.cpp:494-515): x=i, y=j, time 1, overwrite.

__global__ __launch_bounds__(kBlock) void k_normals(DevScene sc, CameraParams cam, FrameParams fp, float4* __restrict__ accum)
{
    extern __shared__ uint32_t s_stack[];      // [stackEntries][kBlock], sized at launch
    LdsStack<kBlock> st = { s_stack + threadIdx.x };

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
    const int prim = trace<LdsStack<kBlock>, false>(sc, st, o, d, 1.0f, t, n, ctr);
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

} // namespace tn
