// tn_fused.h -- the FUSED wavefront pipeline (scenes staged whole into LDS: cornell, veach, gloss, env_loft, features): k_bounce, one launch
// over all bounces of a batch, and its per-wave shading pools.
#pragma once

#include "tn_path_state.h"

namespace tn {

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


#undef TN_SS_BUF
#undef TN_RAD_OUT

} // namespace tn
