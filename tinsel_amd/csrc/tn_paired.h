// tn_paired.h -- the PAIRED wavefront pipeline (TINSEL_PIPELINE_WAVEFRONT_PAIRED): scenes with meshes in HBM, ONE mesh walk and ONE streaming
// kernel per bounce.
//
// The split pipeline (tn_split.h) cuts a bounce of the oracle's loop (render.cpp:250-385) where its dependencies are: closest hit ->
// light samples -> shadow traces -> contributions, BSDF step -- five kernels and two k_walk launches per bounce, every one of them streaming
// the paths' records, every k_walk launch ending in a drain as long as its longest ray (profiles/r06_f_ab_walk_grid.md).  But the BSDF step
// does not depend on the shadow rays' fate: once a path has its hit, BOTH its shadow rays of this bounce and its extension ray of the next are
// known.  So here a bounce is
//
//     k_walk   the mesh walks of  { shadow rays of bounce b-1 }  and  { extension rays of bounce b }  in ONE launch (tn_walk.h, `mixed`)
//     k_step   per path: (1) RESOLVE the pending light samples of bounce b-1 -- scan + walk records for each shadow ray, then the few
//              operations that depend on its fate (nee_combine_light) on BSDF terms evaluated where the sample was drawn --,
//              (2) closest hit of the extension ray (scan + walk records), emission, medium, (3) draw this bounce's light samples and their
//              BSDF terms, the BSDF step -> the survivor, with its K pending samples and its next ray, to its new position.
//
// The same pure functions as every other pipeline (tn_integrator.h), per path in the oracle's order: the light samples' draws before the
// BSDF sample's (one RNG stream), `totalRadiance += throughput_at_sampling * sum` before the next bounce's emission is added.  A path whose
// loop has ended (light hit, last bounce, zero pdf) but whose last light samples are still pending lives ONE step longer as a resolve-only
// path.  Positions, regions, front / back packing (front: ANY of the path's K + 1 rays enters a walked mesh's box) are the split pipeline's.
#pragma once

#include "tn_split.h"

namespace tn {

// in the state's rayType word (BsdfType is 0..2): what the stored path still has to do
constexpr int kStepHasExt = 1 << 8;         // an extension ray: the oracle's loop goes on
constexpr int kStepHasNee = 1 << 9;         // K pending light samples of the bounce that wrote the path
constexpr int kStepTypeMask = 0xff;

#ifndef TN_WAVES_STEP
#define TN_WAVES_STEP 4
#endif

TN_D bool g_is_probe(float dist) { return dist < 0.0f; }     // NeeGeo::dist < 0 marks the probe sample (tn_integrator.h)

// records per position of bounce `bounce`'s walk: bounce 0 has extension rays only, the step after the last bounce shadow rays only
TN_D uint32_t paired_per(int bounce, int maxDepth, int K) { return bounce == 0 ? 1u : bounce >= maxDepth ? (uint32_t)K : (uint32_t)K + 1u; }

template <bool LDS, int WONLY, bool MIXED>
__global__ __launch_bounds__(kBlock, TN_WAVES_STEP) void k_step(DevScene scIn, SplitState ss, QueueCtl q, int bounce, int maxDepth, int rrStart, int stackEntries,
                                                                const float4* __restrict__ walkRec, uint32_t walkPrims, BinPrims bp, const uint32_t* __restrict__ order)
{
    extern __shared__ uint32_t s_stack[];      // [stackEntries][kBlock], sized at launch
    LdsStack<kBlock> st = { s_stack + threadIdx.x };
    typedef SceneT<LDS, WONLY, 2, MIXED> SC;
    SC sc;
    stage_scene_lds(sc, scIn, s_stack + stackEntries*kBlock + kScanWords);
    sc.walkRec = walkRec;

    const uint32_t lane = __lane_id();
    const int cur = bounce & 1, nxt = cur ^ 1;
    const int K = ss.neePerPath;
    const uint32_t per = paired_per(bounce, maxDepth, K);
    const bool fresh = bounce == 0;
    const bool hasMedia = sc.hasMedia != 0;
    const size_t cap = ss.capacity;
    uint32_t rays = 0, shadowRays = 0;
    TraceCounters ctr = { 0, 0, 0 };

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
            const bool valid = j < n;
            const uint32_t pos = region_pos(rBase, rLen, nFront, valid ? j : 0u);
            PathRegs p;
            uint32_t slot = 0;
            int flags = 0;
            if (valid)
            {
                if (fresh)
                {
                    // k_generate wrote the ray and the RNG; the rest of a fresh path's state is path_begin's constants (render.cpp:233-248)
                    const uint32_t at = sidx(pos);
                    const float4 ro = ss.rayO[0][at], rd = ss.rayD[0][at], rr = ss.rngId[0][at];
                    Rng rng;
                    rng.s1 = __float_as_uint(rr.x); rng.s2 = __float_as_uint(rr.y);
                    path_begin(p, V3(ro.x, ro.y, ro.z), V3(rd.x, rd.y, rd.z), ro.w, rng);
                    slot = __float_as_uint(rr.z);
                    flags = kStepHasExt;
                }
                else
                {
                    load_state(sc, ss, cur, pos, p, slot, hasMedia);
                    flags = p.rayType & ~kStepTypeMask;
                    p.rayType &= kStepTypeMask;
                }
            }

            // ---- (1) the light samples of the previous bounce: does each reach its light?  (render.cpp:117-118, 172-196 and what follows them)
            if (valid && (flags & kStepHasNee))
            {
                const float4* thrp = ss.pairThr[cur] + pos;     // throughput when the samples were drawn; .w: sample 0's |dot(wi, n)|
                const float4* rayp = ss.pairRay[cur] + pos;
                const float4* pendp = ss.pairPend[cur] + pos;
                const float time = p.time;
                LightCursor lights;                             // which light a sample belongs to is its index's: recomputed, not carried
                V3 sum = nee_sum(sc, [&](int k) -> V3 {
                    const float4 a = rayp[(size_t)(k*2)*cap], bb = rayp[(size_t)(k*2 + 1)*cap];
                    const float4 pa = pendp[(size_t)(k*2)*cap];
                    const int light = (g_is_probe(a.w)) ? -1 : lights.next(sc);
                    NeeGeo g;
                    g.o = V3(a.x, a.y, a.z); g.dist = a.w;
                    g.wi = V3(bb.x, bb.y, bb.z); g.nl = bb.w;
                    float ts;
                    V3 nn;
                    sc.walkItem = (pos*per + (uint32_t)k)*walkPrims;
                    // (a walk stops at an occluder that decides the sample: shadow_stop, tn_isect.h -- like k_shadow's)
                    const int hp = trace<SC, LdsStack<kBlock>, false, true>(sc, st, g.o, g.wi, time, ts, nn, ctr, shadow_stop(g.dist));
                    rays++;
                    shadowRays++;
                    if (g.dist < 0.0f)
                        return (hp < 0) ? V3(pa.x, pa.y, pa.z) : V3(0.0f);       // probe sample: its whole contribution was known when it was drawn
                    if (!nee_light_reached(g, hp, ts))
                        return V3(0.0f);
                    NeeTerms e;
                    e.f = V3(pa.x, pa.y, pa.z); e.bsdfPdf = pa.w;
                    // (sample 0's rides in the throughput record's spare word; read here, not held in a register across the traces)
                    e.absDot = k == 0 ? reinterpret_cast<const float*>(thrp)[3] : pendp[(size_t)(k*2 + 1)*cap].x;
                    return nee_combine_light(sc, e, g.nl, light, hp, ts);
                });
                const float4 tn4 = *thrp;
                p.rad = p.rad + V3(tn4.x, tn4.y, tn4.z)*sum;
            }

            // ---- (2) this bounce: closest hit, emission, medium (render.cpp:253-310) ------------------------------------------------
            int newFlags = 0;
            int prim = -1;
            HitCtx h;
            NeeGeo g0;                      // the first light sample stays in registers across the append
            V3 sky0;
            float skyPdf0 = 0.0f;
            Rng rngAfter0;                  // the stream behind the first sample: where the re-draw of samples 1.. starts
            LightCursor lightsAfter0;
            V3 thrNee;
            bool front = bp.count == 0;     // no walked mesh: everything goes to the front
            if (valid && (flags & kStepHasExt))
            {
                float t;
                V3 n3;
                sc.walkItem = (pos*per + (per - 1u))*walkPrims;     // the extension ray's records are the position's last
                prim = trace<SC, LdsStack<kBlock>, false>(sc, st, p.o, p.d, p.time, t, n3, ctr);
                rays++;
                if (prim < 0)
                    on_miss(sc, p, bounce);
                else
                {
                    const Mat mat = load_mat(sc.mats, prim);
                    on_hit_begin(p, mat, t, n3, bounce, h, prim);

                    // ---- (3a) SampleLights' draws (render.cpp:107-116, 158-170): every sample advances the path's stream NOW, before the
                    // BSDF sample's draws; sample 0 is kept, the others are drawn once to advance the stream and to see where their rays go,
                    // and drawn again from a copy of the stream when the path has its new position
                    if (K > 0)
                    {
                        newFlags |= kStepHasNee;
                        thrNee = p.thr;
                        LightCursor lights;
                        if (sc.probe.valid)
                            nee_sample_probe(sc, h.p, h.n, p.rng, g0, sky0, skyPdf0);
                        else
                            nee_sample_light(sc, h.p, h.n, p.time, lights.next(sc), p.rng, g0);
                        front = front || ray_enters_big_mesh(sc.primBoxes, bp, g0.o, g0.wi);
                        rngAfter0 = p.rng;
                        lightsAfter0 = lights;
                        for (int k = 1; k < K; ++k)
                        {
                            NeeGeo g;
                            nee_sample_light(sc, h.p, h.n, p.time, lights.next(sc), p.rng, g);
                            front = front || ray_enters_big_mesh(sc.primBoxes, bp, g.o, g.wi);
                        }
                    }

                    // ---- (3b) the BSDF step (render.cpp:322-363); the last iteration's sample is never used by the oracle's loop (:250)
                    if (bounce + 1 < maxDepth)
                    {
                        bool goesOn = bsdf_step(p, mat, h) == kContinue;
                        if (goesOn && rrStart > 0 && bounce + 1 >= rrStart)
                            goesOn = roulette_survives(p);
                        if (goesOn)
                        {
                            newFlags |= kStepHasExt;
                            front = front || ray_enters_big_mesh(sc.primBoxes, bp, p.o, p.d);
                        }
                    }
                }
            }

            // ---- the survivor (a next ray, or light samples still to resolve) to its new position; a finished path's radiance to its slot
            const bool store = newFlags != 0;
            const uint32_t np = out.push(store, front);
            if (store)
            {
                if (!(newFlags & kStepHasExt))
                    p.d = V3(0.0f);                     // (no extension ray: k_walk skips an all-zero direction)
                p.rayType |= newFlags;
                store_state(ss, nxt, np, p, slot);
                if (newFlags & kStepHasNee)
                {
                    float4* rayp = ss.pairRay[nxt] + np;
                    float4* pendp = ss.pairPend[nxt] + np;
                    const Mat mat = load_mat(sc.mats, prim);        // (read again rather than kept across the BSDF step)
                    // sample 0: its |dot(wi, n)| in the throughput record's spare word (one light sample per bounce -- every BASELINE scene but
                    // veach -- then needs 64 B of pending records, not 80)
                    float absDot0 = 0.0f;
                    rayp[0] = make_float4(g0.o.x, g0.o.y, g0.o.z, g0.dist);
                    rayp[cap] = make_float4(g0.wi.x, g0.wi.y, g0.wi.z, g0.nl);
                    if (g0.dist < 0.0f)
                    {
                        const V3 L = nee_contrib_probe(mat, h, g0.wi, sky0, skyPdf0);
                        pendp[0] = make_float4(L.x, L.y, L.z, 0.0f);
                    }
                    else
                    {
                        const NeeTerms e = nee_bsdf_terms(mat, h, g0.wi);
                        pendp[0] = make_float4(e.f.x, e.f.y, e.f.z, e.bsdfPdf);
                        absDot0 = e.absDot;
                    }
                    ss.pairThr[nxt][np] = make_float4(thrNee.x, thrNee.y, thrNee.z, absDot0);
                    // samples 1 .. K-1: the same draws again, from the stream as it stood behind sample 0
                    Rng replay = rngAfter0;
                    LightCursor lights = lightsAfter0;
                    for (int k = 1; k < K; ++k)
                    {
                        NeeGeo g;
                        const int light = lights.next(sc);
                        nee_sample_light(sc, h.p, h.n, p.time, light, replay, g);
                        const NeeTerms e = nee_bsdf_terms(mat, h, g.wi);
                        rayp[(size_t)(k*2)*cap] = make_float4(g.o.x, g.o.y, g.o.z, g.dist);
                        rayp[(size_t)(k*2 + 1)*cap] = make_float4(g.wi.x, g.wi.y, g.wi.z, g.nl);
                        pendp[(size_t)(k*2)*cap] = make_float4(e.f.x, e.f.y, e.f.z, e.bsdfPdf);
                        pendp[(size_t)(k*2 + 1)*cap] = make_float4(e.absDot, 0.0f, 0.0f, 0.0f);
                    }
                }
            }
            else if (valid)
                ss.radOut[slot] = make_float4(p.rad.x, p.rad.y, p.rad.z, __int_as_float(p.rayType));
        }
        if (lane == 0)
        {
            ss.segFront[(size_t)(bounce + 1)*ss.numRegions + r] = out.nFront;
            ss.segBack[(size_t)(bounce + 1)*ss.numRegions + r] = out.nBack;
        }
    }

    wave_add_stat(q.stats, 0, rays);
    wave_add_stat(q.stats, 5, shadowRays);
}

} // namespace tn
