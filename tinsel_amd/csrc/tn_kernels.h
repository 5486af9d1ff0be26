// tn_kernels.h -- the gfx950 kernels, by pipeline:
//
//   tn_path_state.h   what they share: parameter blocks, wave helpers, LDS staging of the scene arena, path slots, the camera sample, the
//                     dense path state (SplitState: regions packed at both ends by wave64 ballots -- no queue, no atomic)
//   tn_fused.h        FUSED wavefront pipeline (scene staged whole into LDS): k_bounce, one launch over all bounces of a batch
//   tn_split.h        SPLIT wavefront pipeline (meshes / a scene BVH in HBM): k_generate, k_extend, k_lights, k_shadow, k_shade(_sorted),
//                     k_region_order, k_seg_prefix / k_seg_expand
//   tn_paired.h       PAIRED wavefront pipeline (meshes in HBM): ONE k_walk and ONE streaming kernel (k_step) per bounce
//   tn_walk.h         k_walk / k_walk_rays: the mesh walk of meshes in HBM with ray replacement
//   tn_swalk.h        k_swalk: the scene-level walk with ray replacement (scenes beyond the flat scan)
//   tn_accumulate.h   AddSample as an order-preserving gather: k_accumulate, k_accumulate_tiled, k_accumulate_piped
//   here              k_mega (one lane per whole path: the A/B arm), k_pass_seeds, k_normals (eNormals), k_leaf (test hook)
//
// Every pipeline runs the same per-path arithmetic (tn_integrator.h, tn_bsdf.h, tn_isect.h) and leaves a finished path's radiance in
// PathState::rad[slot]; all are asserted bitwise equal to each other and to the CPU reference (tests/test_gpu_parity.py).
#pragma once

#include "tn_integrator.h"
#include "tn_display.h"
#include "tn_walk.h"

#include "tn_path_state.h"
#include "tn_fused.h"
#include "tn_split.h"
#include "tn_swalk.h"
#include "tn_paired.h"
#include "tn_accumulate.h"

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
