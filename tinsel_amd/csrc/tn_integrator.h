// tn_integrator.h -- the path integrator of the CPU oracle (reference src/render.cpp:103-388)
// cut into the pieces the streaming pipeline needs.  Every piece is a pure function of the
// path registers, so the megakernel arm and the wavefront arm run the same arithmetic in the
// same order; only WHERE the registers live between pieces differs (VGPRs vs HBM queues).
//
//   PathTrace loop body (render.cpp:250-385):
//     on_hit_begin   :255-310   medium bookkeeping, Beer-Lambert, emission with BSDF-side MIS
//     nee_*          :103-227   SampleLights split into  sample (RNG draws + the shadow ray, before the
//                               trace)  and  contrib (BSDF terms + MIS, after it, for the rays that arrive)
//     bsdf_step      :322-363   light-hit termination, BSDFSample/BSDFEval, throughput, next ray
//     on_miss        :365-384   sky / probe with MIS
#pragma once

#include "tn_bsdf.h"
#include "tn_isect.h"
#include "tn_probe.h"

namespace tn {

constexpr float kRayEpsilon = 0.0001f;      // render.cpp:11
constexpr float kBsdfSamples = 1.0f;        // render.cpp:9
constexpr float kProbeSamples = 1.0f;       // render.cpp:10

struct PathRegs
{
    V3 o, d;            // rayOrigin, rayDir
    float time;         // rayTime
    V3 thr;             // pathThroughput
    V3 rad;             // totalRadiance
    Rng rng;
    float eta;          // rayEta
    V3 absorption;      // rayAbsorption
    float bsdfPdf;
    int rayType;        // BSDFType of the last bounce
    // Where rayAbsorption came from: it is never anything but 0 (outside) or the absorption of the material the path entered last
    // (render.cpp:259-269, 346-351) -- the primitive's index, or -1.  The wavefront pipelines keep THIS in their path state (4 B in a
    // record's spare word) instead of the 16-B vector, and read the vector back out of the material table.
    int medium;
};

// render.cpp:233-248
TN_D void path_begin(PathRegs& p, V3 o, V3 d, float time, Rng rng)
{
    p.o = o; p.d = d; p.time = time;
    p.thr = V3(1.0f, 1.0f, 1.0f);
    p.rad = V3(0.0f, 0.0f, 0.0f);
    p.rng = rng;
    p.eta = 1.0f;
    p.absorption = V3(0.0f);
    p.bsdfPdf = 1.0f;
    p.rayType = kReflected;
    p.medium = -1;
}

struct HitCtx
{
    V3 p;               // hit position
    V3 n;               // face-forwarded normal (surface == shading normal in the oracle, render.cpp:314)
    V3 wo;              // -rayDir
    float etaI, etaO;   // rayEta, outEta
    V3 outAbsorption;
    int outMedium;      // the primitive outAbsorption belongs to (entering: the hit one), or -1 (leaving: 0)
};

// render.cpp:255-310.  `bounce` is the loop index i.
TN_D void on_hit_begin(PathRegs& p, const Mat& mat, float t, V3 n, int bounce, HitCtx& h, int prim = -1)
{
    if (p.eta == 1.0f)
    {
        h.etaO = mat.ior;
        h.outAbsorption = mat.absorption;
        h.outMedium = prim;
    }
    else
    {
        h.etaO = 1.0f;
        h.outAbsorption = V3(0.0f);
        h.outMedium = -1;
    }
    h.etaI = p.eta;

    // pathThroughput *= Exp(-rayAbsorption*t)   (render.cpp:272, maths.h:253)
    // exp(-0*t) == 1 exactly and thr*1 == thr, so the (common) non-absorbing medium skips the three exps
    if (p.absorption.x != 0.0f || p.absorption.y != 0.0f || p.absorption.z != 0.0f)
    {
        V3 a = (-p.absorption)*t;
        p.thr = p.thr*V3(m_expf(a.x), m_expf(a.y), m_expf(a.z));
    }

    h.p = p.o + p.d*t;
    h.n = n;
    h.wo = -p.d;

    if (bounce == 0)
    {
        p.rad = p.rad + mat.emission;
    }
    else
    {
        float lightArea = mat.area;
        if (lightArea > 0.0f)
        {
            // (1.0f/lightArea, kBsdfSamples/N and float(lightSamples)/N with N = lightSamples + kBsdfSamples: the host's, Mat128)
            float lightPdf = (mat.rcpArea*t*t)/clampT(dot(-p.d, n), 1.e-3f, 1.0f);

            float cbsdf = mat.cbsdf;
            float clight = mat.clight;
            float weight = cbsdf*p.bsdfPdf/(cbsdf*p.bsdfPdf + clight*lightPdf);

            if (p.rayType == kSpecular)
                weight = 1.0f;

            p.rad = p.rad + weight*p.thr*mat.emission;
        }
    }
}

// One NEE shadow ray: what SampleLights knows BEFORE its Trace() call and without the BSDF -- the RNG draws and the
// geometry of the sample.  The BSDF terms of the sample (render.cpp:198-199, :127-128) are pure functions of the hit
// and of `wi`: they are evaluated AFTER the trace, only for the samples that reach their light (nee_contrib_*), which
// is the oracle's own order (render.cpp:171-219) and the same floats.
struct NeeGeo
{
    V3 o;               // shadow origin
    V3 wi;
    float dist;         // sqrtf_cr(dSq) for area lights; < 0 marks a probe sample
    float nl;           // |dot(lightNormal, wi)|
};

// PrimitiveSample (intersection.h:855-904) for light `prim`
// (Every active lane of a light loop samples the same light, so its record, the cursor's and nee_sum's reads could take the scalar path: built
// in round 6 with a light table behind the primitive records, and slower -- cornell 5776 -> 5680 cursor only, 5702 with the record in SGPRs,
// profiles/r06_2f_ab_light_table.md: k_bounce is short of SGPRs before it is short of LDS round trips.)
template <class SC>
TN_D void primitive_sample(const SC& sc, int index, float time, V3& pos, V3& normal, Rng& rng)
{
    const Prim64 p = load_prim(sc.prims, index);
    const Xform x = prim_pose(sc, p, time);

    if (p.type == kPrimSphere)
    {
        float u1 = rng.randf();
        float u2 = rng.randf();
        pos = pose_xform_point(p, x, uniform_sample_sphere(u1, u2)*p.g0);
        normal = normalize(pos - x.p);
    }
    else if (p.type == kPrimMesh)
    {
        const Tri48* mtris;
        const float* nr;
        int tri;
        float r = rng.randf();
#if TN_QUAD_RECORD
        if (p.flags & kPrimQuadArena)
        {
            // a quad in the arena (tn_isect.h): its arrays through the offsets in the record; LowerBound over two entries written down
            // (mid = 1 first: cdf[1] < r -> lo = 2 -> clamped to 1; else mid = 0: cdf[0] < r -> 1, else 0) -- both entries in one round trip
            const unsigned char* base = (SC::kLds || sc.arenaLdsBytes != 0u) ? sc.ldsBase : sc.arena;
            const QuadOffsets q = quad_offsets(__float_as_uint(p.g0), __float_as_uint(p.g1));
            const float* mcdf = reinterpret_cast<const float*>(base + q.cdf);
            const float c0 = mcdf[0], c1 = mcdf[1];
            tri = (c1 < r) ? 1 : (c0 < r) ? 1 : 0;
            mtris = reinterpret_cast<const Tri48*>(base + q.tris);
            nr = reinterpret_cast<const float*>(base + q.normals);
        }
        else
#endif
        {
        const DevMesh m = sc.meshes[p.mesh];
        const float* mcdf = mesh_cdf(sc, m);

        // LowerBound(cdf, cdf+numTris, r) (probe.h:162-183), clamped (intersection.h:880-881)
        int lo = 0, hi = m.numTris;
        while (lo < hi)
        {
            int mid = lo + (hi - lo)/2;
            if (mcdf[mid] < r)
                lo = mid + 1;
            else
                hi = mid;
        }
        tri = minI(lo, m.numTris - 1);
        mtris = mesh_tris(sc, m);
        nr = mesh_normals(sc, m);
        }

        float u, v;
        uniform_sample_triangle(rng, u, v);

        const float4* tp = reinterpret_cast<const float4*>(mtris + tri);
        float4 ta = tp[0], tb = tp[1], tc = tp[2];
        V3 a(ta.x, ta.y, ta.z), b(tb.x, tb.y, tb.z), c(tc.x, tc.y, tc.z);
        int i0 = __float_as_int(ta.w), i1 = __float_as_int(tb.w), i2 = __float_as_int(tc.w);
        V3 n1(nr[i0*3 + 0], nr[i0*3 + 1], nr[i0*3 + 2]);
        V3 n2(nr[i1*3 + 0], nr[i1*3 + 1], nr[i1*3 + 2]);
        V3 n3(nr[i2*3 + 0], nr[i2*3 + 1], nr[i2*3 + 2]);

        pos = pose_xform_point(p, x, u*a + v*b + (1.0f - u - v)*c);
        normal = safe_normalize(pose_xform_vector(p, x, u*n1 + v*n2 + (1.0f - u - v)*n3), V3(0.0f));
    }
    // planes are never lights (PrimitiveSample asserts, intersection.h:871-875)
}

// render.cpp:158-170: sample a point of `light`, the shadow ray towards it
template <class SC>
TN_D void nee_sample_light(const SC& sc, V3 hitP, V3 hitN, float time, int light, Rng& rng, NeeGeo& g)
{
    V3 lightPos, lightNormal;
    primitive_sample(sc, light, time, lightPos, lightNormal, rng);

    V3 wi = lightPos - hitP;
    float dSq = length_sq(wi);
    wi = divs(wi, sqrtf_cr(dSq));                  // wi /= sqrtf_cr(dSq)  (maths.h:251)

    g.o = hitP + face_forward(hitN, wi)*kRayEpsilon;
    g.wi = wi;
    g.dist = sqrtf_cr(dSq);
    g.nl = absf(dot(lightNormal, wi));
}

// render.cpp:175-196, the tests between Trace() and the BSDF: does the shadow ray (closest hit `hitPrim` at `t`) reach
// its light sample?
TN_D bool nee_light_reached(const NeeGeo& g, int hitPrim, float t)
{
    const float kTolerance = 1.e-2f;
    return hitPrim >= 0 && fabsf(t - g.dist) <= kTolerance && !(absf(g.nl) < 1.e-6f);
}

// render.cpp:196-219 for a sample that reached its light, in two halves.  The BSDF terms of the sample (render.cpp:198-199) are pure
// functions of the hit and of `wi` -- they do not depend on the shadow ray's fate -- and the rest (light pdf, MIS weight, the product) is a
// handful of operations on them, on the shadow hit's t and on the emission of the primitive it hit.  nee_contrib_light below evaluates both
// halves after the trace (only samples that arrive pay for the BSDF: the fused and the split pipelines); the paired pipeline (tn_paired.h)
// evaluates the first half where the sample is drawn and carries its five floats across the trace.  Same operations on the same operands in
// the same order either way: the same bits.
struct NeeTerms
{
    V3 f;               // BSDFEval toward wi (0 when bsdfPdf <= 0: never used then)
    float bsdfPdf;      // BSDFPdf toward wi
    float absDot;       // |dot(wi, n)|
};

TN_D NeeTerms nee_bsdf_terms(const Mat& surf, const HitCtx& h, V3 wi)
{
    NeeTerms e;
    e.f = V3(0.0f);
    e.absDot = 0.0f;
    e.bsdfPdf = bsdf_pdf(surf, h.etaI, h.etaO, h.n, h.wo, wi);
    if (e.bsdfPdf > 0.0f)
    {
        e.f = bsdf_eval(surf, h.etaI, h.etaO, h.n, h.wo, wi);
        e.absDot = absf(dot(wi, h.n));
    }
    return e;
}

TN_D V3 nee_combine_light(const DevScene& sc, const NeeTerms& e, float nl, int light, int hitPrim, float t)
{
    V3 L(0.0f);
    const Mat128* lm = sc.mats + light;
    float tSq = t*t;
    float lightPdf = (lm->rcpArea*tSq)/nl;          // ((1.0f/lightArea)*t*t)/nl, the reciprocal divided on the host (Mat128)
    if (e.bsdfPdf > 0.0f)
    {
        float cbsdf = lm->cbsdf;                    // kBsdfSamples/N, float(lightSamples)/N: the host's (Mat128)
        float clight = lm->clight;
        float weight = clight*lightPdf/(cbsdf*e.bsdfPdf + clight*lightPdf);
        const Mat128* hm = sc.mats + hitPrim;
        V3 em(hm->emission[0], hm->emission[1], hm->emission[2]);
        L = weight*e.f*em*(e.absDot/maxT(1.e-3f, lightPdf));
    }
    return L;
}

TN_D V3 nee_contrib_light(const DevScene& sc, const Mat& surf, const HitCtx& h, V3 wi, float nl, int light, int hitPrim, float t)
{
    return nee_combine_light(sc, nee_bsdf_terms(surf, h, wi), nl, light, hitPrim, t);
}

// render.cpp:107-116: the probe sample and its shadow ray
TN_D void nee_sample_probe(const DevScene& sc, V3 hitP, V3 hitN, Rng& rng, NeeGeo& g, V3& skyColor, float& skyPdf)
{
    V3 wi;
    probe_sample(sc.probe, wi, skyColor, skyPdf, rng);

    g.o = hitP + face_forward(hitN, wi)*kRayEpsilon;
    g.wi = wi;
    g.dist = -1.0f;
    g.nl = 0.0f;
}

// render.cpp:118-139 for an unoccluded probe sample
TN_D V3 nee_contrib_probe(const Mat& surf, const HitCtx& h, V3 wi, V3 skyColor, float skyPdf)
{
    V3 L(0.0f);
    float bsdfPdf = bsdf_pdf(surf, h.etaI, h.etaO, h.n, h.wo, wi);
    if (bsdfPdf > 0.0f)
    {
        V3 f = bsdf_eval(surf, h.etaI, h.etaO, h.n, h.wo, wi);
        int N = int(kProbeSamples + kBsdfSamples);
        float cbsdf = kBsdfSamples/N;
        float csky = float(kProbeSamples)/N;
        float weight = csky*skyPdf/(cbsdf*bsdfPdf + csky*skyPdf);
        if (weight > 0.0f)
            L = divs(weight*skyColor*f*absf(dot(wi, h.n)), skyPdf);
    }
    return L;
}

// One fetch per light in the light loops (LightRec, tn_scene.h) -- the parity arm only: cornell +0.3-0.8 % there, while the tolerance arm's k_bounce
// loses its register allocation to it (56 VGPRs spilled: cornell 7147 -> 5812 Msamples/s, profiles/r06_2t_ab_fast_arm.md)
#ifndef TN_LIGHT_RECS
#ifdef TN_FAST
#define TN_LIGHT_RECS 0
#else
#define TN_LIGHT_RECS 1
#endif
#endif
// Which light does NEE ray k belong to?  Rays arrive in order (probe first, then lights x samples): the cursor walks along.
struct LightCursor
{
    int li = 0, sInLight = 0;
    TN_D int next(const DevScene& sc)
    {
#if TN_LIGHT_RECS
        for (;;)
        {
            const int4 e = *reinterpret_cast<const int4*>(sc.lights + li);      // {primitive, lightSamples, ..}: one fetch
            if (sInLight < e.y)
            {
                ++sInLight;
                return e.x;
            }
            ++li;
            sInLight = 0;
        }
#else
        while (sInLight >= sc.mats[sc.lights[li].prim].lightSamples) { ++li; sInLight = 0; }
        ++sInLight;
        return sc.lights[li].prim;
#endif
    }
};

// Sums per-light contributions in the oracle's order.  `contrib(k)` returns the resolved
// contribution of NEE ray k (k counts the probe ray first, then lights x samples).
template <class Contrib>
TN_D V3 nee_sum(const DevScene& sc, Contrib contrib)
{
    V3 sum(0.0f);
    int k = 0;
    if (sc.probe.valid)
    {
        sum = sum + contrib(k++);
        sum = divs(sum, float(kProbeSamples));              // render.cpp:142-143
    }
    for (int li = 0; li < sc.numLights; ++li)
    {
#if TN_LIGHT_RECS
        const int4 e = *reinterpret_cast<const int4*>(sc.lights + li);
        const int numSamples = e.y;
        const float rcpSamples = __int_as_float(e.z);
#else
        const int numSamples = sc.mats[sc.lights[li].prim].lightSamples;
#endif
        V3 L(0.0f);
        for (int s = 0; s < numSamples; ++s)
            L = L + contrib(k++);
#if TN_LIGHT_RECS
        sum = sum + L*rcpSamples;                                       // L*(1.0f/numSamples), render.cpp:223: divided on the host
#else
        sum = sum + L*sc.mats[sc.lights[li].prim].rcpLightSamples;
#endif
    }
    return sum;
}

enum StepResult : int { kContinue = 0, kTerminate = 1 };

// render.cpp:322-363
TN_D int bsdf_step(PathRegs& p, const Mat& mat, const HitCtx& h)
{
    if (mat.lightSamples)
        return kTerminate;

    V3 u, v;
    basis_from_vector(h.n, u, v);

    V3 bsdfDir;
    int bsdfType = kReflected;
    float pdf = 0.0f;
    bsdf_sample(mat, h.etaI, h.etaO, u, v, h.n, h.wo, bsdfDir, pdf, bsdfType, p.rng);
    p.bsdfPdf = pdf;

    if (pdf <= 0.0f)
        return kTerminate;

    V3 f = bsdf_eval(mat, h.etaI, h.etaO, h.n, h.wo, bsdfDir);

    if (dot(bsdfDir, h.n) <= 0.0f)
    {
        p.eta = h.etaO;
        p.absorption = h.outAbsorption;
        p.medium = h.outMedium;
    }

    // pathThroughput *= f * Abs(Dot(n, bsdfDir))/bsdfPdf   (Vec3/Real == a*(1.0/s), maths.h:242)
    p.thr = p.thr*divs(f*absf(dot(h.n, bsdfDir)), pdf);

    p.rayType = bsdfType;
    p.d = bsdfDir;
    p.o = h.p + face_forward(h.n, bsdfDir)*kRayEpsilon;
    return kContinue;
}

// Russian roulette (NOT in the reference: render.cpp:250 runs every path to maxDepth; opt-in through
// tinsel_hip_set_russian_roulette, restated identically in oracle/tinsel_oracle.c).  Called after a bounce whose
// successor will be traced: the path survives with probability q = min(1, max(throughput)) and is compensated by 1/q,
// so the estimate stays unbiased.  One extra draw from the path's stream when q < 1.
TN_D bool roulette_survives(PathRegs& p)
{
    const float q = minT(1.0f, maxT(p.thr.x, maxT(p.thr.y, p.thr.z)));
    if (!(q > 0.0f))
        return false;
    if (q < 1.0f)
    {
        const float u = p.rng.randf();
        if (u >= q)
            return false;
        p.thr = p.thr*rcpf_cr(q);
    }
    return true;
}

// render.cpp:365-383
TN_D void on_miss(const DevScene& sc, PathRegs& p, int bounce)
{
    float weight = 1.0f;
    if (sc.probe.valid && bounce > 0 && p.rayType != kSpecular)
    {
        float skyPdf = probe_pdf(sc.probe, p.d);
        int N = int(kProbeSamples + kBsdfSamples);
        float cbsdf = kBsdfSamples/N;
        float csky = float(kProbeSamples)/N;
        weight = cbsdf*p.bsdfPdf/(cbsdf*p.bsdfPdf + csky*skyPdf);
    }
    p.rad = p.rad + weight*sky_eval(sc, p.d)*p.thr;
}

} // namespace tn
