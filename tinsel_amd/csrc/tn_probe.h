// tn_probe.h -- lat-long environment probe: eval / pdf / importance sampling
// (reference src/probe.h:105-236) and Sky::Eval (src/scene.h:168-178).
#pragma once

#include "tn_scene.h"

namespace tn {

struct V2 { float x, y; };

// ProbeDirToUV (probe.h:105-113)
TN_D V2 probe_dir_to_uv(V3 dir)
{
    float theta = m_acosf(clampT(dir.y, -1.0f, 1.0f));
    float phi = (dir.x == 0.0f && dir.z == 0.0f) ? 0.0f : m_atan2f(dir.z, dir.x);
    float u = (kPi + phi)*kInvPi*0.5f;
    float v = theta*kInvPi;
    V2 r = { u, v };
    return r;
}

// ProbeUVToDir (probe.h:115-125)
TN_D V3 probe_uv_to_dir(V2 uv)
{
    float theta = uv.y*kPi;
    float phi = uv.x*2.0f*kPi;
    float st, ct, sp, cp;
    m_sincosf(theta, st, ct);
    m_sincosf(phi, sp, cp);
    float x = -st*cp;
    float y = ct;
    float z = -st*sp;
    return V3(x, y, z);
}

TN_D int clampI(int x, int lo, int hi) { return minI(maxI(x, lo), hi); }

// ProbeEval (probe.h:128-134)
TN_D V3 probe_eval(const DevProbe& p, V2 uv)
{
    int px = clampI(int(uv.x*p.width), 0, p.width - 1);
    int py = clampI(int(uv.y*p.height), 0, p.height - 1);
    float4 c = p.data[py*p.width + px];
    return V3(c.x, c.y, c.z);
}

// ProbePdf (probe.h:136-160)
TN_D float probe_pdf(const DevProbe& p, V3 d)
{
    V2 uv = probe_dir_to_uv(d);
    int col = clampI(int(uv.x*p.width), 0, p.width - 1);
    int row = clampI(int(uv.y*p.height), 0, p.height - 1);

    float pdf = p.pdfX[row*p.width + col]*p.pdfY[row];

    float sinTheta = m_sinf(uv.y*kPi);
    if (fabsf(sinTheta) < 0.0001f)
        pdf = 0.0f;
    else
        pdf *= float(p.width)*float(p.height)/(2.0f*kPi*kPi*sinTheta);
    return pdf;
}

// LowerBound(array, lower, upper, value) (probe.h:185-203)
TN_D int lower_bound(const float* array, int lower, int upper, float value)
{
    while (lower < upper)
    {
        int mid = lower + (upper - lower)/2;
        if (array[mid] < value)
            lower = mid + 1;
        else
            upper = mid;
    }
    return lower;
}

// ProbeSample (probe.h:205-236)
TN_D void probe_sample(const DevProbe& p, V3& dir, V3& color, float& pdf, Rng& rng)
{
    const uint32_t u1 = rng.rand();
    float r1 = (float)u1*(1.0f/4294967296.0f);      // Randf() of that draw
    float r2 = rng.randf();

    int row, col;
    if (p.alias)
    {
        // opt-in alias table (not sample-identical to the reference: same two draws, same distribution over the texels,
        // ONE dependent 8-B load instead of ~21 for the two binary searches over 800 rows and 1600 columns).  The bin comes
        // from the draw's 32 INTEGER bits: the fp32 uniform has 24, which over n = 1.28 M bins (loft.hdr) would give a bin 13 or
        // 14 of the representable values -- a selection probability up to 7 % off the pdf returned; with 32 bits a bin gets
        // 3355 or 3356 of the 2^32 values (3e-4).
        const int n = p.width*p.height;
        const int k = (int)(((unsigned long long)u1*(unsigned long long)n) >> 32);
        const uint2 e = p.alias[k];
        const int idx = (r2 < __uint_as_float(e.x)) ? k : (int)e.y;
        row = idx/p.width;
        col = idx - row*p.width;
    }
    else
    {
        row = lower_bound(p.cdfY, 0, p.height, r1);
        col = lower_bound(p.cdfX, row*p.width, (row + 1)*p.width, r2) - row*p.width;
    }

    float4 c = p.data[row*p.width + col];
    color = V3(c.x, c.y, c.z);
    pdf = p.pdfX[row*p.width + col]*p.pdfY[row];

    float u = col/float(p.width);
    float v = row/float(p.height);

    float sinTheta = m_sinf(v*kPi);
    if (sinTheta == 0.0f)
        pdf = 0.0f;
    else
        pdf *= (p.width*p.height)/(2.0f*kPi*kPi*sinTheta);

    V2 uv = { u, v };
    dir = probe_uv_to_dir(uv);
}

// Sky::Eval (scene.h:168-178)
TN_D V3 sky_eval(const DevScene& sc, V3 dir)
{
    if (sc.probe.valid)
        return probe_eval(sc.probe, probe_dir_to_uv(dir));
    V3 h(sc.horizon[0], sc.horizon[1], sc.horizon[2]);
    V3 z(sc.zenith[0], sc.zenith[1], sc.zenith[2]);
    return lerp3(h, z, sqrtf_cr(absf(dir.y)));
}

} // namespace tn
