/*
 * tinsel_oracle.c -- TEST INFRASTRUCTURE ONLY.  A plain-C99 restatement of the reference's CPU
 * path-tracing path (mmacklin/tinsel: src/render.cpp PathTrace and everything below it), used as
 * the "port" checker of the HIP path.  Never linked into, imported by or executed from the product.
 *
 * Parity of THIS file is pinned: tests/test_oracle.py requires its per-path radiance to be
 * BIT-IDENTICAL to the reference's own PathTrace (oracle/_ref, the unmodified reference sources)
 * on every committed fixture (tests/golden/*.golden.npz were produced by the reference itself).
 * That only holds because every expression below keeps the reference's operation order and its
 * silent float->double promotions; each function cites the reference lines it restates.
 *
 * Build: gcc -std=c99 -O2 -ffp-contract=off -fno-fast-math -fPIC -shared -pthread (oracle/Makefile).
 * Unlike the reference it also COUNTS what it does (rays, node visits, triangle / primitive
 * tests): these feed the algorithmic-bytes model B_ray of DESIGN.md / SURVEY.md 8(d).
 */
#include "../include/tinsel_hip.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct { float x, y, z; } vec3;
typedef struct { float x, y, z, w; } quat;
typedef struct { vec3 p; quat r; float s; } xform;

#define kPi (3.141592653589793f)                    /* maths.h:32 */
#define k2Pi (3.141592653589793f*2.0f)              /* maths.h:33 */
#define kInvPi (1.0f/kPi)                           /* maths.h:34 */
#define kInv2Pi (1.0f/k2Pi)                         /* maths.h:35 */
#define kRayEpsilon 0.0001f                         /* render.cpp:11 */
#define kBsdfSamples 1.0f                           /* render.cpp:9 */
#define kProbeSamples 1.0f                          /* render.cpp:10 */

typedef struct {
    uint64_t rays, samples, internal, tris, prims, shadow, fetches, pad;
} counters;

/* ------------------------------------------------------------------------- maths.h */

static vec3 v3(float x, float y, float z) { vec3 r = { x, y, z }; return r; }
static vec3 v3s(float s) { vec3 r = { s, s, s }; return r; }
static vec3 vneg(vec3 a) { return v3(-a.x, -a.y, -a.z); }                                  /* :236 */
static vec3 vadd(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }           /* :237 */
static vec3 vsub(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }           /* :238 */
static vec3 vscale(vec3 a, float s) { return v3(a.x*s, a.y*s, a.z*s); }                    /* :239-240 */
static vec3 vmul(vec3 a, vec3 b) { return v3(a.x*b.x, a.y*b.y, a.z*b.z); }                 /* :241 */
static vec3 vdivs(vec3 a, float s) { return vscale(a, (float)(1.0/(double)s)); }           /* :242,251: a*(1.0/s) */
static float vdot(vec3 a, vec3 b) { return a.x*b.x + a.y*b.y + a.z*b.z; }                  /* :257 */
static vec3 vcross(vec3 a, vec3 b) { return v3(a.y*b.z - b.y*a.z, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x); } /* :256 */
static float vlength(vec3 a) { return sqrtf(vdot(a, a)); }                                  /* :259 */
static vec3 vnormalize(vec3 a) { return vdivs(a, vlength(a)); }                             /* :260 */

static vec3 vsafe_normalize(vec3 a, vec3 fallback)                                          /* :261-273 */
{
    float m = vdot(a, a);
    if (m > 0.0)
        return vscale(a, (float)(1.0/(double)sqrtf(m)));
    return fallback;
}

static float absT(float x) { if (x < 0.0) return -x; else return x; }                       /* :67-74 */
static float minT(float a, float b) { return (a < b) ? a : b; }                             /* :55-56 */
static float maxT(float a, float b) { return (a < b) ? b : a; }                             /* :58-59 */
static int minI(int a, int b) { return (a < b) ? a : b; }
static int maxI(int a, int b) { return (a < b) ? b : a; }
static float clampT(float x, float lo, float hi) { return minT(maxT(x, lo), hi); }          /* :61-65 */
static int clampI(int x, int lo, int hi) { return minI(maxI(x, lo), hi); }
static float lerpf(float a, float b, float t) { return a + (b - a)*t; }                     /* :76-80 */
static vec3 vlerp(vec3 a, vec3 b, float t) { return vadd(a, vscale(vsub(b, a), t)); }
static float sqrf(float x) { return x*x; }                                                  /* :40 */
static vec3 face_forward(vec3 n, vec3 v) { if (vdot(v, n) < 0.0f) return vneg(n); else return n; }   /* :1592-1598 */

static quat qmul(quat a, quat b)                                                            /* :531-537 */
{
    quat r;
    r.x = a.w*b.x + b.w*a.x + a.y*b.z - b.y*a.z;
    r.y = a.w*b.y + b.w*a.y + a.z*b.x - b.z*a.x;
    r.z = a.w*b.z + b.w*a.z + a.x*b.y - b.x*a.y;
    r.w = a.w*b.w - a.x*b.x - a.y*b.y - a.z*b.z;
    return r;
}
static quat qconj(quat q) { quat r = { -q.x, -q.y, -q.z, q.w }; return r; }                 /* :555 */
static vec3 qrotate(quat q, vec3 v)                                                         /* :558-563 */
{
    quat qv = { v.x, v.y, v.z, 0.0f };
    quat t = qmul(qmul(q, qv), qconj(q));
    return v3(t.x, t.y, t.z);
}
static quat qnormalize(quat q)                                                              /* :547-553 */
{
    float length = sqrtf(q.x*q.x + q.y*q.y + q.z*q.z + q.w*q.w);
    float rcp = (float)(1.0/(double)length);
    quat r = { q.x*rcp, q.y*rcp, q.z*rcp, q.w*rcp };
    return r;
}

static xform to_xform(const tinsel_transform* t)
{
    xform x;
    x.p = v3(t->p.x, t->p.y, t->p.z);
    x.r.x = t->r.x; x.r.y = t->r.y; x.r.z = t->r.z; x.r.w = t->r.w;
    x.s = t->s;
    return x;
}

static xform interpolate_xform(xform a, xform b, float t)                                   /* :1566-1569 */
{
    xform o;
    o.p = vlerp(a.p, b.p, t);
    quat q = { a.r.x + (b.r.x - a.r.x)*t, a.r.y + (b.r.y - a.r.y)*t, a.r.z + (b.r.z - a.r.z)*t, a.r.w + (b.r.w - a.r.w)*t };
    o.r = qnormalize(q);
    o.s = lerpf(a.s, b.s, t);
    return o;
}
static vec3 xform_vector(xform t, vec3 v) { return qrotate(t.r, vscale(v, t.s)); }                          /* :601-604 */
static vec3 xform_point(xform t, vec3 v) { return vadd(t.p, qrotate(t.r, vscale(v, t.s))); }                /* :606-609 */
static vec3 inv_xform_vector(xform t, vec3 v) { return vscale(qrotate(qconj(t.r), v), 1.0f/t.s); }          /* :611-614 */
static vec3 inv_xform_point(xform t, vec3 v) { return vscale(qrotate(qconj(t.r), vsub(v, t.p)), 1.0f/t.s); } /* :616-619 */

/* Random (maths.h:1036-1091) */
typedef struct { uint32_t seed1, seed2; } rng_t;
static rng_t rng_seeded(uint32_t seed) { rng_t r; r.seed1 = 315645664u + seed; r.seed2 = r.seed1 ^ 0x13ab45feu; return r; }
static uint32_t rng_rand(rng_t* r)
{
    r->seed1 = (r->seed2 ^ ((r->seed1 << 5) | (r->seed1 >> 27))) ^ (r->seed1*r->seed2);
    r->seed2 = r->seed1 ^ ((r->seed2 << 12) | (r->seed2 >> 20));
    return r->seed1;
}
static float rng_randf(rng_t* r)
{
    unsigned int value = rng_rand(r);
    unsigned int limit = 0xffffffff;
    return (float)value*(1.0f/(float)limit);
}
/* Randf(min,max) (maths.h:1080-1084); Sample1D/Sample2D call it with (0,1) (sampler.h:238-289) */
static float rng_randf01(rng_t* r) { float t = rng_randf(r); return (1.0f - t)*0.0f + t*1.0f; }

static uint32_t pass_seed(uint32_t passIndex)      /* render.cu:1050-1052 `seed = Random(frame)`, :1099 `seed.Rand()` */
{
    rng_t r = rng_seeded(1u);
    uint32_t v = 0;
    for (uint32_t i = 0; i <= passIndex; ++i)
        v = rng_rand(&r);
    return v;
}

static void basis_from_vector(vec3 w, vec3* u, vec3* v)                                     /* :1261-1275 */
{
    if (fabsf(w.x) > fabsf(w.y))
    {
        float invLen = (float)(1.0/(double)sqrtf(w.x*w.x + w.z*w.z));
        *u = v3(-w.z*invLen, 0.0f, w.x*invLen);
    }
    else
    {
        float invLen = (float)(1.0/(double)sqrtf(w.y*w.y + w.z*w.z));
        *u = v3(0.0f, w.z*invLen, -w.y*invLen);
    }
    *v = vcross(w, *u);
}

static vec3 uniform_sample_sphere(float u1, float u2)                                       /* :1278-1287 */
{
    float z = 1.f - 2.f*u1;
    float r = sqrtf(maxT(0.f, 1.f - z*z));
    float phi = 2.f*kPi*u2;
    float x = r*cosf(phi);
    float y = r*sinf(phi);
    return v3(x, y, z);
}

static vec3 uniform_sample_hemisphere(rng_t* rand)                                          /* :1291-1302 */
{
    float z = rng_randf01(rand);
    float w = sqrtf(1.0f - z*z);
    float phi = k2Pi*rng_randf01(rand);
    float x = cosf(phi)*w;
    float y = sinf(phi)*w;
    return v3(x, y, z);
}

static vec3 cosine_sample_hemisphere(float u1, float u2)                                    /* :1304-1310, 1319-1325 */
{
    float r = sqrtf(u1);
    float theta = k2Pi*u2;
    float sx = r*cosf(theta), sy = r*sinf(theta);
    float z = sqrtf(maxT(0.0f, 1.0f - sx*sx - sy*sy));
    return v3(sx, sy, z);
}

static void uniform_sample_triangle(rng_t* rand, float* u, float* v)                        /* :1312-1317 */
{
    float r = sqrtf(rng_randf(rand));
    *u = 1.0f - r;
    *v = rng_randf(rand)*r;
}

static vec3 clamp_length(vec3 v, float maxLength)                                           /* :1577-1589 */
{
    float l = vlength(v);
    if (l > maxLength)
        return vscale(v, maxLength/l);
    return v;
}

/* ------------------------------------------------------------------------- scene */

typedef struct {
    unsigned char* blob;
    const tinsel_primitive* prims;
    int numPrims;
    const tinsel_bvh_node* nodes;
    int numNodes;
    vec3 horizon, zenith;
    int probeValid, probeW, probeH;
    const tinsel_vec4* probeData;
    const float *pdfX, *cdfX, *pdfY, *cdfY;
    tinsel_camera camera;
    tinsel_options options;
} scene_t;

static int node_leaf(const tinsel_bvh_node* n) { return (int)(n->right_index_leaf >> 31); }
static uint32_t node_right(const tinsel_bvh_node* n) { return n->right_index_leaf & 0x7fffffffu; }

/* ------------------------------------------------------------------------- intersection.h */

typedef struct { vec3 origin, dir; float time; } ray_t;

static float minf_ref(float a, float b) { return a < b ? a : b; }                           /* :369 */
static float maxf_ref(float a, float b) { return a > b ? a : b; }                           /* :370 */

static int ray_aabb(vec3 pos, vec3 rcp, tinsel_vec3 mn, tinsel_vec3 mx, float* t)          /* IntersectRayAABBFast :373-397 */
{
    float l1 = (mn.x - pos.x)*rcp.x, l2 = (mx.x - pos.x)*rcp.x;
    float lmin = minf_ref(l1, l2), lmax = maxf_ref(l1, l2);
    l1 = (mn.y - pos.y)*rcp.y; l2 = (mx.y - pos.y)*rcp.y;
    lmin = maxf_ref(minf_ref(l1, l2), lmin); lmax = minf_ref(maxf_ref(l1, l2), lmax);
    l1 = (mn.z - pos.z)*rcp.z; l2 = (mx.z - pos.z)*rcp.z;
    lmin = maxf_ref(minf_ref(l1, l2), lmin); lmax = minf_ref(maxf_ref(l1, l2), lmax);
    int hit = ((lmax >= 0.f) & (lmax >= lmin));
    if (hit)
        *t = lmin;
    return hit;
}

static int ray_sphere(vec3 center, float radius, vec3 o, vec3 d, float* outT, vec3* outN)   /* :30-83 */
{
    vec3 q = vsub(o, center);
    float a = 1.0f;
    float b = 2.0f*vdot(q, d);
    float c = vdot(q, q) - (radius*radius);
    float minTv, maxTv;
    /* SolveQuadratic; a == 1 so the degenerate branch (:32) is dead */
    float disc = b*b - 4.0f*a*c;
    if (disc < 0.0f)
        return 0;
    float t = -0.5f*(b + ((b < 0.0f) ? -1.0f : 1.0f)*sqrtf(disc));
    minTv = t/a;
    maxTv = c/t;
    if (maxTv < minTv) { float tmp = minTv; minTv = maxTv; maxTv = tmp; }
    if (minTv < 0.0f && maxTv < 0.0f)
        return 0;
    if (minTv < 0.0f && maxTv > 0.0f)
        minTv = maxTv;
    *outN = vnormalize(vsub(vadd(o, vscale(d, minTv)), center));
    *outT = minTv;
    return 1;
}

static int ray_plane(vec3 p, vec3 dir, const float* pl, float* t)                           /* :85-99 */
{
    float d = pl[0]*dir.x + pl[1]*dir.y + pl[2]*dir.z + pl[3]*0.0f;
    if (d == 0.0f)
        return 0;
    *t = -(pl[0]*p.x + pl[1]*p.y + pl[2]*p.z + pl[3]*1.0f)/d;
    return *t > 0.0f;
}

static int ray_tri(vec3 p, vec3 dir, vec3 a, vec3 b, vec3 c, float* t, float* u, float* v, float* w, float* sign, vec3* n) /* :117-145 */
{
    vec3 ab = vsub(b, a), ac = vsub(c, a);
    *n = vcross(ab, ac);
    vec3 nd = vneg(dir);
    float d = vdot(nd, *n);
    float ood = 1.0f/d;
    vec3 ap = vsub(p, a);
    *t = vdot(ap, *n)*ood;
    if (*t < 0.0f)
        return 0;
    vec3 e = vcross(nd, ap);
    *v = vdot(ac, e)*ood;
    if (*v < 0.0f || *v > 1.0f)
        return 0;
    *w = -vdot(ab, e)*ood;
    if (*w < 0.0f || *v + *w > 1.0f)
        return 0;
    *u = 1.0f - *v - *w;
    *sign = d;
    return 1;
}

static vec3 tv(const tinsel_vec3* a, int i) { return v3(a[i].x, a[i].y, a[i].z); }

/* IntersectRayMesh + MeshQuery (:629-749) on the reference's own 32-B nodes */
static int ray_mesh(const tinsel_mesh_geometry* m, vec3 origin, vec3 dir, float* t, float* u, float* v, float* w, int* tri, vec3* triN, counters* ct)
{
    float closestT = FLT_MAX, cu = 0, cv = 0, cw = 0;
    vec3 cn = v3s(0.0f);
    int ctri = 0;
    float tmax = FLT_MAX;
    vec3 rcp = v3(1.0f/dir.x, 1.0f/dir.y, 1.0f/dir.z);
    int stack[64];
    int count = 1;
    stack[0] = 0;
    while (count)
    {
        tinsel_bvh_node node = m->nodes[stack[--count]];
        ct->fetches++;
        if (node_leaf(&node))
        {
            int i = (int)node.left_index;
            float tt, uu, vv, ww, sign;
            vec3 n;
            ct->tris++;
            if (ray_tri(origin, dir, tv(m->positions, m->indices[i*3 + 0]), tv(m->positions, m->indices[i*3 + 1]),
                        tv(m->positions, m->indices[i*3 + 2]), &tt, &uu, &vv, &ww, &sign, &n))
            {
                if (tt > 0.0f && tt < closestT)
                {
                    closestT = tt; cu = uu; cv = vv; cw = ww; ctri = i;
                    cn = vscale(n, sign);
                }
            }
            tmax = closestT;
        }
        else
        {
            const tinsel_bvh_node* left = &m->nodes[node.left_index];
            const tinsel_bvh_node* right = &m->nodes[node_right(&node)];
            uint32_t li = node.left_index, ri = node_right(&node);
            float tLeft, tRight;
            ct->fetches += 2;
            ct->internal++;
            int hitLeft = ray_aabb(origin, rcp, left->lower, left->upper, &tLeft) && tLeft < tmax;
            int hitRight = ray_aabb(origin, rcp, right->lower, right->upper, &tRight) && tRight < tmax;
            if (hitLeft && hitRight && (tLeft < tRight)) { uint32_t tmp = li; li = ri; ri = tmp; }
            if (hitLeft) stack[count++] = (int)li;
            if (hitRight) stack[count++] = (int)ri;
        }
    }
    if (closestT < FLT_MAX)
    {
        *t = closestT; *u = cu; *v = cv; *w = cw; *tri = ctri; *triN = cn;
        return 1;
    }
    return 0;
}

static int primitive_intersect(const tinsel_primitive* p, const ray_t* ray, float* outT, vec3* outN, counters* ct)  /* :951-1020 */
{
    xform x = interpolate_xform(to_xform(&p->start_transform), to_xform(&p->end_transform), ray->time);
    ct->prims++;
    switch (p->type)
    {
    case TINSEL_GEOM_SPHERE:
        return ray_sphere(x.p, p->geo.sphere.radius*x.s, ray->origin, ray->dir, outT, outN);
    case TINSEL_GEOM_PLANE:
    {
        int hit = ray_plane(ray->origin, ray->dir, p->geo.plane.plane, outT);
        if (hit)
            *outN = v3(p->geo.plane.plane[0], p->geo.plane.plane[1], p->geo.plane.plane[2]);
        return hit;
    }
    case TINSEL_GEOM_MESH:
    {
        const tinsel_mesh_geometry* m = &p->geo.mesh;
        vec3 lo = inv_xform_point(x, ray->origin);
        vec3 ld = inv_xform_vector(x, ray->dir);
        float t, u, v, w;
        int tri;
        vec3 triN;
        if (!ray_mesh(m, lo, ld, &t, &u, &v, &w, &tri, &triN, ct))
            return 0;
        vec3 n1 = tv(m->normals, m->indices[tri*3 + 0]);
        vec3 n2 = tv(m->normals, m->indices[tri*3 + 1]);
        vec3 n3 = tv(m->normals, m->indices[tri*3 + 2]);
        vec3 smooth = vadd(vadd(vscale(n1, u), vscale(n2, v)), vscale(n3, w));
        if (vdot(smooth, triN) < 0.0f)
            smooth = vscale(smooth, -1.0f);
        *outT = t;
        *outN = vsafe_normalize(xform_vector(x, smooth), triN);
        return 1;
    }
    }
    return 0;
}

/* Trace (render.cpp:17-62) over QueryBVH (intersection.h:751-799) */
static const tinsel_primitive* trace(const scene_t* sc, const ray_t* ray, float* outT, vec3* outN, counters* ct)
{
    float minTv = FLT_MAX;
    vec3 cn = v3s(0.0f);
    const tinsel_primitive* closest = NULL;
    vec3 rcp = v3(1.0f/ray->dir.x, 1.0f/ray->dir.y, 1.0f/ray->dir.z);
    int stack[64];
    int count = 1;
    stack[0] = 0;
    ct->rays++;
    while (count)
    {
        tinsel_bvh_node node = sc->nodes[stack[--count]];
        ct->fetches++;
        if (node_leaf(&node))
        {
            float t;
            vec3 n;
            const tinsel_primitive* prim = &sc->prims[node.left_index];
            if (primitive_intersect(prim, ray, &t, &n, ct))
            {
                if (t < minTv && t > 0.0f)
                {
                    minTv = t;
                    closest = prim;
                    cn = n;
                }
            }
        }
        else
        {
            const tinsel_bvh_node* left = &sc->nodes[node.left_index];
            const tinsel_bvh_node* right = &sc->nodes[node_right(&node)];
            uint32_t li = node.left_index, ri = node_right(&node);
            float tLeft, tRight;
            ct->fetches += 2;
            ct->internal++;
            int hitLeft = ray_aabb(ray->origin, rcp, left->lower, left->upper, &tLeft);
            int hitRight = ray_aabb(ray->origin, rcp, right->lower, right->upper, &tRight);
            if (hitLeft && hitRight && (tLeft < tRight)) { uint32_t tmp = li; li = ri; ri = tmp; }
            if (hitLeft) stack[count++] = (int)li;
            if (hitRight) stack[count++] = (int)ri;
        }
    }
    *outT = minTv;
    *outN = face_forward(cn, vneg(ray->dir));
    return closest;
}

static float primitive_area(const tinsel_primitive* p)                                      /* :833-853 */
{
    switch (p->type)
    {
    case TINSEL_GEOM_SPHERE: return 4.0f*kPi*p->geo.sphere.radius*p->geo.sphere.radius;
    case TINSEL_GEOM_PLANE: return 0.0f;
    case TINSEL_GEOM_MESH: return p->geo.mesh.area*p->end_transform.s;
    }
    return 0.0f;
}

static void primitive_sample(const tinsel_primitive* p, float time, vec3* pos, vec3* normal, rng_t* rand)   /* :855-904 */
{
    xform x = interpolate_xform(to_xform(&p->start_transform), to_xform(&p->end_transform), time);
    if (p->type == TINSEL_GEOM_SPHERE)
    {
        float u1 = rng_randf01(rand), u2 = rng_randf01(rand);
        *pos = xform_point(x, vscale(uniform_sample_sphere(u1, u2), p->geo.sphere.radius));
        *normal = vnormalize(vsub(*pos, x.p));
    }
    else if (p->type == TINSEL_GEOM_MESH)
    {
        const tinsel_mesh_geometry* m = &p->geo.mesh;
        float r = rng_randf(rand);
        int numTris = m->num_indices/3;
        int lo = 0, hi = numTris;                   /* LowerBound (probe.h:162-183) */
        while (lo < hi)
        {
            int mid = lo + (hi - lo)/2;
            if (m->cdf[mid] < r) lo = mid + 1; else hi = mid;
        }
        int tri = minI(lo, numTris - 1);
        float u, v;
        uniform_sample_triangle(rand, &u, &v);
        int i0 = m->indices[tri*3 + 0], i1 = m->indices[tri*3 + 1], i2 = m->indices[tri*3 + 2];
        vec3 a = tv(m->positions, i0), b = tv(m->positions, i1), c = tv(m->positions, i2);
        vec3 n1 = tv(m->normals, i0), n2 = tv(m->normals, i1), n3 = tv(m->normals, i2);
        *pos = xform_point(x, vadd(vadd(vscale(a, u), vscale(b, v)), vscale(c, 1.0f - u - v)));
        *normal = vsafe_normalize(xform_vector(x, vadd(vadd(vscale(n1, u), vscale(n2, v)), vscale(n3, 1.0f - u - v))), v3s(0.0f));
    }
}

/* ------------------------------------------------------------------------- disney.h */

enum { eReflected = 0, eTransmitted = 1, eSpecular = 2 };

static float mat_ior(const tinsel_material* m)                                              /* scene.h:72-78 */
{
    if (m->eta == 0.0f)
        return 2.0f/(1.0f - sqrtf((float)(0.08*(double)m->specular))) - 1.0f;
    return m->eta;
}

static int refract(vec3 wi, vec3 n, float eta, vec3* wt)                                    /* :34-47 */
{
    float cosThetaI = vdot(n, wi);
    float sin2ThetaI = maxT(0.0f, (float)(1.0f - cosThetaI*cosThetaI));
    float sin2ThetaT = eta*eta*sin2ThetaI;
    if (sin2ThetaT >= 1)
        return 0;
    float cosThetaT = sqrtf(1.0f - sin2ThetaT);
    *wt = vadd(vscale(vneg(wi), eta), vscale(n, eta*cosThetaI - cosThetaT));
    return 1;
}

static float schlick_fresnel(float u) { float m = clampT(1 - u, 0.0f, 1.0f); float m2 = m*m; return m2*m2*m; }     /* :49-54 */
static float gtr1(float NDotH, float a)                                                     /* :56-62 */
{
    if (a >= 1) return kInvPi;
    float a2 = a*a;
    float t = 1 + (a2 - 1)*NDotH*NDotH;
    return (a2 - 1)/(kPi*logf(a2)*t);
}
static float gtr2(float NDotH, float a) { float a2 = a*a; float t = 1.0f + (a2 - 1.0f)*NDotH*NDotH; return a2/(kPi*t*t); }  /* :64-69 */
static float smith_ggx(float NDotv, float alphaG) { float a = alphaG*alphaG; float b = NDotv*NDotv; return 1/(NDotv + sqrtf(a + b - a*b)); }  /* :71-76 */

static float fresnel(float VDotN, float etaI, float etaT)                                   /* Fr :79-96 */
{
    float SinThetaT2 = sqrf(etaI/etaT)*(1.0f - VDotN*VDotN);
    if (SinThetaT2 > 1.0f)
        return 1.0f;
    float LDotN = sqrtf(1.0f - SinThetaT2);
    float eta = etaT/etaI;
    float r1 = (VDotN - eta*LDotN)/(VDotN + eta*LDotN);
    float r2 = (LDotN - eta*VDotN)/(LDotN + eta*VDotN);
    return 0.5f*(sqrf(r1) + sqrf(r2));
}

static float bsdf_pdf(const tinsel_material* mat, float etaI, float etaO, vec3 n, vec3 V, vec3 L)   /* :125-166 */
{
    if (vdot(L, n) <= 0.0f)
    {
        float bsdfPdf = 0.0f;
        float brdfPdf = kInv2Pi*mat->subsurface*0.5f;
        return lerpf(brdfPdf, bsdfPdf, mat->transmission);
    }
    else
    {
        float F = fresnel(vdot(n, V), etaI, etaO);
        const float a = maxT(0.001f, mat->roughness);
        const vec3 half = vsafe_normalize(vadd(L, V), v3s(0.0f));
        const float cosThetaHalf = absT(vdot(half, n));
        const float pdfHalf = gtr2(cosThetaHalf, a)*cosThetaHalf;
        float pdfSpec = 0.25f*pdfHalf/maxT(1.e-6f, vdot(L, half));
        float pdfDiff = absT(vdot(L, n))*kInvPi*(1.0f - mat->subsurface);
        float bsdfPdf = pdfSpec*F;
        float brdfPdf = lerpf(pdfDiff, pdfSpec, 0.5f);
        return lerpf(brdfPdf, bsdfPdf, mat->transmission);
    }
}

static vec3 sample_ggx(const tinsel_material* mat, vec3 U, vec3 Vt, vec3 N, vec3 view, float r1, float r2)  /* :184-204, 265-285 */
{
    const float a = maxT(0.001f, mat->roughness);
    const float phiHalf = r1*k2Pi;
    const float cosThetaHalf = sqrtf((1.0f - r2)/(1.0f + (sqrf(a) - 1.0f)*r2));
    const float sinThetaHalf = sqrtf(maxT(0.0f, 1.0f - sqrf(cosThetaHalf)));
    const float sinPhiHalf = sinf(phiHalf);
    const float cosPhiHalf = cosf(phiHalf);
    vec3 half = vadd(vadd(vscale(U, sinThetaHalf*cosPhiHalf), vscale(Vt, sinThetaHalf*sinPhiHalf)), vscale(N, cosThetaHalf));
    if (vdot(half, view) <= 0.0f)
        half = vscale(half, -1.0f);
    return vsub(vscale(half, 2.0f*vdot(view, half)), view);
}

static void bsdf_sample(const tinsel_material* mat, float etaI, float etaO, vec3 U, vec3 Vt, vec3 N, vec3 view,
                        vec3* light, float* pdf, int* type, rng_t* rand)                    /* :170-293 */
{
    if (rng_randf(rand) < mat->transmission)
    {
        float F = fresnel(vdot(N, view), etaI, etaO);
        if (rng_randf(rand) < F)
        {
            float r1 = rng_randf01(rand), r2 = rng_randf01(rand);
            *type = eReflected;
            *light = sample_ggx(mat, U, Vt, N, view, r1, r2);
        }
        else
        {
            float eta = etaI/etaO;
            if (refract(view, N, eta, light))
            {
                *type = eSpecular;
                *pdf = (1.0f - F)*mat->transmission;
                return;
            }
            *pdf = 0.0f;
            return;
        }
    }
    else
    {
        float r1 = rng_randf01(rand), r2 = rng_randf01(rand);
        if (rng_randf(rand) < 0.5f)
        {
            if (rng_randf(rand) < mat->subsurface)
            {
                const vec3 d = uniform_sample_hemisphere(rand);
                *light = vsub(vadd(vscale(U, d.x), vscale(Vt, d.y)), vscale(N, d.z));
                *type = eTransmitted;
            }
            else
            {
                const vec3 d = cosine_sample_hemisphere(r1, r2);
                *light = vadd(vadd(vscale(U, d.x), vscale(Vt, d.y)), vscale(N, d.z));
                *type = eReflected;
            }
        }
        else
        {
            *light = sample_ggx(mat, U, Vt, N, view, r1, r2);
            *type = eReflected;
        }
    }
    *pdf = bsdf_pdf(mat, etaI, etaO, N, view, *light);
}

static vec3 bsdf_eval(const tinsel_material* mat, float etaI, float etaO, vec3 N, vec3 V, vec3 L)   /* :296-405 */
{
    float NDotL = vdot(N, L);
    float NDotV = vdot(N, V);
    vec3 H = vnormalize(vadd(L, V));
    float NDotH = vdot(N, H);
    float LDotH = vdot(L, H);

    vec3 Cdlin = v3(mat->color.x, mat->color.y, mat->color.z);
    float Cdlum = (float)(.3*(double)Cdlin.x + .6*(double)Cdlin.y + .1*(double)Cdlin.z);
    vec3 Ctint = Cdlum > 0.0f ? vdivs(Cdlin, Cdlum) : v3s(1.0f);
    /* mat.specular*.08 is a double product narrowed to Real by operator*(Real, Vec3) */
    vec3 Cspec0 = vlerp(vscale(vlerp(v3s(1.0f), Ctint, mat->specular_tint), (float)((double)mat->specular*.08)), Cdlin, mat->metallic);

    vec3 bsdf = v3s(0.0f), brdf = v3s(0.0f);

    if (mat->transmission > 0.0f)
    {
        if (NDotL <= 0)
        {
            float F = fresnel(NDotV, etaI, etaO);
            bsdf = v3s(mat->transmission*(1.0f - F)/absT(NDotL)*(1.0f - mat->metallic));
        }
        else
        {
            float a = maxT(0.001f, mat->roughness);
            float Ds = gtr2(NDotH, a);
            float FH = fresnel(LDotH, etaI, etaO);
            vec3 Fs = vlerp(Cspec0, v3s(1.0f), FH);
            float roughg = a;
            float Gs = smith_ggx(NDotV, roughg)*smith_ggx(NDotL, roughg);
            bsdf = vscale(vscale(Fs, Gs), Ds);
        }
    }

    if (mat->transmission < 1.0f)
    {
        if (NDotL <= 0)
        {
            if (mat->subsurface > 0.0f)
            {
                vec3 s = v3(sqrtf(mat->color.x), sqrtf(mat->color.y), sqrtf(mat->color.z));
                float FL = schlick_fresnel(absT(NDotL)), FV = schlick_fresnel(NDotV);
                float Fd = (1.0f - 0.5f*FL)*(1.0f - 0.5f*FV);
                brdf = vscale(vscale(vscale(vscale(s, kInvPi), mat->subsurface), Fd), 1.0f - mat->metallic);
            }
        }
        else
        {
            float a = maxT(0.001f, mat->roughness);
            float Ds = gtr2(NDotH, a);
            float FH = schlick_fresnel(LDotH);
            vec3 Fs = vlerp(Cspec0, v3s(1.0f), FH);
            float roughg = a;
            float Gs = smith_ggx(NDotV, roughg)*smith_ggx(NDotL, roughg);
            float FL = schlick_fresnel(NDotL), FV = schlick_fresnel(NDotV);
            float Fd90 = (float)(0.5 + (double)(2.0f*LDotH*LDotH*mat->roughness));
            float Fd = lerpf(1.0f, Fd90, FL)*lerpf(1.0f, Fd90, FV);
            float Dr = gtr1(NDotH, (float)(.1 + (.001 - .1)*(double)mat->clearcoat_gloss));
            float Fc = lerpf(.04f, 1.0f, FH);
            float Gr = smith_ggx(NDotL, .25f)*smith_ggx(NDotV, .25f);
            vec3 t1 = vscale(vscale(vscale(Cdlin, kInvPi*Fd), 1.0f - mat->metallic), 1.0f - mat->subsurface);
            vec3 t2 = vscale(vscale(Fs, Gs), Ds);
            brdf = vadd(vadd(t1, t2), v3s(mat->clearcoat*Gr*Fc*Dr));
        }
    }
    return vlerp(brdf, bsdf, mat->transmission);
}

/* ------------------------------------------------------------------------- probe.h / scene.h */

static void probe_dir_to_uv(vec3 dir, float* u, float* v)                                   /* :105-113 */
{
    float theta = acosf(clampT(dir.y, -1.0f, 1.0f));
    float phi = (dir.x == 0.0f && dir.z == 0.0f) ? 0.0f : atan2f(dir.z, dir.x);
    *u = (kPi + phi)*kInvPi*0.5f;
    *v = theta*kInvPi;
}

static vec3 probe_eval(const scene_t* sc, float u, float v)                                 /* :128-134 */
{
    int px = clampI((int)(u*sc->probeW), 0, sc->probeW - 1);
    int py = clampI((int)(v*sc->probeH), 0, sc->probeH - 1);
    tinsel_vec4 c = sc->probeData[py*sc->probeW + px];
    return v3(c.x, c.y, c.z);
}

static float probe_pdf(const scene_t* sc, vec3 d)                                           /* :136-160 */
{
    float u, v;
    probe_dir_to_uv(d, &u, &v);
    int col = clampI((int)(u*sc->probeW), 0, sc->probeW - 1);
    int row = clampI((int)(v*sc->probeH), 0, sc->probeH - 1);
    float pdf = sc->pdfX[row*sc->probeW + col]*sc->pdfY[row];
    float sinTheta = sinf(v*kPi);
    if (fabsf(sinTheta) < 0.0001f)
        pdf = 0.0f;
    else
        pdf *= (float)sc->probeW*(float)sc->probeH/(2.0f*kPi*kPi*sinTheta);
    return pdf;
}

static int lower_bound(const float* a, int lower, int upper, float value)                   /* :185-203 */
{
    while (lower < upper)
    {
        int mid = lower + (upper - lower)/2;
        if (a[mid] < value) lower = mid + 1; else upper = mid;
    }
    return lower;
}

static void probe_sample(const scene_t* sc, vec3* dir, vec3* color, float* pdf, rng_t* rand)    /* :205-236 */
{
    float r1 = rng_randf01(rand), r2 = rng_randf01(rand);
    int row = lower_bound(sc->cdfY, 0, sc->probeH, r1);
    int col = lower_bound(sc->cdfX, row*sc->probeW, (row + 1)*sc->probeW, r2) - row*sc->probeW;
    tinsel_vec4 c = sc->probeData[row*sc->probeW + col];
    *color = v3(c.x, c.y, c.z);
    *pdf = sc->pdfX[row*sc->probeW + col]*sc->pdfY[row];
    float u = col/(float)sc->probeW;
    float v = row/(float)sc->probeH;
    float sinTheta = sinf(v*kPi);
    if (sinTheta == 0.0f)
        *pdf = 0.0f;
    else
        *pdf *= (sc->probeW*sc->probeH)/(2.0f*kPi*kPi*sinTheta);
    /* ProbeUVToDir :115-125 */
    float theta = v*kPi;
    float phi = u*2.0f*kPi;
    *dir = v3(-sinf(theta)*cosf(phi), cosf(theta), -sinf(theta)*sinf(phi));
}

static vec3 sky_eval(const scene_t* sc, vec3 dir)                                           /* scene.h:168-178 */
{
    if (sc->probeValid)
    {
        float u, v;
        probe_dir_to_uv(dir, &u, &v);
        return probe_eval(sc, u, v);
    }
    return vlerp(sc->horizon, sc->zenith, sqrtf(absT(dir.y)));
}

/* ------------------------------------------------------------------------- render.cpp */

/* developer trace of ONE path (port_set_trace(1), scratch/path_probe.py): every decision of the loop as hex floats on stderr */
static int g_trace = 0;
void port_set_trace(int on) { g_trace = on; }
#define TRC(...) do { if (g_trace) fprintf(stderr, __VA_ARGS__); } while (0)
static unsigned fbits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }

static vec3 sample_lights(const scene_t* sc, const tinsel_primitive* surf, float etaI, float etaO, vec3 surfacePos,
                          vec3 surfaceNormal, vec3 shadingNormal, vec3 wo, float time, rng_t* rand, counters* ct)  /* :103-227 */
{
    vec3 sum = v3s(0.0f);

    if (sc->probeValid)
    {
        for (int i = 0; i < kProbeSamples; ++i)
        {
            vec3 skyColor, wi;
            float skyPdf;
            probe_sample(sc, &wi, &skyColor, &skyPdf, rand);
            float t;
            vec3 n;
            ray_t ray = { vadd(surfacePos, vscale(face_forward(surfaceNormal, wi), kRayEpsilon)), wi, time };
            ct->shadow++;
            if (trace(sc, &ray, &t, &n, ct) == NULL)
            {
                float bsdfPdf = bsdf_pdf(&surf->material, etaI, etaO, surfaceNormal, wo, wi);
                vec3 f = bsdf_eval(&surf->material, etaI, etaO, surfaceNormal, wo, wi);
                if (bsdfPdf > 0.0f)
                {
                    int N = (int)(kProbeSamples + kBsdfSamples);
                    float cbsdf = kBsdfSamples/N;
                    float csky = (float)(kProbeSamples)/N;
                    float weight = csky*skyPdf/(cbsdf*bsdfPdf + csky*skyPdf);
                    if (weight > 0.0f)
                        sum = vadd(sum, vdivs(vscale(vmul(vscale(skyColor, weight), f), absT(vdot(wi, surfaceNormal))), skyPdf));
                }
            }
        }
        if (kProbeSamples > 0)
            sum = vdivs(sum, (float)kProbeSamples);
    }

    for (int i = 0; i < sc->numPrims; ++i)
    {
        const tinsel_primitive* light = &sc->prims[i];
        vec3 L = v3s(0.0f);
        int numSamples = light->light_samples;
        if (numSamples == 0)
            continue;

        for (int s = 0; s < numSamples; ++s)
        {
            vec3 lightPos, lightNormal;
            primitive_sample(light, time, &lightPos, &lightNormal, rand);
            vec3 wi = vsub(lightPos, surfacePos);
            float dSq = vdot(wi, wi);
            wi = vdivs(wi, sqrtf(dSq));

            float t;
            vec3 n;
            ray_t ray = { vadd(surfacePos, vscale(face_forward(surfaceNormal, wi), kRayEpsilon)), wi, time };
            ct->shadow++;
            const tinsel_primitive* hit = trace(sc, &ray, &t, &n, ct);
            TRC("    light %d sample %d: pos %08x %08x %08x nrm %08x %08x %08x wi %08x %08x %08x dist %08x -> hit %d t %08x (%g vs %g)\n", i, s, fbits(lightPos.x), fbits(lightPos.y), fbits(lightPos.z),
                fbits(lightNormal.x), fbits(lightNormal.y), fbits(lightNormal.z), fbits(wi.x), fbits(wi.y), fbits(wi.z), fbits(sqrtf(dSq)), hit ? (int)(hit - sc->prims) : -1, fbits(t), t, sqrtf(dSq));
            if (hit)
            {
                float tSq = t*t;
                const float kTolerance = 1.e-2f;
                if (fabsf(t - sqrtf(dSq)) <= kTolerance)
                {
                    const float nl = absT(vdot(lightNormal, wi));
                    if (absT(nl) < 1.e-6f)
                        continue;
                    float lightArea = primitive_area(light);
                    float lightPdf = ((1.0f/lightArea)*tSq)/nl;
                    float bsdfPdf = bsdf_pdf(&surf->material, etaI, etaO, shadingNormal, wo, wi);
                    vec3 f = bsdf_eval(&surf->material, etaI, etaO, shadingNormal, wo, wi);
                    TRC("      in tolerance: nl %08x bsdfPdf %08x etaI %08x etaO %08x wo %08x %08x %08x n %08x %08x %08x\n", fbits(nl), fbits(bsdfPdf), fbits(etaI), fbits(etaO), fbits(wo.x), fbits(wo.y), fbits(wo.z), fbits(shadingNormal.x), fbits(shadingNormal.y), fbits(shadingNormal.z));
                    if (bsdfPdf > 0.0f)
                    {
                        int N = (int)(light->light_samples + kBsdfSamples);
                        float cbsdf = kBsdfSamples/N;
                        float clight = (float)(light->light_samples)/N;
                        float weight = clight*lightPdf/(cbsdf*bsdfPdf + clight*lightPdf);
                        vec3 em = v3(hit->material.emission.x, hit->material.emission.y, hit->material.emission.z);
                        L = vadd(L, vscale(vmul(vscale(f, weight), em), absT(vdot(wi, shadingNormal))/maxT(1.e-3f, lightPdf)));
                        TRC("      arrives: nl %08x lightPdf %08x bsdfPdf %08x f %08x %08x %08x weight %08x L %08x %08x %08x\n", fbits(nl), fbits(lightPdf), fbits(bsdfPdf), fbits(f.x), fbits(f.y), fbits(f.z), fbits(weight), fbits(L.x), fbits(L.y), fbits(L.z));
                    }
                }
            }
        }
        sum = vadd(sum, vscale(L, 1.0f/numSamples));
    }
    return sum;
}

/* Russian roulette: NOT in the reference (render.cpp:250 runs every path to maxDepth) -- an opt-in of the HIP library
 * (tinsel_hip_set_russian_roulette) restated here so that the mode has an oracle too.  0 = off = the reference. */
static int g_rr_start = 0;
void port_set_russian_roulette(int start_bounce) { g_rr_start = start_bounce; }

static vec3 path_trace(const scene_t* sc, vec3 startOrigin, vec3 startDir, float time, int maxDepth, rng_t* rand, counters* ct)  /* :230-388 */
{
    vec3 pathThroughput = v3(1.0f, 1.0f, 1.0f);
    vec3 totalRadiance = v3(0.0f, 0.0f, 0.0f);
    vec3 rayOrigin = startOrigin, rayDir = startDir;
    float rayTime = time;
    float rayEta = 1.0f;
    vec3 rayAbsorption = v3s(0.0f);
    int rayType = eReflected;
    float t = 0.0f;
    vec3 n;
    float bsdfPdf = 1.0f;

    for (int i = 0; i < maxDepth; ++i)
    {
        ray_t ray = { rayOrigin, rayDir, rayTime };
        const tinsel_primitive* hit = trace(sc, &ray, &t, &n, ct);
        TRC("bounce %d: o %08x %08x %08x d %08x %08x %08x -> hit %d t %08x n %08x %08x %08x | thr %08x %08x %08x rad %08x %08x %08x rng %08x %08x\n", i, fbits(rayOrigin.x), fbits(rayOrigin.y), fbits(rayOrigin.z),
            fbits(rayDir.x), fbits(rayDir.y), fbits(rayDir.z), hit ? (int)(hit - sc->prims) : -1, fbits(t), fbits(n.x), fbits(n.y), fbits(n.z), fbits(pathThroughput.x), fbits(pathThroughput.y), fbits(pathThroughput.z),
            fbits(totalRadiance.x), fbits(totalRadiance.y), fbits(totalRadiance.z), rand->seed1, rand->seed2);
        if (hit)
        {
            float outEta;
            vec3 outAbsorption;
            if (rayEta == 1.0f)
            {
                outEta = mat_ior(&hit->material);
                outAbsorption = v3(hit->material.absorption.x, hit->material.absorption.y, hit->material.absorption.z);
            }
            else
            {
                outEta = 1.0f;
                outAbsorption = v3s(0.0f);
            }

            vec3 a = vscale(vneg(rayAbsorption), t);
            pathThroughput = vmul(pathThroughput, v3(expf(a.x), expf(a.y), expf(a.z)));

            const vec3 p = vadd(rayOrigin, vscale(rayDir, t));
            vec3 emission = v3(hit->material.emission.x, hit->material.emission.y, hit->material.emission.z);

            if (i == 0)
            {
                totalRadiance = vadd(totalRadiance, emission);
            }
            else if (kBsdfSamples > 0)
            {
                float lightArea = primitive_area(hit);
                if (lightArea > 0.0f)
                {
                    float lightPdf = ((1.0f/lightArea)*t*t)/clampT(vdot(vneg(rayDir), n), 1.e-3f, 1.0f);
                    int N = (int)(hit->light_samples + kBsdfSamples);
                    float cbsdf = kBsdfSamples/N;
                    float clight = (float)(hit->light_samples)/N;
                    float weight = cbsdf*bsdfPdf/(cbsdf*bsdfPdf + clight*lightPdf);
                    if (rayType == eSpecular)
                        weight = 1.0f;
                    totalRadiance = vadd(totalRadiance, vmul(vscale(pathThroughput, weight), emission));
                }
            }

            totalRadiance = vadd(totalRadiance, vmul(pathThroughput, sample_lights(sc, hit, rayEta, outEta, p, n, n, vneg(rayDir), rayTime, rand, ct)));

            if (hit->light_samples)
                break;

            vec3 u, v;
            basis_from_vector(n, &u, &v);
            vec3 bsdfDir = v3s(0.0f);
            int bsdfType = eReflected;
            bsdf_sample(&hit->material, rayEta, outEta, u, v, n, vneg(rayDir), &bsdfDir, &bsdfPdf, &bsdfType, rand);
            TRC("  bsdf sample: dir %08x %08x %08x pdf %08x type %d rad %08x %08x %08x\n", fbits(bsdfDir.x), fbits(bsdfDir.y), fbits(bsdfDir.z), fbits(bsdfPdf), bsdfType, fbits(totalRadiance.x), fbits(totalRadiance.y), fbits(totalRadiance.z));
            if (bsdfPdf <= 0.0f)
                break;

            vec3 f = bsdf_eval(&hit->material, rayEta, outEta, n, vneg(rayDir), bsdfDir);
            if (vdot(bsdfDir, n) <= 0.0f)
            {
                rayEta = outEta;
                rayType = eTransmitted;
                rayAbsorption = outAbsorption;
            }
            else
            {
                rayType = eReflected;
            }
            pathThroughput = vmul(pathThroughput, vdivs(vscale(f, absT(vdot(n, bsdfDir))), bsdfPdf));
            rayType = bsdfType;
            rayDir = bsdfDir;
            rayOrigin = vadd(p, vscale(face_forward(n, bsdfDir), kRayEpsilon));

            if (g_rr_start > 0 && i + 1 >= g_rr_start && i + 1 < maxDepth)
            {
                float q = minT(1.0f, maxT(pathThroughput.x, maxT(pathThroughput.y, pathThroughput.z)));
                if (!(q > 0.0f))
                    break;
                if (q < 1.0f)
                {
                    float u = rng_randf(rand);
                    if (u >= q)
                        break;
                    pathThroughput = vscale(pathThroughput, 1.0f/q);
                }
            }
        }
        else
        {
            float weight = 1.0f;
            if (sc->probeValid && i > 0 && rayType != eSpecular)
            {
                float skyPdf = probe_pdf(sc, rayDir);
                int N = (int)(kProbeSamples + kBsdfSamples);
                float cbsdf = kBsdfSamples/N;
                float csky = (float)(kProbeSamples)/N;
                weight = cbsdf*bsdfPdf/(cbsdf*bsdfPdf + csky*skyPdf);
            }
            totalRadiance = vadd(totalRadiance, vmul(vscale(sky_eval(sc, rayDir), weight), pathThroughput));
            break;
        }
    }
    return totalRadiance;
}

/* ------------------------------------------------------------------------- util.h / render.h / render.cpp driver */

typedef struct { float r2w[16]; vec3 origin; } camera_sampler;

static void mat_mul(float* result, const float* a, const float* b)                          /* MatrixMultiply<4,4,4> maths.h:83-99 */
{
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
        {
            float t = 0.0f;
            for (int k = 0; k < 4; ++k)
                t += a[i + k*4]*b[k + j*4];
            result[i + j*4] = t;
        }
}

static void camera_sampler_init(camera_sampler* cs, const tinsel_camera* c, int width, int height)  /* util.h:45-71, maths.h:841-849 */
{
    quat q = { c->rotation.x, c->rotation.y, c->rotation.z, c->rotation.w };
    const float s = 1.0f;       /* Transform(camera.position, camera.rotation): s = 1 (render.cpp:451) */
    vec3 c0 = vscale(qrotate(q, v3(1.0f, 0.0f, 0.0f)), s);
    vec3 c1 = vscale(qrotate(q, v3(0.0f, 1.0f, 0.0f)), s);
    vec3 c2 = vscale(qrotate(q, v3(0.0f, 0.0f, 1.0f)), s);
    vec3 c3 = vscale(v3(c->position.x, c->position.y, c->position.z), s);
    float c2w[16] = { c0.x, c0.y, c0.z, 0.0f, c1.x, c1.y, c1.z, 0.0f, c2.x, c2.y, c2.z, 0.0f, c3.x, c3.y, c3.z, 1.0f };
    float r2s[16] = { 2.0f/width, 0.0f, 0.0f, 0.0f, 0.0f, -2.0f/height, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, -1.0f, 1.0f, 1.0f, 1.0f };
    float f = tanf(c->fov*0.5f);
    float aspect = (float)width/height;
    float s2c[16] = { f*aspect, 0.0f, 0.0f, 0.0f, 0.0f, f, 0.0f, 0.0f, 0.0f, 0.0f, -1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f };
    float tmp[16];
    mat_mul(tmp, c2w, s2c);
    mat_mul(cs->r2w, tmp, r2s);
    cs->origin = v3(c2w[12], c2w[13], c2w[14]);
}

static void generate_ray(const camera_sampler* cs, float rx, float ry, vec3* o, vec3* d)   /* util.h:73-79, maths.h:917-924 */
{
    const float vz = 0.0f;
    vec3 p;
    p.x = cs->r2w[0]*rx + cs->r2w[4]*ry + cs->r2w[8]*vz + cs->r2w[12];
    p.y = cs->r2w[1]*rx + cs->r2w[5]*ry + cs->r2w[9]*vz + cs->r2w[13];
    p.z = cs->r2w[2]*rx + cs->r2w[6]*ry + cs->r2w[10]*vz + cs->r2w[14];
    *o = cs->origin;
    *d = vnormalize(vsub(p, *o));
}

static float filter_gaussian(const tinsel_filter* f, float x)                               /* render.h:29-32 */
{
    return maxT(0.0f, (float)(expf(-f->falloff*x*x)) - f->offset);
}

static void add_sample(float* output, int width, int height, float rasterX, float rasterY, float clampv, const tinsel_filter* filter, vec3 sample)  /* render.cpp:401-445 */
{
    int startX = maxI(0, (int)(rasterX - filter->width));
    int startY = maxI(0, (int)(rasterY - filter->width));
    int endX = minI((int)(rasterX + filter->width), width - 1);
    int endY = minI((int)(rasterY + filter->width), height - 1);
    vec3 c = clamp_length(sample, clampv);
    for (int x = startX; x <= endX; ++x)
        for (int y = startY; y <= endY; ++y)
        {
            float* o = output + ((size_t)y*width + x)*4;
            if (filter->type == TINSEL_FILTER_BOX)
            {
                o[0] += c.x; o[1] += c.y; o[2] += c.z; o[3] += 1.0f;
            }
            else
            {
                float w = filter_gaussian(filter, x - rasterX)*filter_gaussian(filter, y - rasterY);
                o[0] += c.x*w; o[1] += c.y*w; o[2] += c.z*w; o[3] += w;
            }
        }
}

/* ------------------------------------------------------------------------- C API (prefix port_) */

void* port_scene_load_pack(const void* data, size_t size)
{
    if (size < sizeof(tinsel_pack_header))
        return NULL;
    scene_t* sc = (scene_t*)calloc(1, sizeof(scene_t));
    sc->blob = (unsigned char*)malloc(size);
    memcpy(sc->blob, data, size);
    tinsel_pack_header hdr;
    memcpy(&hdr, sc->blob, sizeof(hdr));
    if (memcmp(hdr.magic, TINSEL_PACK_MAGIC, 8) != 0 || hdr.version != 1 || hdr.total_bytes > size)
    {
        free(sc->blob);
        free(sc);
        return NULL;
    }
    tinsel_primitive* prims = (tinsel_primitive*)(sc->blob + hdr.off_primitives);
    for (uint32_t i = 0; i < hdr.num_primitives; ++i)
    {
        if (prims[i].type != TINSEL_GEOM_MESH)
            continue;
        tinsel_mesh_geometry* g = &prims[i].geo.mesh;
        g->positions = (const tinsel_vec3*)(sc->blob + (size_t)g->positions);
        g->normals = (const tinsel_vec3*)(sc->blob + (size_t)g->normals);
        g->indices = (const int32_t*)(sc->blob + (size_t)g->indices);
        g->nodes = (const tinsel_bvh_node*)(sc->blob + (size_t)g->nodes);
        g->cdf = (const float*)(sc->blob + (size_t)g->cdf);
    }
    sc->prims = prims;
    sc->numPrims = (int)hdr.num_primitives;
    sc->nodes = (const tinsel_bvh_node*)(sc->blob + hdr.off_bvh_nodes);
    sc->numNodes = (int)hdr.num_bvh_nodes;
    sc->horizon = v3(hdr.sky_horizon.x, hdr.sky_horizon.y, hdr.sky_horizon.z);
    sc->zenith = v3(hdr.sky_zenith.x, hdr.sky_zenith.y, hdr.sky_zenith.z);
    if (hdr.off_probe_data)
    {
        sc->probeValid = 1;
        sc->probeW = hdr.probe_width;
        sc->probeH = hdr.probe_height;
        sc->probeData = (const tinsel_vec4*)(sc->blob + hdr.off_probe_data);
        sc->pdfX = (const float*)(sc->blob + hdr.off_probe_pdf_x);
        sc->cdfX = (const float*)(sc->blob + hdr.off_probe_cdf_x);
        sc->pdfY = (const float*)(sc->blob + hdr.off_probe_pdf_y);
        sc->cdfY = (const float*)(sc->blob + hdr.off_probe_cdf_y);
    }
    sc->camera = hdr.camera;
    sc->options = hdr.options;
    return sc;
}

void port_scene_free(void* h)
{
    scene_t* sc = (scene_t*)h;
    if (!sc)
        return;
    free(sc->blob);
    free(sc);
}

void port_scene_get(void* h, tinsel_camera* cam, tinsel_options* opt)
{
    scene_t* sc = (scene_t*)h;
    *cam = sc->camera;
    *opt = sc->options;
}

uint32_t port_pass_seed(uint32_t i) { return pass_seed(i); }

/* PrimitiveBounds (intersection.h:906-939) with TransformBounds (maths.h:1004-1021) and Union (:1023-1026): the leaf box
 * Scene::Build (scene.cpp:4-16) hands to the scene-level BVHBuilder.  A mesh reads its local box from its BVH root. */
static void transform_bounds(xform x, vec3 lower, vec3 upper, vec3* outLower, vec3* outUpper)
{
    vec3 c0 = qrotate(x.r, v3(1.0f, 0.0f, 0.0f));                  /* Mat33(Quat), maths.h:654-663 */
    vec3 c1 = qrotate(x.r, v3(0.0f, 1.0f, 0.0f));
    vec3 c2 = qrotate(x.r, v3(0.0f, 0.0f, 1.0f));
    vec3 halfEdgeWidth = vscale(vscale(vsub(upper, lower), x.s), 0.5f);       /* xform.s*bounds.GetEdges()*0.5f */
    vec3 ax = vscale(v3(absT(c0.x), absT(c0.y), absT(c0.z)), halfEdgeWidth.x);
    vec3 ay = vscale(v3(absT(c1.x), absT(c1.y), absT(c1.z)), halfEdgeWidth.y);
    vec3 az = vscale(v3(absT(c2.x), absT(c2.y), absT(c2.z)), halfEdgeWidth.z);
    vec3 center = xform_point(x, vscale(vadd(lower, upper), 0.5f));           /* GetCenter(): 0.5*(lower+upper), maths.h:949 */
    *outLower = vsub(vsub(vsub(center, ax), ay), az);
    *outUpper = vadd(vadd(vadd(center, ax), ay), az);
}

void port_primitive_bounds(void* h, int prim, float* out6)
{
    scene_t* sc = (scene_t*)h;
    const tinsel_primitive* p = &sc->prims[prim];
    vec3 lo = v3s(0.0f), hi = v3s(0.0f);
    if (p->type == TINSEL_GEOM_SPHERE)
    {
        lo = v3s(-p->geo.sphere.radius);
        hi = v3s(p->geo.sphere.radius);
    }
    else if (p->type == TINSEL_GEOM_PLANE)
    {
        lo = v3s(-1.e+8f);
        hi = v3s(1.e+8f);
    }
    else if (p->type == TINSEL_GEOM_MESH)
    {
        const tinsel_bvh_node* root = &p->geo.mesh.nodes[0];
        lo = v3(root->lower.x, root->lower.y, root->lower.z);
        hi = v3(root->upper.x, root->upper.y, root->upper.z);
    }
    vec3 sl, su, el, eu;
    transform_bounds(to_xform(&p->start_transform), lo, hi, &sl, &su);
    transform_bounds(to_xform(&p->end_transform), lo, hi, &el, &eu);
    out6[0] = minT(sl.x, el.x); out6[1] = minT(sl.y, el.y); out6[2] = minT(sl.z, el.z);
    out6[3] = maxT(su.x, eu.x); out6[4] = maxT(su.y, eu.y); out6[5] = maxT(su.z, eu.z);
}

void port_leaf_random(uint32_t seed, int n, uint32_t* outRand, float* outRandf)
{
    rng_t a = rng_seeded(seed), b = rng_seeded(seed);
    for (int i = 0; i < n; ++i)
    {
        outRand[i] = rng_rand(&a);
        outRandf[i] = rng_randf(&b);
    }
}

typedef struct {
    const scene_t* sc;
    const tinsel_camera* cam;
    const tinsel_options* opt;
    camera_sampler cs;
    uint32_t passSeed;
    int x0, y0, x1, y1;
    int tid, numThreads;
    int shardRank, shardWorld, shardTile;
    vec3* samples;
    float* rasters;     /* x,y per path; x < -1e29 marks "not generated by this shard" */
    counters ct;
} work_t;

static int pixel_owned(const work_t* w, int i, int j)
{
    if (w->shardWorld <= 1)
        return 1;
    int tilesX = (w->opt->width + w->shardTile - 1)/w->shardTile;
    int t = (j/w->shardTile)*tilesX + (i/w->shardTile);
    return (t % w->shardWorld) == w->shardRank;
}

static void* worker(void* arg)
{
    work_t* w = (work_t*)arg;
    const int W = w->opt->width;
    const int winW = w->x1 - w->x0;
    for (int j = w->y0 + w->tid; j < w->y1; j += w->numThreads)
    {
        for (int i = w->x0; i < w->x1; ++i)
        {
            size_t k = (size_t)(j - w->y0)*winW + (i - w->x0);
            if (!pixel_owned(w, i, j))
            {
                w->samples[k] = v3s(0.0f);
                w->rasters[k*2 + 0] = -1e30f;
                w->rasters[k*2 + 1] = -1e30f;
                continue;
            }
            /* render.cpp:476-486 under the per-path seed contract (render.cu:940) */
            rng_t rand = rng_seeded((uint32_t)i + (uint32_t)j*(uint32_t)W + w->passSeed);
            float x = rng_randf01(&rand);
            float y = rng_randf01(&rand);
            float t = rng_randf01(&rand);
            float time = lerpf(w->cam->shutter_start, w->cam->shutter_end, t);
            x += i;
            y += j;
            vec3 origin, dir;
            generate_ray(&w->cs, x, y, &origin, &dir);
            w->samples[k] = path_trace(w->sc, origin, dir, time, w->opt->max_depth, &rand, &w->ct);
            w->ct.samples++;
            w->rasters[k*2 + 0] = x;
            w->rasters[k*2 + 1] = y;
        }
    }
    return NULL;
}

static double now_seconds(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9*(double)ts.tv_nsec;
}

/* Same contract as ref_render_seeded (oracle/ref_harness.cpp) plus pixel-tile sharding and counters.
 * counts8 (may be NULL): rays, samples, internal visits, tri tests, prim tests, shadow rays, node fetches, - */
double port_render_sharded(void* h, const tinsel_camera* cam, const tinsel_options* opt, uint32_t passBegin, uint32_t numPasses,
                           int x0, int y0, int x1, int y1, float* accum, float* radiance, int numThreads,
                           int shardRank, int shardWorld, int shardTile, uint64_t* counts8)
{
    const scene_t* sc = (const scene_t*)h;
    const int W = opt->width, H = opt->height;
    if (x1 <= x0 || y1 <= y0) { x0 = 0; y0 = 0; x1 = W; y1 = H; }
    const int winW = x1 - x0, winH = y1 - y0;
    if (numThreads < 1)
        numThreads = 1;
    if (numThreads > 1024)
        numThreads = 1024;

    vec3* samples = (vec3*)malloc(sizeof(vec3)*(size_t)winW*winH);
    float* rasters = (float*)malloc(sizeof(float)*2*(size_t)winW*winH);
    work_t* works = (work_t*)calloc((size_t)numThreads, sizeof(work_t));
    pthread_t* threads = (pthread_t*)calloc((size_t)numThreads, sizeof(pthread_t));
    counters total;
    memset(&total, 0, sizeof(total));
    double traceSeconds = 0.0;

    for (uint32_t s = 0; s < numPasses; ++s)
    {
        for (int t = 0; t < numThreads; ++t)
        {
            work_t* w = &works[t];
            memset(w, 0, sizeof(*w));
            w->sc = sc; w->cam = cam; w->opt = opt;
            camera_sampler_init(&w->cs, cam, W, H);     /* render.cpp:450-456 */
            w->passSeed = pass_seed(passBegin + s);
            w->x0 = x0; w->y0 = y0; w->x1 = x1; w->y1 = y1;
            w->tid = t; w->numThreads = numThreads;
            w->shardRank = shardRank; w->shardWorld = shardWorld; w->shardTile = shardTile > 0 ? shardTile : 32;
            w->samples = samples; w->rasters = rasters;
        }
        double t0 = now_seconds();
        if (numThreads == 1)
        {
            worker(&works[0]);
        }
        else
        {
            for (int t = 0; t < numThreads; ++t)
                pthread_create(&threads[t], NULL, worker, &works[t]);
            for (int t = 0; t < numThreads; ++t)
                pthread_join(threads[t], NULL);
        }
        traceSeconds += now_seconds() - t0;

        for (int t = 0; t < numThreads; ++t)
        {
            const counters* c = &works[t].ct;
            total.rays += c->rays; total.samples += c->samples; total.internal += c->internal; total.tris += c->tris;
            total.prims += c->prims; total.shadow += c->shadow; total.fetches += c->fetches;
        }

        if (radiance)
            memcpy(radiance + (size_t)s*winW*winH*3, samples, sizeof(vec3)*(size_t)winW*winH);

        if (accum)
        {
            /* AddSample in raster order, pass by pass (render.cpp:462-490) */
            for (int j = y0; j < y1; ++j)
                for (int i = x0; i < x1; ++i)
                {
                    size_t k = (size_t)(j - y0)*winW + (i - x0);
                    if (rasters[k*2] < -1e29f)
                        continue;
                    add_sample(accum, W, H, rasters[k*2 + 0], rasters[k*2 + 1], opt->clamp, &opt->filter, samples[k]);
                }
        }
    }

    if (counts8)
    {
        counts8[0] = total.rays; counts8[1] = total.samples; counts8[2] = total.internal; counts8[3] = total.tris;
        counts8[4] = total.prims; counts8[5] = total.shadow; counts8[6] = total.fetches; counts8[7] = 0;
    }
    free(samples);
    free(rasters);
    free(works);
    free(threads);
    return traceSeconds;
}

double port_render_seeded(void* h, const tinsel_camera* cam, const tinsel_options* opt, uint32_t passBegin, uint32_t numPasses,
                          int x0, int y0, int x1, int y1, float* accum, float* radiance, int numThreads)
{
    return port_render_sharded(h, cam, opt, passBegin, numPasses, x0, y0, x1, y1, accum, radiance, numThreads, 0, 1, 32, NULL);
}

double port_render_seeded_counts(void* h, const tinsel_camera* cam, const tinsel_options* opt, uint32_t passBegin, uint32_t numPasses,
                                 int x0, int y0, int x1, int y1, float* accum, float* radiance, int numThreads, uint64_t* counts8)
{
    return port_render_sharded(h, cam, opt, passBegin, numPasses, x0, y0, x1, y1, accum, radiance, numThreads, 0, 1, 32, counts8);
}

/* eNormals mode of CpuRenderer::Render (render.cpp:494-515) */
void port_render_normals(void* h, const tinsel_camera* cam, const tinsel_options* opt, float* out)
{
    const scene_t* sc = (const scene_t*)h;
    camera_sampler cs;
    camera_sampler_init(&cs, cam, opt->width, opt->height);
    counters ct;
    memset(&ct, 0, sizeof(ct));
    for (int j = 0; j < opt->height; ++j)
        for (int i = 0; i < opt->width; ++i)
        {
            vec3 o, d;
            generate_ray(&cs, (float)i, (float)j, &o, &d);
            ray_t ray = { o, d, 1.0f };
            float t;
            vec3 n;
            float* px = out + ((size_t)j*opt->width + i)*4;
            if (trace(sc, &ray, &t, &n, &ct))
            {
                n = vadd(vscale(n, 0.5f), v3s(0.5f));
                px[0] = n.x; px[1] = n.y; px[2] = n.z; px[3] = 1.0f;
            }
            else
            {
                px[0] = px[1] = px[2] = px[3] = 0.0f;
            }
        }
}

/* ---------------------------------------------------------------------------------------------
 * Display stage: what main.cpp does with the accumulated pixels every frame (main.cpp:258-282)
 * and the 8-bit conversion of WritePng (png.cpp:323-343).  powf / expf are the host libm's.
 * ------------------------------------------------------------------------------------------- */

static float max_t(float a, float b) { return (a < b) ? b : a; }                              /* maths.h:58-59 */
static float min_t(float a, float b) { return (a < b) ? a : b; }                              /* maths.h:55-56 */

/* ToneMap (util.h:25-42): filmic curve per channel, then SrgbToLinear (maths.h:1551-1555) */
static float tonemap_channel(float c)
{
    float x = max_t(0.0f, c - 0.004f);
    float num = x*(6.2f*x + 0.5f);
    float den = x*(6.2f*x + 1.7f) + 0.06f;     /* Vec3(0.06): the double literal narrows in the float ctor */
    return powf(num/den, 2.2f);
}

/* g_filtered[i] = LinearToSrgb(ToneMap(g_pixels[i]*(exposure/w), limit))   main.cpp:262-271 */
void port_present(const float* pixels, int numPixels, float exposure, float limit, float* filtered)
{
    const float kInvGamma = 1.0f/2.2f;                                                         /* maths.h:1547 */
    (void)limit;                                                                               /* only the disabled Reinhard curve read it */
    for (int i = 0; i < numPixels; ++i)
    {
        const float* p = pixels + 4*i;
        float s = exposure/p[3];
        filtered[4*i + 0] = powf(tonemap_channel(p[0]*s), kInvGamma);
        filtered[4*i + 1] = powf(tonemap_channel(p[1]*s), kInvGamma);
        filtered[4*i + 2] = powf(tonemap_channel(p[2]*s), kInvGamma);
        filtered[4*i + 3] = powf(0.0f, 2.2f);  /* ToneMap rebuilds the colour with w = 0; LinearToSrgb keeps w */
    }
}

/* AverageFilter + NonLocalMeansFilter (nlm.cpp:4-77): windows clipped to the image, walked column by column */
void port_nlm(const float* in, float* out, int width, int height, float falloff, int radius)
{
    float* means = (float*)malloc(sizeof(float)*4*(size_t)width*height);
    for (int y = 0; y < height; ++y)
    {
        for (int x = 0; x < width; ++x)
        {
            int xlower = x - radius > 0 ? x - radius : 0, xupper = x + radius < width - 1 ? x + radius : width - 1;
            int ylower = y - radius > 0 ? y - radius : 0, yupper = y + radius < height - 1 ? y + radius : height - 1;
            int count = 0;
            float sum[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
            for (int fx = xlower; fx <= xupper; ++fx)
                for (int fy = ylower; fy <= yupper; ++fy)
                {
                    for (int c = 0; c < 4; ++c)
                        sum[c] = sum[c] + in[4*(fy*width + fx) + c];
                    count += 1;
                }
            float rc = 1.0f/count;
            for (int c = 0; c < 4; ++c)
                means[4*(y*width + x) + c] = sum[c]*rc;
        }
    }
    for (int y = 0; y < height; ++y)
    {
        for (int x = 0; x < width; ++x)
        {
            int xlower = x - radius > 0 ? x - radius : 0, xupper = x + radius < width - 1 ? x + radius : width - 1;
            int ylower = y - radius > 0 ? y - radius : 0, yupper = y + radius < height - 1 ? y + radius : height - 1;
            float totalWeight = 0.0f;
            float sum[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
            const float* mean = means + 4*(y*width + x);
            for (int fx = xlower; fx <= xupper; ++fx)
                for (int fy = ylower; fy <= yupper; ++fy)
                {
                    const float* m = means + 4*(fy*width + fx);
                    float dx = mean[0] - m[0], dy = mean[1] - m[1], dz = mean[2] - m[2], dw = mean[3] - m[3];
                    float lsq = dx*dx + dy*dy + dz*dz + dw*dw;                                 /* maths.h:331-332 */
                    float weight = expf(-falloff*lsq);
                    for (int c = 0; c < 4; ++c)
                        sum[c] = sum[c] + in[4*(fy*width + fx) + c]*weight;
                    totalWeight += weight;
                }
            float rc = 1.0f/totalWeight;
            for (int c = 0; c < 4; ++c)
                out[4*(y*width + x) + c] = sum[c]*rc;
        }
    }
    free(means);
}

/* the float -> byte step of WritePng (png.cpp:331-343): Quantize(c*255.0 + Randf + Randf - 0.5f), default-seeded stream */
void port_quantize_rgb8(const float* rgba, int width, int height, unsigned char* rgb)
{
    rng_t rand = rng_seeded(0u);
    for (size_t i = 0; i < (size_t)width*height; ++i)
        for (int c = 0; c < 3; ++c)
        {
            double a = rgba[4*i + c]*255.0;
            a = a + rng_randf(&rand);
            a = a + rng_randf(&rand);
            float x = (float)(a - 0.5f);
            rgb[3*i + c] = (unsigned char)min_t(max_t(x, 0.0f), 255.0f);                       /* Quantize, png.cpp:323-326 */
        }
}
