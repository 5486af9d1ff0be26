// tn_isect.h -- ray/primitive tests and the two-level closest-hit traversal.
//
// Semantics follow the CPU oracle exactly (reference src/intersection.h, src/render.cpp:17-62):
//   * scene level: DFS over the scene BVH, near child first, NO closest-t cull   (intersection.h:751-799)
//   * mesh level : DFS with `tChild < tmax` cull, near child first               (intersection.h:678-749)
//   * closest hit keeps the FIRST visited among equal t (`t < minT`, strict)     (render.cpp:45, intersection.h:650)
// Only the memory layout differs (tn_scene.h): one 64-B record per internal visit, leaves in the ref.
#pragma once

#include "tn_scene.h"

namespace tn {

// A/B arms of the flat scan's fetch order (trace_flat, prim_intersect below); every setting gives the same results
#ifndef TN_SCAN_MASK
#define TN_SCAN_MASK 1
#endif
#ifndef TN_QUAD_RECORD
#define TN_QUAD_RECORD 1
#endif
#ifndef TN_SCAN_AHEAD
#define TN_SCAN_AHEAD 1
#endif

// Per-lane traversal stack living in LDS as stack[entry][lane] (conflict-free: consecutive
// lanes hit consecutive banks).  `base` already points at this lane's column.
template <int STRIDE>
struct LdsStack
{
    uint32_t* base;
    TN_D void set(int i, uint32_t v) { base[i*STRIDE] = v; }
    TN_D uint32_t get(int i) const { return base[i*STRIDE]; }
};

struct TraceCounters
{
    uint32_t internal;      // Node64 records visited (= internal BVH nodes, both levels)
    uint32_t tris;          // triangle tests
    uint32_t prims;         // PrimitiveIntersect calls
#ifdef TN_PROFILE_TRACE
    uint32_t cyc[6];        // dev-only: s_memtime deltas of trace_flat by section (boxes, plane, sphere, mesh, finish, fallback walk)
    long long tp;
#endif
};

#ifdef TN_PROFILE_TRACE
#define TN_TTICK0(c) { __builtin_amdgcn_sched_barrier(0); (c).tp = clock64(); __builtin_amdgcn_sched_barrier(0); }
#define TN_TTICK(c, k) { __builtin_amdgcn_sched_barrier(0); const long long _t = clock64(); (c).cyc[k] += (uint32_t)(_t - (c).tp); (c).tp = _t; __builtin_amdgcn_sched_barrier(0); }
#else
#define TN_TTICK0(c)
#define TN_TTICK(c, k)
#endif

#if TN_FAST
TN_D float minf_ref(float a, float b) { return fminf(a, b); }       // tolerance arm: v_min_f32 / v_max_f32
TN_D float maxf_ref(float a, float b) { return fmaxf(a, b); }
#else
TN_D float minf_ref(float a, float b) { return a < b ? a : b; }     // intersection.h:369
TN_D float maxf_ref(float a, float b) { return a > b ? a : b; }     // intersection.h:370
#endif

// IntersectRayAABBFast (intersection.h:373-397)
TN_D bool ray_aabb(V3 pos, V3 rcp, float minx, float miny, float minz, float maxx, float maxy, float maxz, float& t)
{
    float l1 = (minx - pos.x)*rcp.x;
    float l2 = (maxx - pos.x)*rcp.x;
    float lmin = minf_ref(l1, l2);
    float lmax = maxf_ref(l1, l2);

    l1 = (miny - pos.y)*rcp.y;
    l2 = (maxy - pos.y)*rcp.y;
    lmin = maxf_ref(minf_ref(l1, l2), lmin);
    lmax = minf_ref(maxf_ref(l1, l2), lmax);

    l1 = (minz - pos.z)*rcp.z;
    l2 = (maxz - pos.z)*rcp.z;
    lmin = maxf_ref(minf_ref(l1, l2), lmin);
    lmax = minf_ref(maxf_ref(l1, l2), lmax);

    bool hit = ((lmax >= 0.f) & (lmax >= lmin));
    if (hit)
        t = lmin;
    return hit;
}

#ifndef TN_FLAT_MINMAX
#define TN_FLAT_MINMAX 1
#endif

// IntersectRayAABBFast (intersection.h:373-397) with hardware min/max.  Only for rays whose 1/d is finite in
// all three components: then every product below is finite or +-inf, never NaN, and v_min/v_max return what the
// reference's ternaries return up to the sign of a zero, which no comparison below or in the caller can see.
TN_D bool ray_aabb_minmax(V3 pos, V3 rcp, float minx, float miny, float minz, float maxx, float maxy, float maxz, float& t)
{
    float l1 = (minx - pos.x)*rcp.x;
    float l2 = (maxx - pos.x)*rcp.x;
    float lmin = fminf(l1, l2);
    float lmax = fmaxf(l1, l2);

    l1 = (miny - pos.y)*rcp.y;
    l2 = (maxy - pos.y)*rcp.y;
    lmin = fmaxf(fminf(l1, l2), lmin);
    lmax = fminf(fmaxf(l1, l2), lmax);

    l1 = (minz - pos.z)*rcp.z;
    l2 = (maxz - pos.z)*rcp.z;
    lmin = fmaxf(fminf(l1, l2), lmin);
    lmax = fminf(fmaxf(l1, l2), lmax);

    t = lmin;
    return (lmax >= 0.f) & (lmax >= lmin);
}

TN_D bool finite_bits(float x) { return (__float_as_uint(x) & 0x7f800000u) != 0x7f800000u; }

TN_D Node64 load_node(const Node64* nodes, uint32_t idx)
{
    // four 16-B loads of one 64-B aligned record
    const float4* p = reinterpret_cast<const float4*>(nodes + idx);
    float4 a = p[0], b = p[1], c = p[2], d = p[3];
    Node64 n;
    n.lminx = a.x; n.lminy = a.y; n.lminz = a.z; n.lmaxx = a.w;
    n.lmaxy = b.x; n.lmaxz = b.y; n.rminx = b.z; n.rminy = b.w;
    n.rminz = c.x; n.rmaxx = c.y; n.rmaxy = c.z; n.rmaxz = c.w;
    n.left = __float_as_uint(d.x);
    n.right = __float_as_uint(d.y);
    return n;
}

// IntersectRaySphere + SolveQuadratic (intersection.h:30-83), a == 1
TN_D bool ray_sphere(V3 center, float radius, V3 o, V3 d, float& outT, V3& outN)
{
    V3 q = o - center;
    float a = 1.0f;
    float b = 2.0f*dot(q, d);
    float c = dot(q, q) - (radius*radius);

    float disc = b*b - 4.0f*a*c;
    if (disc < 0.0f)
        return false;

    float sgn = (b < 0.0f) ? -1.0f : 1.0f;                 // Sign (maths.h:43)
    float tt = -0.5f*(b + sgn*sqrtf_cr(disc));
    float t0 = tt/a;
    float t1 = c/tt;
    if (t1 < t0) { float tmp = t0; t0 = t1; t1 = tmp; }    // Sort2 (intersection.h:9-13)

    if (t0 < 0.0f && t1 < 0.0f)
        return false;
    if (t0 < 0.0f && t1 > 0.0f)
        t0 = t1;

    outN = normalize((o + d*t0) - center);
    outT = t0;
    return true;
}

// IntersectRayPlane (intersection.h:85-99); Dot(Vec4,Vec4) keeps the w term (maths.h:331)
TN_D bool ray_plane(V3 p, V3 dir, float px, float py, float pz, float pw, float& t)
{
    float d = px*dir.x + py*dir.y + pz*dir.z + pw*0.0f;
    if (d == 0.0f)
        return false;
    t = -(px*p.x + py*p.y + pz*p.z + pw*1.0f)/d;
    return t > 0.0f;
}

// IntersectRayTriTwoSided (intersection.h:117-145)
TN_D bool ray_tri(V3 p, V3 dir, V3 a, V3 b, V3 c, float& t, float& u, float& v, float& w, float& sign, V3& n)
{
    V3 ab = b - a;
    V3 ac = c - a;
    n = cross(ab, ac);

    V3 nd = -dir;
    float d = dot(nd, n);
    float ood = rcpf_cr(d);
    V3 ap = p - a;

    t = dot(ap, n)*ood;
    if (t < 0.0f)
        return false;

    V3 e = cross(nd, ap);
    v = dot(ac, e)*ood;
    if (v < 0.0f || v > 1.0f)
        return false;
    w = -dot(ab, e)*ood;
    if (w < 0.0f || v + w > 1.0f)
        return false;

    u = 1.0f - v - w;
    sign = d;
    return true;
}

struct MeshHit
{
    float t, u, v, w;
    int tri;
    V3 n;
};

// IntersectRayMesh + MeshQuery (intersection.h:629-749).  `sp` = first free stack slot.
// ray_mesh for a tree that is ONE internal node over two one-triangle leaves (quads: cornell's lamp, veach's plates -- meshes
// every shadow ray or every ray enters).  The loop below unrolled for that tree: the same box tests on the same record, the
// near child first exactly as the stack would pop it, both leaves tested when their boxes are hit (the `tChild < tmax` cull is
// evaluated before any triangle is: tmax is still FLT_MAX), strict `<` between the two hits.  No stack, no loop.
template <bool COUNT>
TN_D bool ray_mesh_two_leaves(const Node64* mnodes, const Tri48* mtris, uint32_t mroot, V3 o, V3 d, V3 rcp, MeshHit& hit, TraceCounters& ctr)
{
    // rcp = rcp3_cr(d): MeshQuery's rcpDir (intersection.h:669), the caller's (pose_inv_ray)
    const Node64 nd = load_node(mnodes, mroot);
    if (COUNT) ctr.internal++;

    float tL, tR;
    const bool hL = ray_aabb(o, rcp, nd.lminx, nd.lminy, nd.lminz, nd.lmaxx, nd.lmaxy, nd.lmaxz, tL) && tL < kFltMax;
    const bool hR = ray_aabb(o, rcp, nd.rminx, nd.rminy, nd.rminz, nd.rmaxx, nd.rmaxy, nd.rmaxz, tR) && tR < kFltMax;

    // ray_mesh pushes `first` (if the left box is hit) then `second` (if the right one is) and pops the last pushed first
    uint32_t first = nd.left, second = nd.right;
    if (hL && hR && (tL < tR))
    {
        first = nd.right;
        second = nd.left;
    }
    float closestT = kFltMax;
    auto leaf = [&](uint32_t ref) {
        const int i = (int)(ref & ~kLeafBit);
        const float4* tp = reinterpret_cast<const float4*>(mtris + i);
        const float4 ta = tp[0], tb = tp[1], tc = tp[2];
        if (COUNT) ctr.tris++;
        float t, u, v, w, sign;
        V3 n;
        if (ray_tri(o, d, V3(ta.x, ta.y, ta.z), V3(tb.x, tb.y, tb.z), V3(tc.x, tc.y, tc.z), t, u, v, w, sign, n))
        {
            if (t > 0.0f && t < closestT)
            {
                closestT = t;
                hit.t = t; hit.u = u; hit.v = v; hit.w = w;
                hit.tri = i;
                hit.n = n*sign;
            }
        }
    };
    if (hR)
        leaf(second);
    if (hL)
        leaf(first);
    return closestT < kFltMax;
}

// ANYHIT / tStop (shadow rays, shadow_stop below): the walk may stop at the first accepted hit with t < tStop.
template <class Stack, bool COUNT, bool ANYHIT = false>
TN_D bool ray_mesh(const Node64* mnodes, const Tri48* mtris, uint32_t mroot, Stack& st, int sp, V3 o, V3 d, V3 rcp, MeshHit& hit, TraceCounters& ctr, float tStop = 0.0f)
{
    float closestT = kFltMax;
    float tmax = kFltMax;
    // rcp = rcp3_cr(d), the caller's (pose_inv_ray)

    const int base = sp;
    st.set(sp++, mroot);

    while (sp > base)
    {
        uint32_t ref = st.get(--sp);

        if (ref & kLeafBit)
        {
            const int i = (int)(ref & ~kLeafBit);
            const float4* tp = reinterpret_cast<const float4*>(mtris + i);
            float4 ta = tp[0], tb = tp[1], tc = tp[2];
            if (COUNT) ctr.tris++;

            float t, u, v, w, sign;
            V3 n;
            if (ray_tri(o, d, V3(ta.x, ta.y, ta.z), V3(tb.x, tb.y, tb.z), V3(tc.x, tc.y, tc.z), t, u, v, w, sign, n))
            {
                if (t > 0.0f && t < closestT)
                {
                    closestT = t;
                    hit.t = t; hit.u = u; hit.v = v; hit.w = w;
                    hit.tri = i;
                    hit.n = n*sign;
                    if (ANYHIT && t < tStop)
                        break;          // shadow ray: an occluder this far in front of the light decides it (shadow_stop)
                }
            }
            tmax = closestT;    // "truncate ray" (intersection.h:701-702)
        }
        else
        {
            Node64 nd = load_node(mnodes, ref);
            if (COUNT) ctr.internal++;

            float tL, tR;
            bool hL = ray_aabb(o, rcp, nd.lminx, nd.lminy, nd.lminz, nd.lmaxx, nd.lmaxy, nd.lmaxz, tL) && tL < tmax;
            bool hR = ray_aabb(o, rcp, nd.rminx, nd.rminy, nd.rminz, nd.rmaxx, nd.rmaxy, nd.rmaxz, tR) && tR < tmax;

            uint32_t first = nd.left, second = nd.right;
            if (hL && hR && (tL < tR))      // traverse closest first: push far, then near
            {
                first = nd.right;
                second = nd.left;
            }
            if (hL)
                st.set(sp++, first);
            if (hR)
                st.set(sp++, second);
        }
    }
    return closestT < kFltMax;
}

TN_D Xform prim_pose(const DevScene& sc, const Prim64& p, float time)
{
    Xform x;
    if (p.flags & kPrimMoving)
    {
        const float4* mp = reinterpret_cast<const float4*>(sc.moving + p.moving);
        float4 a0 = mp[0], a1 = mp[1], b0 = mp[2], b1 = mp[3];
        Xform s, e;
        s.p = V3(a0.x, a0.y, a0.z); s.s = a0.w; s.r = { a1.x, a1.y, a1.z, a1.w };
        e.p = V3(b0.x, b0.y, b0.z); e.s = b0.w; e.r = { b1.x, b1.y, b1.z, b1.w };
        x = interpolate_xform(s, e, time);
    }
    else
    {
        x.p = V3(p.px, p.py, p.pz);
        x.s = p.s;
        x.r = { p.rx, p.ry, p.rz, p.rw };
    }
    return x;
}

// Rotate() by the identity quaternion, written down.  Most primitives of most scenes are not rotated: their pose carries the quaternion
// (+0, +0, +0, 1), and the reference still runs two quaternion products per Rotate() on it (maths.h:558-563: 56 multiplies and adds; a mesh
// primitive's ray costs two of them per Trace(), a light sample two more).  With that q every product term but one is a signed zero:
//   * forward, Rotate(q, v) (TransformVector / TransformPoint): a non-zero component passes through exactly and a zero one comes out
//     as +0 whatever its sign and its neighbours' (each partial sum starts at +0 and only ever adds signed zeros): v + 0.0f per
//     component -- for finite v (0 * inf would be NaN);
//   * inverse, Rotate(Conjugate(q), v) = Rotate((-0, -0, -0, 1), v) (InverseTransform*): non-zero components pass through exactly; the sign
//     of a zero one depends on its neighbours' signs, so a vector with a zero component takes the reference's formula.
// The flag is set on the host from the bits of the stored quaternion (kPrimNoRot, set_prim_derived); a wave takes the short form only when
// all its active lanes can (the flat scan's waves are at one primitive at a time).  tests: every fixture has such primitives; axis-aligned
// rays (one-pixel frames) and axis-aligned light normals take the zero-component branches.
TN_D bool finite_bits3(V3 v) { return finite_bits(v.x) && finite_bits(v.y) && finite_bits(v.z); }
TN_D bool finite_nonzero(float x) { return ((__float_as_uint(x) & 0x7fffffffu) - 1u) < 0x7f7fffffu; }
TN_D bool finite_nonzero3(V3 v) { return finite_nonzero(v.x) && finite_nonzero(v.y) && finite_nonzero(v.z); }

TN_D V3 pose_rotate(const Prim64& p, const Xform& x, V3 v)              // Rotate(x.r, v)
{
    if (__all((p.flags & kPrimNoRot) != 0u && finite_bits3(v)))
        return V3(v.x + 0.0f, v.y + 0.0f, v.z + 0.0f);
    return qrotate(x.r, v);
}
TN_D V3 pose_xform_vector(const Prim64& p, const Xform& x, V3 v) { return pose_rotate(p, x, x.s*v); }              // TransformVector (maths.h:601-604)
TN_D V3 pose_xform_point(const Prim64& p, const Xform& x, V3 v) { return x.p + pose_rotate(p, x, x.s*v); }         // TransformPoint (maths.h:606-609)

// InverseTransformPoint(o), InverseTransformVector(d) (maths.h:611-619): a mesh primitive's ray into mesh space, and the reciprocal of its
// direction (MeshQuery, intersection.h:669) -- which IS the world ray's, `rcpWorld` = rcp3_cr(d) (Trace computes it for the scene-level
// boxes), where the mesh-space direction is d bit for bit: no rotation and 1.0f/s == 1 (1.0f*x is x)
TN_D void pose_inv_ray(const Prim64& p, const Xform& x, V3 o, V3 d, V3& lo, V3& ld, V3& lrcp, V3 rcpWorld = V3(0.0f), bool haveRcpWorld = false)
{
    // 1.0f/s: a static mesh has it in its record
    const float rs = (p.flags & kPrimMoving) ? rcpf_cr(x.s) : p.g3;
    const V3 op = o - x.p;
    if (__all((p.flags & kPrimNoRot) != 0u && finite_nonzero3(op) && finite_nonzero3(d)))
    {
        lo = rs*op;
        ld = rs*d;
        if (haveRcpWorld && __all(rs == 1.0f))
        {
            lrcp = rcpWorld;
            return;
        }
    }
    else
    {
        const Q4 c = qconj(x.r);
        lo = rs*qrotate(c, op);
        ld = rs*qrotate(c, d);
    }
    lrcp = rcp3_cr(ld);
}

TN_D Prim64 load_prim(const Prim64* prims, int idx)
{
    const float4* pp = reinterpret_cast<const float4*>(prims + idx);
    float4 a = pp[0], b = pp[1], c = pp[2], d = pp[3];
    Prim64 p;
    p.px = a.x; p.py = a.y; p.pz = a.z; p.s = a.w;
    p.rx = b.x; p.ry = b.y; p.rz = b.z; p.rw = b.w;
    p.g0 = c.x; p.g1 = c.y; p.g2 = c.z; p.g3 = c.w;
    p.type = __float_as_uint(d.x); p.flags = __float_as_uint(d.y);
    p.mesh = __float_as_uint(d.z); p.moving = __float_as_uint(d.w);
    return p;
}

// the same record through the constant-address-space pointer, at a wave-uniform index (scalar loads)
TN_D Prim64 load_prim_uniform(ConstF4 prims, int idx)
{
    ConstF4 pp = prims + (size_t)idx*4;
    const ConstF4V a = pp[0], b = pp[1], c = pp[2], d = pp[3];
    Prim64 p;
    p.px = a.x; p.py = a.y; p.pz = a.z; p.s = a.w;
    p.rx = b.x; p.ry = b.y; p.rz = b.z; p.rw = b.w;
    p.g0 = c.x; p.g1 = c.y; p.g2 = c.z; p.g3 = c.w;
    p.type = __float_as_uint(d.x); p.flags = __float_as_uint(d.y);
    p.mesh = __float_as_uint(d.z); p.moving = __float_as_uint(d.w);
    return p;
}

// PrimitiveIntersect (intersection.h:951-1020)
// UNIFORM: `index` is the same in every lane (the flat scan's loop counter)
template <class SC, class Stack, bool COUNT, bool ANYHIT = false, bool UNIFORM = false>
TN_D bool prim_intersect(const SC& sc, int index, Stack& st, int sp, V3 o, V3 d, float time, float& outT, V3& outN, TraceCounters& ctr, float tStop = 0.0f,
                         V3 rcpWorld = V3(0.0f), bool haveRcpWorld = false, const Prim64* fetched = nullptr)
{
    // (`fetched`: the flat scan's record, requested together with the leaf box)
    const Prim64 p = fetched ? *fetched : UNIFORM ? load_prim_uniform(sc.kPrims, index) : load_prim(sc.prims, index);
    if (COUNT) ctr.prims++;

    if (p.type == kPrimPlane)
    {
        // the reference interpolates the pose here too but the plane test never reads it
        bool hit = ray_plane(o, d, p.g0, p.g1, p.g2, p.g3, outT);
        if (hit)
            outN = V3(p.g0, p.g1, p.g2);
        return hit;
    }

    const Xform x = prim_pose(sc, p, time);

    if (p.type == kPrimSphere)
    {
        return ray_sphere(x.p, p.g0*x.s, o, d, outT, outN);
    }

    // mesh: ray into mesh space
    V3 lo, ld, lrcp;
    pose_inv_ray(p, x, o, d, lo, ld, lrcp, rcpWorld, haveRcpWorld);

    MeshHit h;
    const Tri48* mtris;
    const float* nr;
#if TN_QUAD_RECORD
    if (!(SC::kWalkedOnly && !SC::kQuadsInline) && (p.flags & kPrimQuadArena))
    {
        // a quad in the arena: node, triangles and normals at the offsets its primitive record carries (no mesh-table record in between)
        const unsigned char* base = (SC::kLds || sc.arenaLdsBytes != 0u) ? sc.ldsBase : sc.arena;
        const QuadOffsets q = quad_offsets(__float_as_uint(p.g0), __float_as_uint(p.g1));
        mtris = reinterpret_cast<const Tri48*>(base + q.tris);
        nr = reinterpret_cast<const float*>(base + q.normals);
        if (!ray_mesh_two_leaves<COUNT>(reinterpret_cast<const Node64*>(base + q.nodes), mtris, 0u, lo, ld, lrcp, h, ctr))
            return false;
    }
    else
#endif
    {
    const DevMesh m = sc.meshes[p.mesh];
    mtris = mesh_tris(sc, m);
    nr = mesh_normals(sc, m);
    const bool fromRecord = SC::kQuadsInline ? (p.flags & kPrimWalked) != 0u
                          : SC::kWalkedOnly || (!COUNT && sc.walkRec != nullptr && (p.flags & kPrimWalked));
    if (fromRecord)
    {
        // the mesh-space closest hit of this (ray, primitive) was computed by k_walk with the same ray_mesh arithmetic
        // on the same lo / ld (tn_walk.h); t == FLT_MAX marks "no hit" (ray_mesh's own `closestT < FLT_MAX`)
        const uint32_t kb = (p.flags >> kPrimWalkLaneShift) & 7u;
        const float4* rp = sc.walkRec + (size_t)(sc.walkItem + kb)*2;
        const float4 ra = rp[0];
        if (!(ra.x < kFltMax))
            return false;
        const float4 rb = rp[1];
        h.t = ra.x; h.u = ra.y; h.v = ra.z; h.w = ra.w;
        h.n = V3(rb.x, rb.y, rb.z);
        h.tri = __float_as_int(rb.w);
    }
    else if (SC::kWalkedOnly && !SC::kQuadsInline)
        return false;
    else if (m.twoLeaves)
    {
        if (!ray_mesh_two_leaves<COUNT>(mesh_nodes(sc, m), mtris, m.root, lo, ld, lrcp, h, ctr))
            return false;
    }
    else if (SC::kWalkedOnly)
        return false;               // (the host runs these variants only where every other mesh is walked by k_walk)
    else if (!ray_mesh<Stack, COUNT, ANYHIT>(mesh_nodes(sc, m), mtris, m.root, st, sp, lo, ld, lrcp, h, ctr, tStop))
        return false;
    }

    // interpolate vertex normals (intersection.h:996-1012)
    const float4* tp = reinterpret_cast<const float4*>(mtris + h.tri);
    const int i0 = __float_as_int(tp[0].w), i1 = __float_as_int(tp[1].w), i2 = __float_as_int(tp[2].w);
    V3 n1(nr[i0*3 + 0], nr[i0*3 + 1], nr[i0*3 + 2]);
    V3 n2(nr[i1*3 + 0], nr[i1*3 + 1], nr[i1*3 + 2]);
    V3 n3(nr[i2*3 + 0], nr[i2*3 + 1], nr[i2*3 + 2]);

    V3 smooth = h.u*n1 + h.v*n2 + h.w*n3;
    if (dot(smooth, h.n) < 0.0f)
        smooth = smooth*(-1.0f);

    outT = h.t;
    outN = safe_normalize(pose_xform_vector(p, x, smooth), h.n);
    return true;
}

// Scene level as a WAVE-UNIFORM scan (scenes with few primitives -- every BASELINE config).
//
// The oracle's QueryBVH (intersection.h:751-799) has no closest-t cull at scene level: it calls
// PrimitiveIntersect for exactly the primitives whose leaf box (and hence, the slab test being
// monotone under box inclusion, every ancestor box) the ray hits, and keeps the smallest t > 0, the
// FIRST VISITED winning exact ties (render.cpp:45).  So the set of tests and the winner do not depend
// on the visit order unless two candidate hits tie.  This scan runs the same leaf-box tests and the
// same PrimitiveIntersect calls in primitive order -- every lane is at the same primitive at the same
// time, so the primitive records are wave-uniform loads, the type switch does not diverge and there is
// no stack traffic -- and reports `tie` when two accepted hits are (nearly) equal; the caller then
// re-traces that ray with the BVH walk, which IS the oracle's order.  Results are therefore identical
// to the BVH walk in all cases.
TN_D unsigned long long wave_uniform64(unsigned long long v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

template <class SC, class Stack, bool COUNT, bool ANYHIT = false>
TN_D int trace_flat(const SC& sc, Stack& st, V3 o, V3 d, V3 rcp, float time, float& outT, V3& outN, bool& tie, TraceCounters& ctr, float tStop = 0.0f)
{
    float minT = kFltMax;
    int closest = -1;
    V3 cn;
    tie = false;

    auto accept = [&](int i, float t, V3 n) {
        if (t > 0.0f)
        {
            // two accepted hits closer than a few ulps: let the oracle's visit order decide.  Whatever the order of the
            // calls, every hit within that distance of the final minimum is compared with a running minimum that lies
            // between the two, so the flag does not depend on the order either.
            if (fabsf(t - minT) <= 1e-5f*fabsf(t))
                tie = true;
            if (t < minT)
            {
                minT = t;
                closest = i;
                cn = n;
            }
        }
    };

    // Planes and spheres are tested as the scan meets them (the record is a wave-uniform load, the kind does not diverge).
    // In a scene with several mesh primitives (sc.deferMeshes) the meshes whose leaf box the ray enters are only NOTED:
    // walked one after the other inside this loop, each would run with the few lanes that enter that one box; walked
    // afterwards, every lane with a mesh left takes ITS next one and they all walk together (veach.tin, three plates:
    // 1295 -> 1377 Msamples/s; with ONE mesh there is nothing to merge and the second loop only costs: cornell -3.5 %).
    unsigned long long meshes = 0;
    // leaf-box tests with hardware min / max when no 0*inf can occur in the wave (every lane's origin and 1/d finite: k_walk's rule,
    // ray_aabb_minmax above) -- 12 instructions per box instead of 24; build with -DTN_FLAT_MINMAX=0 to keep the ternaries (A/B)
    const bool finiteAll = TN_FLAT_MINMAX && __all(finite_bits(rcp.x) && finite_bits(rcp.y) && finite_bits(rcp.z) &&
                                                  finite_bits(o.x) && finite_bits(o.y) && finite_bits(o.z));
    TN_TTICK0(ctr)
    // The scene's always-hit planes FOUR AT A TIME, ahead of the loop (DevScene::planeEq: the same equations, padded with planes no ray
    // meets; their boxes say 2 and the loop below passes them by).  Every lane is at the same primitive in this scan, so one
    // IntersectRayPlane after the other is one IEEE division's dependent chain after the other; four in one block are four independent
    // chains the scheduler interleaves (then two, then one: what is left).  The order of the accept() calls does not matter (above).
    const int numPlanes = sc.numPlanes;
    int pk = 0;
    for (; pk + 4 <= numPlanes; pk += 4)
    {
        const ConstF4V e0 = sc.kPlaneEq[pk], e1 = sc.kPlaneEq[pk + 1], e2 = sc.kPlaneEq[pk + 2], e3 = sc.kPlaneEq[pk + 3];
        const ConstF4V id = sc.kPlaneIdx[pk >> 2];
        float t0, t1, t2, t3;
        const bool h0 = ray_plane(o, d, e0.x, e0.y, e0.z, e0.w, t0);
        const bool h1 = ray_plane(o, d, e1.x, e1.y, e1.z, e1.w, t1);
        const bool h2 = ray_plane(o, d, e2.x, e2.y, e2.z, e2.w, t2);
        const bool h3 = ray_plane(o, d, e3.x, e3.y, e3.z, e3.w, t3);
        if (h0) accept(__float_as_int(id.x), t0, V3(e0.x, e0.y, e0.z));
        if (h1) accept(__float_as_int(id.y), t1, V3(e1.x, e1.y, e1.z));
        if (h2) accept(__float_as_int(id.z), t2, V3(e2.x, e2.y, e2.z));
        if (h3) accept(__float_as_int(id.w), t3, V3(e3.x, e3.y, e3.z));
    }
    if (pk < numPlanes)
    {
        // the one, two or three that are left (the table is padded to a multiple of four: the loads are in bounds)
        const ConstF4V id = sc.kPlaneIdx[pk >> 2];
        const int left = numPlanes - pk;
        if (left >= 2)
        {
            const ConstF4V e0 = sc.kPlaneEq[pk], e1 = sc.kPlaneEq[pk + 1];
            float t0, t1;
            const bool h0 = ray_plane(o, d, e0.x, e0.y, e0.z, e0.w, t0);
            const bool h1 = ray_plane(o, d, e1.x, e1.y, e1.z, e1.w, t1);
            if (h0) accept(__float_as_int(id.x), t0, V3(e0.x, e0.y, e0.z));
            if (h1) accept(__float_as_int(id.y), t1, V3(e1.x, e1.y, e1.z));
        }
        if (left & 1)
        {
            const ConstF4V e0 = sc.kPlaneEq[pk + (left & 2)];
            float t0;
            if (ray_plane(o, d, e0.x, e0.y, e0.z, e0.w, t0))
                accept(__float_as_int(left == 1 ? id.x : id.z), t0, V3(e0.x, e0.y, e0.z));
        }
    }
    // The primitives that are not in the plane table (DevScene::scanMask), as a scalar bit loop: the loop over all primitives fetched a table
    // plane's leaf box only to read "pass by" (-DTN_SCAN_MASK=0: the A/B arm).  And a visited primitive's RECORD is requested together with its
    // box, one scalar round trip instead of two dependent ones (-DTN_SCAN_AHEAD=0: fetched behind the box test).
#if TN_SCAN_MASK
    for (unsigned long long todo = wave_uniform64(sc.scanMask); todo != 0ull; todo &= todo - 1ull)
    {
        const int i = (int)__builtin_ctzll(todo);
#else
    for (int i = 0; i < sc.numPrims; ++i)
    {
#endif
        TN_TTICK(ctr, 4)
        ConstF4V b0 = sc.kBoxes[i*2], b1 = sc.kBoxes[i*2 + 1];
#if TN_SCAN_AHEAD
        ConstF4V ra = sc.kPrims[i*4], rb = sc.kPrims[i*4 + 1], rc = sc.kPrims[i*4 + 2], rd = sc.kPrims[i*4 + 3];
        asm volatile("" : "+s"(b0), "+s"(b1), "+s"(ra), "+s"(rb), "+s"(rc), "+s"(rd));        // (all six in flight before the first is used)
        Prim64 rec;
        rec.px = ra.x; rec.py = ra.y; rec.pz = ra.z; rec.s = ra.w;
        rec.rx = rb.x; rec.ry = rb.y; rec.rz = rb.z; rec.rw = rb.w;
        rec.g0 = rc.x; rec.g1 = rc.y; rec.g2 = rc.z; rec.g3 = rc.w;
        rec.type = __float_as_uint(rd.x); rec.flags = __float_as_uint(rd.y);
        rec.mesh = __float_as_uint(rd.z); rec.moving = __float_as_uint(rd.w);
#endif
#if !TN_SCAN_MASK
        if (__float_as_uint(b1.z) == 2u)
            continue;
#endif
        if (__float_as_uint(b1.z) == 0u)
        {
            float tb;
            if (!(finiteAll ? ray_aabb_minmax(o, rcp, b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, tb) : ray_aabb(o, rcp, b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, tb)))
                continue;
        }
        TN_TTICK(ctr, 0)
#if TN_SCAN_AHEAD
        if (SC::kDefer != 0 && (SC::kDefer == 1 || sc.deferMeshes) && rec.type == (uint32_t)kPrimMesh)
#else
        if (SC::kDefer != 0 && (SC::kDefer == 1 || sc.deferMeshes) && __float_as_uint(sc.kPrims[i*4 + 3].x) == (uint32_t)kPrimMesh)
#endif
        {
            meshes |= 1ull << i;
            continue;
        }
        float t;
        V3 n;
#if TN_SCAN_AHEAD
        const bool primHit = prim_intersect<SC, Stack, COUNT, ANYHIT, true>(sc, i, st, 0, o, d, time, t, n, ctr, tStop, rcp, true, &rec);
#else
        const bool primHit = prim_intersect<SC, Stack, COUNT, ANYHIT, true>(sc, i, st, 0, o, d, time, t, n, ctr, tStop, rcp, true);
#endif
#ifdef TN_PROFILE_TRACE
        { const uint32_t ty = __builtin_amdgcn_readfirstlane(__float_as_uint(reinterpret_cast<const float4*>(sc.prims + i)[3].x)); TN_TTICK(ctr, ty == kPrimPlane ? 1 : ty == kPrimSphere ? 2 : 3) }
#endif
        if (primHit)
            accept(i, t, n);
        if (ANYHIT && minT < tStop)
        {
            meshes = 0;                 // decided (shadow_stop): nothing further can change what the caller does with it
            break;
        }
    }
    if (SC::kDefer != 0)
    {
        while (meshes)
        {
            const int i = (int)__builtin_ctzll(meshes);
            meshes &= meshes - 1ull;
            float t;
            V3 n;
            if (prim_intersect<SC, Stack, COUNT, ANYHIT>(sc, i, st, 0, o, d, time, t, n, ctr, tStop, rcp, true))
                accept(i, t, n);
            if (ANYHIT && minT < tStop)
                break;
        }
        TN_TTICK(ctr, 3)
    }

    outT = minT;
    outN = face_forward(cn, -d);
    return closest;
}

// Does this ray pass the leaf-box test of any primitive that is NOT an always-hit plane (flat-scan scenes)?  The
// producer of a ray queue sorts by this bit: a wave of rays that can only hit the planes skips the sphere and mesh code
// of trace_flat altogether, and the waves that do run it have their lanes on it.  The same box tests as trace_flat's.
template <class SC>
TN_D bool ray_meets_bounded_prim(const SC& sc, V3 o, V3 d)
{
    const V3 rcp = rcp3_cr(d);
    bool any = false;
    for (int i = 0; i < sc.numPrims; ++i)
    {
        const float4* bp = reinterpret_cast<const float4*>(sc.primBoxes + i);
        const float4 b0 = bp[0], b1 = bp[1];
        if (__float_as_uint(b1.z) != 0u)
            continue;
        float tb;
        any = any || ray_aabb(o, rcp, b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, tb);
    }
    return any;
}

// Rays the flat scan handles; the others take the scene BVH walk (which does not box-test a root leaf), so the queue
// sort and k_walk (tn_walk.h) treat them as entering every walked mesh.
TN_D bool ray_sane(V3 o) { return fabsf(o.x) < 1e6f && fabsf(o.y) < 1e6f && fabsf(o.z) < 1e6f; }

// Trace (render.cpp:17-62) over QueryBVH (intersection.h:751-799).
// Returns the primitive index or -1; outN is already FaceForward(n, -dir) (render.cpp:59).
// What SampleLights does with a shadow ray's closest hit (render.cpp:118, 175-196): a probe sample counts iff NOTHING is hit, a
// light sample iff the closest hit lies within 1e-2 of the sampled point's distance.  So a hit well in front of the light --
// closer than dist by more than the tolerance, with margin for the rounding of the two subtractions -- decides the sample
// whatever else the ray would meet, and the traversal may stop there (ANYHIT traces).  The result the caller derives is the
// oracle's; what is NOT the oracle's is the (t, primitive) a stopped trace returns, which the callers use only to decide.
TN_D float shadow_stop(float dist)
{
    return dist < 0.0f ? kFltMax : dist - 0.02f - 1e-5f*dist;
}

template <class SC, class Stack, bool COUNT, bool ANYHIT = false>
TN_D int trace(const SC& sc, Stack& st, V3 o, V3 d, float time, float& outT, V3& outN, TraceCounters& ctr, float tStop = 0.0f)
{
    float minT = kFltMax;
    int closest = -1;
    V3 cn;

    V3 rcp = rcp3_cr(d);

    // detail counting (COUNT) always walks the BVH: the counters define the reference algorithm's
    // per-ray constants I, T, P of the algorithmic-bytes model
    if (!COUNT && sc.flatScan)
    {
        // "sane" rays only: the always-hit shortcut for infinite boxes assumes |origin| << 1e8
        const bool sane = ray_sane(o);
        bool tie = false;
        int prim = -1;
        // (the flat scan always runs to its end: stopping it early was measured and costs more in the scan's shape than the
        // skipped tests give back -- cornell 2835 -> 2713 Msamples/s, veach 1454 -> 1366; what stops early is the walks)
        if (sane)
            prim = trace_flat<SC, Stack, COUNT, false>(sc, st, o, d, rcp, time, outT, outN, tie, ctr, 0.0f);
        if (sane && !tie)
            return prim;
    }
    TN_TTICK0(ctr)

    int sp = 0;
    st.set(sp++, sc.root);

    while (sp)
    {
        uint32_t ref = st.get(--sp);

        if (ref & kLeafBit)
        {
            float t;
            V3 n;
            const int index = (int)(ref & ~kLeafBit);
            if (prim_intersect<SC, Stack, COUNT, ANYHIT>(sc, index, st, sp, o, d, time, t, n, ctr, tStop, rcp, true))
            {
                if (t < minT && t > 0.0f)
                {
                    minT = t;
                    closest = index;
                    cn = n;
                    if (ANYHIT && t < tStop)
                        break;          // decided (shadow_stop)
                }
            }
        }
        else
        {
            Node64 nd = load_node(sc.nodes, ref);
            if (COUNT) ctr.internal++;

            float tL, tR;
            bool hL = ray_aabb(o, rcp, nd.lminx, nd.lminy, nd.lminz, nd.lmaxx, nd.lmaxy, nd.lmaxz, tL);
            bool hR = ray_aabb(o, rcp, nd.rminx, nd.rminy, nd.rminz, nd.rmaxx, nd.rmaxy, nd.rmaxz, tR);

            uint32_t first = nd.left, second = nd.right;
            if (hL && hR && (tL < tR))
            {
                first = nd.right;
                second = nd.left;
            }
            if (hL)
                st.set(sp++, first);
            if (hR)
                st.set(sp++, second);
        }
    }

    outT = minT;
    outN = face_forward(cn, -d);
    TN_TTICK(ctr, 5)
    return closest;
}

} // namespace tn
