// tn_host_layout.h -- BVH re-layout (reference 32-B nodes -> Node64), leaf boxes, the scene arena and its device copy
// (part of the library's one host translation unit: included by tinsel_hip.hip, in this order, never on its own)
#pragma once

namespace {

// ---------------------------------------------------------------------------
// BVH re-layout: reference 32-B nodes -> one Node64 per internal node (tn_scene.h)

struct ConvertedBvh
{
    std::vector<Node64> nodes;
    uint32_t root = 0;
    int maxLeafDepth = 0;
    int topCount = 0;           // nodes [0, topCount) are numbered breadth-first
};

inline bool ref_is_leaf(const tinsel_bvh_node& n) { return (n.right_index_leaf >> 31) != 0; }
inline uint32_t ref_right(const tinsel_bvh_node& n) { return n.right_index_leaf & 0x7fffffffu; }

// `numItems`: what a leaf may index (primitives / triangles); `topBudget`: how many internal nodes to number
// breadth-first from the root (the part of a tree in HBM that k_walk stages into LDS, tn_walk.h) before the rest is
// numbered depth-first (a node's left subtree follows it immediately: the builder's locality).  Node numbers are labels
// only: boxes, children and the visit order of a traversal do not depend on them.
// Refuses malformed input: child / item indices out of range, a node reachable twice (a cycle or a DAG).
bool convert_bvh(const tinsel_bvh_node* ref, int numNodes, int numItems, int topBudget, ConvertedBvh& out)
{
    out.nodes.clear();
    out.maxLeafDepth = 0;
    out.topCount = 0;
    if (numNodes <= 0 || !ref)
        return false;

    if (ref_is_leaf(ref[0]))
    {
        if (ref[0].left_index >= (uint32_t)numItems)
            return false;
        out.root = kLeafBit | ref[0].left_index;
        return true;
    }

    std::vector<uint32_t> internalIndex((size_t)numNodes, kNoNode);
    struct Item { uint32_t node; int depth; };
    std::vector<uint32_t> order;
    order.reserve((size_t)numNodes/2 + 1);

    // takes one node off a work list: leaves only report their depth, internal nodes get the next number
    auto visit = [&](const Item& it, uint32_t& left, uint32_t& right) -> int {      // 0 leaf, 1 internal, -1 malformed
        const tinsel_bvh_node& n = ref[it.node];
        if (ref_is_leaf(n))
        {
            if (n.left_index >= (uint32_t)numItems)
                return -1;
            if (it.depth > out.maxLeafDepth)
                out.maxLeafDepth = it.depth;
            return 0;
        }
        if (n.left_index >= (uint32_t)numNodes || ref_right(n) >= (uint32_t)numNodes || internalIndex[it.node] != kNoNode)
            return -1;
        internalIndex[it.node] = (uint32_t)order.size();
        order.push_back(it.node);
        left = n.left_index;
        right = ref_right(n);
        return 1;
    };

    // breadth-first part
    std::vector<Item> frontier;
    frontier.push_back({ 0u, 0 });
    size_t head = 0;
    while (head < frontier.size() && (int)order.size() < topBudget)
    {
        const Item it = frontier[head++];
        uint32_t l = 0, r = 0;
        const int kind = visit(it, l, r);
        if (kind < 0)
            return false;
        if (kind == 1)
        {
            frontier.push_back({ l, it.depth + 1 });
            frontier.push_back({ r, it.depth + 1 });
        }
    }
    out.topCount = (int)order.size();

    // depth-first pre-order below the frontier
    std::vector<Item> stack;
    for (; head < frontier.size(); ++head)
    {
        stack.push_back(frontier[head]);
        while (!stack.empty())
        {
            const Item it = stack.back();
            stack.pop_back();
            uint32_t l = 0, r = 0;
            const int kind = visit(it, l, r);
            if (kind < 0)
                return false;
            if (kind == 1)
            {
                stack.push_back({ r, it.depth + 1 });
                stack.push_back({ l, it.depth + 1 });
            }
        }
    }

    out.nodes.resize(order.size());
    for (size_t k = 0; k < order.size(); ++k)
    {
        const tinsel_bvh_node& n = ref[order[k]];
        const tinsel_bvh_node& l = ref[n.left_index];
        const tinsel_bvh_node& r = ref[ref_right(n)];
        Node64& o = out.nodes[k];
        memset(&o, 0, sizeof(o));
        o.lminx = l.lower.x; o.lminy = l.lower.y; o.lminz = l.lower.z;
        o.lmaxx = l.upper.x; o.lmaxy = l.upper.y; o.lmaxz = l.upper.z;
        o.rminx = r.lower.x; o.rminy = r.lower.y; o.rminz = r.lower.z;
        o.rmaxx = r.upper.x; o.rmaxy = r.upper.y; o.rmaxz = r.upper.z;
        o.left = ref_is_leaf(l) ? (kLeafBit | l.left_index) : internalIndex[n.left_index];
        o.right = ref_is_leaf(r) ? (kLeafBit | r.left_index) : internalIndex[ref_right(n)];
    }
    out.root = 0;
    return true;
}

// ---------------------------------------------------------------------------
// material digestion: every material-only sub-expression, in the reference's own precision

// the MIS constants of a (light) primitive, divided here once with the reference's fp32 expressions (tn_scene.h Mat128); again whenever
// PrimitiveArea changes (refit, a moved mesh light)
void set_light_constants(Mat128& m)
{
    m.rcpArea = 1.0f/m.area;                                                // (1.0f/lightArea), render.cpp:182, 292
    m.rcpLightSamples = 1.0f/(float)m.lightSamples;                         // (1.0f/numSamples), :223
    const int N = (int)((float)m.lightSamples + 1.0f);                      // lightSamples + kBsdfSamples, :209, :296
    m.cbsdf = 1.0f/(float)N;                                                // kBsdfSamples/N
    m.clight = (float)m.lightSamples/(float)N;
}

void make_material(const tinsel_primitive& p, Mat128& m)
{
    const tinsel_material& a = p.material;
    memset(&m, 0, sizeof(m));
    m.emission[0] = a.emission.x; m.emission[1] = a.emission.y; m.emission[2] = a.emission.z;
    m.color[0] = a.color.x; m.color[1] = a.color.y; m.color[2] = a.color.z;
    m.absorption[0] = a.absorption.x; m.absorption[1] = a.absorption.y; m.absorption[2] = a.absorption.z;

    // Material::GetIndexOfRefraction (scene.h:72-78): sqrtf(0.08*specular) with a double product
    if (a.eta == 0.0f)
        m.ior = 2.0f/(1.0f - sqrtf((float)(0.08*(double)a.specular))) - 1.0f;
    else
        m.ior = a.eta;

    m.metallic = a.metallic;
    m.subsurface = a.subsurface;
    m.roughness = a.roughness;
    m.transmission = a.transmission;
    m.clearcoat = a.clearcoat;

    // disney.h:306-310
    const float c[3] = { a.color.x, a.color.y, a.color.z };
    const float Cdlum = (float)(.3*(double)c[0] + .6*(double)c[1] + .1*(double)c[2]);
    float Ctint[3] = { 1.0f, 1.0f, 1.0f };
    if (Cdlum > 0.0f)
    {
        const float rcp = (float)(1.0/(double)Cdlum);      // Cdlin/Cdlum == Cdlin*(1.0/Cdlum), maths.h:242
        for (int k = 0; k < 3; ++k)
            Ctint[k] = c[k]*rcp;
    }
    const float spec08 = (float)((double)a.specular*.08);   // `mat.specular*.08` is a double, narrowed by operator*(Real, Vec3)
    for (int k = 0; k < 3; ++k)
    {
        const float tint = 1.0f + (Ctint[k] - 1.0f)*a.specular_tint;   // Lerp(Vec3(1), Ctint, specularTint)
        const float s = tint*spec08;
        m.cspec0[k] = s + (c[k] - s)*a.metallic;                       // Lerp(., Cdlin, metallic)
        m.sqrtColor[k] = sqrtf(c[k]);                                  // disney.h:352
    }

    // Lerp(.1,.001, clearcoatGloss) evaluated in double (disney.h:387)
    m.clearcoatAlpha = (float)(.1 + (.001 - .1)*(double)a.clearcoat_gloss);
    m.clearcoatA2 = m.clearcoatAlpha*m.clearcoatAlpha;     // GTR1: a2 = a*a; logf(a2) by the host libm = the oracle's own
    m.clearcoatLogA2 = logf(m.clearcoatA2);

    // PrimitiveArea (intersection.h:833-853)
    if (p.type == TINSEL_GEOM_SPHERE)
        m.area = 4.0f*kPi*p.geo.sphere.radius*p.geo.sphere.radius;
    else if (p.type == TINSEL_GEOM_MESH)
        m.area = p.geo.mesh.area*p.end_transform.s;
    else
        m.area = 0.0f;

    m.lightSamples = p.light_samples;
    set_light_constants(m);
}

// the leaf box of a primitive as the flat scan reads it
PrimBox make_prim_box(const tinsel_bvh_node& nd)
{
    PrimBox b;
    memset(&b, 0, sizeof(b));
    b.minx = nd.lower.x; b.miny = nd.lower.y; b.minz = nd.lower.z;
    b.maxx = nd.upper.x; b.maxy = nd.upper.y; b.maxz = nd.upper.z;
    b.alwaysHit = (nd.lower.x <= -1e7f && nd.lower.y <= -1e7f && nd.lower.z <= -1e7f &&
                   nd.upper.x >= 1e7f && nd.upper.y >= 1e7f && nd.upper.z >= 1e7f) ? 1u : 0u;
    return b;
}

// TransformBounds (maths.h:1004-1021), in the reference's operation order
void transform_bounds(const Xform& x, V3 lower, V3 upper, V3& outLower, V3& outUpper)
{
    const V3 c0 = qrotate(x.r, V3(1.0f, 0.0f, 0.0f)), c1 = qrotate(x.r, V3(0.0f, 1.0f, 0.0f)), c2 = qrotate(x.r, V3(0.0f, 0.0f, 1.0f));    // Mat33(Quat), maths.h:654-663
    const V3 halfEdgeWidth = (x.s*(upper - lower))*0.5f;
    const V3 ax = V3(absf(c0.x), absf(c0.y), absf(c0.z))*halfEdgeWidth.x;
    const V3 ay = V3(absf(c1.x), absf(c1.y), absf(c1.z))*halfEdgeWidth.y;
    const V3 az = V3(absf(c2.x), absf(c2.y), absf(c2.z))*halfEdgeWidth.z;
    const V3 center = xform_point(x, 0.5f*(lower + upper));
    outLower = center - ax - ay - az;
    outUpper = center + ax + ay + az;
}

Moving64 make_moving(const Xform& xs, const Xform& xe)
{
    Moving64 mv;
    mv.spx = xs.p.x; mv.spy = xs.p.y; mv.spz = xs.p.z; mv.ss = xs.s;
    mv.srx = xs.r.x; mv.sry = xs.r.y; mv.srz = xs.r.z; mv.srw = xs.r.w;
    mv.epx = xe.p.x; mv.epy = xe.p.y; mv.epz = xe.p.z; mv.es = xe.s;
    mv.erx = xe.r.x; mv.ery = xe.r.y; mv.erz = xe.r.z; mv.erw = xe.r.w;
    return mv;
}

Xform to_xform(const tinsel_transform& t);

// the pose part of a primitive's record: static primitives carry InterpolateTransform(a, a, t), evaluated once with the same function
void set_prim_pose(Prim64& o, const Xform& xs, const Xform& xe, bool isStatic)
{
    if (isStatic)
    {
        const Xform x = interpolate_xform(xs, xe, 0.0f);
        o.px = x.p.x; o.py = x.p.y; o.pz = x.p.z; o.s = x.s;
        o.rx = x.r.x; o.ry = x.r.y; o.rz = x.r.z; o.rw = x.r.w;
        o.flags &= ~(uint32_t)kPrimMoving;
    }
    else
    {
        o.px = o.py = o.pz = o.s = 0.0f;
        o.rx = o.ry = o.rz = o.rw = 0.0f;
        o.flags |= kPrimMoving;
    }
}

// what the device derives from a STATIC pose once instead of per ray (call when o.type and the pose are both set): the reciprocal of a
// mesh's scale -- InverseTransformPoint / InverseTransformVector divide 1.0f by it per call (maths.h:611-619), the same IEEE division
// here -- and whether the rotation is the identity quaternion bit for bit (pose_rotate_*, tn_isect.h)
void set_prim_derived(Prim64& o)
{
    o.flags &= ~(uint32_t)kPrimNoRot;
    if (o.flags & kPrimMoving)
        return;
    uint32_t rb[4];
    const float rr[4] = { o.rx, o.ry, o.rz, o.rw };
    memcpy(rb, rr, sizeof(rb));
    if (rb[0] == 0u && rb[1] == 0u && rb[2] == 0u && o.rw == 1.0f)
        o.flags |= kPrimNoRot;
    if (o.type == kPrimMesh)
        o.g3 = 1.0f/o.s;
}

Xform to_xform(const tinsel_transform& t)
{
    Xform x;
    x.p = V3(t.p.x, t.p.y, t.p.z);
    x.r = { t.r.x, t.r.y, t.r.z, t.r.w };
    x.s = t.s;
    return x;
}

// ---------------------------------------------------------------------------

// Host-side image of DevScene::arena: 128-B aligned sections, uploaded as one allocation.
struct ArenaBuilder
{
    std::vector<unsigned char> bytes;
    template <class T>
    size_t add(const T* data, size_t count)
    {
        const size_t off = (bytes.size() + 127) & ~size_t(127);
        bytes.resize(off + sizeof(T)*count, 0);
        if (count)
            memcpy(&bytes[off], data, sizeof(T)*count);
        return off;
    }
};

constexpr size_t kSmallMeshBytes = 4096;        // meshes up to this size ride inside the arena
constexpr int kInlineMaxTris = 7;               // ... of a scene that has a mesh in HBM, only up to this many triangles
constexpr int kWalkTopNodes = 2048;             // internal nodes of a mesh in HBM numbered breadth-first (128 KB: more than LDS can take)
constexpr size_t kArenaLdsLimit = 32768;        // arenas up to this size are staged into LDS by the kernels

struct DeviceArena
{
    std::vector<void*> allocs;

    template <class T>
    T* upload(const T* host, size_t count)
    {
        if (count == 0)
            return nullptr;
        void* d = nullptr;
        if (hipMalloc(&d, sizeof(T)*count) != hipSuccess)
            return nullptr;
        allocs.push_back(d);
        if (hipMemcpy(d, host, sizeof(T)*count, hipMemcpyHostToDevice) != hipSuccess)
            return nullptr;
        return (T*)d;
    }

    void release()
    {
        for (void* p : allocs)
            (void)hipFree(p);
        allocs.clear();
    }
};

const char* kKernelNames[] = { "k_generate", "k_extend", "k_shade", "k_shadow", "k_accumulate", "k_mega", "k_normals", "k_bounce",
                               "k_present", "k_nlm_means", "k_nlm", "k_walk", "k_lights", "k_seg", "k_step" };
enum { KN_GENERATE = 0, KN_EXTEND, KN_SHADE, KN_SHADOW, KN_ACCUMULATE, KN_MEGA, KN_NORMALS, KN_BOUNCE, KN_PRESENT, KN_NLM_MEANS, KN_NLM, KN_WALK, KN_LIGHTS, KN_SEG, KN_STEP, KN_COUNT };

struct TimedSpan { int kernel; hipEvent_t start, stop; };

} // namespace
