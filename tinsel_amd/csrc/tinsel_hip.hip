// tinsel_hip.hip -- host side of the C-ABI (include/tinsel_hip.h): scene flattening/upload,
// camera set-up, batch scheduling of the streaming pipeline, statistics and timing.
//
// Replaces the reference's GpuRenderer (src/render.cu:978-1110).  Built only with
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off  (tinsel_amd/build.py)

#include "../../include/tinsel_hip.h"

#include "tn_launch.h"
#include "tn_lbvh.h"
#include "tn_ubench.h"
#include "tn_selftest.h"

#include "tn_sort.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types only: the library is dlopen'ed by the first multi-GPU group (no link-time dependency)

#include <dlfcn.h>

#include <cmath>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace tn;

namespace {

thread_local std::string g_error;

int fail(const std::string& msg)
{
    g_error = msg;
    fprintf(stderr, "tinsel_hip: %s\n", msg.c_str());
    return -1;
}

#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess)                                                                               \
            return fail(std::string(#expr) + ": " + hipGetErrorString(_e));                                 \
    } while (0)

#define HIP_TRY_NULL(expr)                                                                                  \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess) {                                                                             \
            fail(std::string(#expr) + ": " + hipGetErrorString(_e));                                        \
            return nullptr;                                                                                 \
        }                                                                                                   \
    } while (0)

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
                               "k_present", "k_nlm_means", "k_nlm", "k_walk", "k_lights", "k_seg" };
enum { KN_GENERATE = 0, KN_EXTEND, KN_SHADE, KN_SHADOW, KN_ACCUMULATE, KN_MEGA, KN_NORMALS, KN_BOUNCE, KN_PRESENT, KN_NLM_MEANS, KN_NLM, KN_WALK, KN_LIGHTS, KN_SEG, KN_COUNT };

struct TimedSpan { int kernel; hipEvent_t start, stop; };

} // namespace

struct tinsel_hip
{
    int device = 0;
    int numCUs = 256;

    DeviceArena sceneMem;
    DevScene scene;
    int stackNeed = 16;
    int neePerPath = 0;

    // mesh table as uploaded (reference trees) and as it currently is; device-built trees (tn_lbvh.h)
    std::vector<DevMesh> meshesRef, meshesNow;
    // what a refit needs on the host (tinsel_hip_refit_mesh): per mesh the vertex count and the index triples, per
    // primitive its mesh and the endTransform scale of PrimitiveArea
    std::vector<int> meshNumVertices;
    std::vector<std::vector<int32_t>> meshIndices;
    std::vector<int> primMesh;
    std::vector<float> primEndScale;
    // ... and what moving a PRIMITIVE needs (tinsel_hip_set_primitive_transform / tinsel_hip_rebuild_scene): the Prim64 records as
    // uploaded, where they and the Moving64 slots (one per primitive) sit in the arena, every mesh's root box in mesh space and its
    // area (PrimitiveBounds, PrimitiveArea), whether a transform changed since the scene BVH was last built
    std::vector<Prim64> primsHost;
    size_t arenaOffPrims = 0, arenaOffMoving = 0, arenaOffMats = 0;
    std::vector<V3> meshRootLo, meshRootHi;
    std::vector<float> meshArea;
    bool sceneDirty = false;
    // ... and to follow a refitted mesh at the SCENE level (its primitives' leaf boxes and their ancestors in the scene BVH):
    // the primitives' start / end transforms, the reference's scene BVH as handed in, where its device form and the leaf boxes
    // sit in the arena
    std::vector<Xform> primStart, primEnd;
    std::vector<tinsel_bvh_node> sceneBvhHost;
    size_t arenaOffNodes = 0, arenaOffBoxes = 0;
    std::vector<int32_t> planeTablePrims;       // the planes DevScene::planeEq holds (their PrimBox says 2: re-marked when the boxes are rewritten)
    int sceneStackNeed = 1;
    std::string prepRefused;            // non-empty: a kernel whose dynamic-LDS limit the runtime refused to raise (prepare_kernels_once)
    int bounceWaves = 3;                // k_bounce's waves per SIMD = its resident workgroups per CU for the batch being launched (plan_bounce)
    bool sceneEnclosed = false;         // two planes face each other: (practically) no ray leaves the scene (k_bounce's shading pools stay off)
    int bvhMode = TINSEL_BVH_REFERENCE;
    int rrStart = 0;                    // > 0: Russian roulette from this bounce on (opt-in)
    std::vector<void*> lbvhAllocs;

    int width = 0, height = 0;
    float4* accum = nullptr;
    bool accumOwned = true;

    // sharded renders: accumulate tiles that have candidate paths of this shard (k_accumulate_tiled)
    int* accTilesDev = nullptr;
    int accTilesCount = 0;
    int accTilesKey[6] = { 0, 0, 0, 0, 0, 0 };     // width, height, rank, world, shard tile, halo reach

    // display stage (tn_display.h): [0] filtered, [1] NLM means, [2] NLM output; sized width*height on first use
    float4* display[3] = { nullptr, nullptr, nullptr };
    size_t displayPixels = 0;
    const float4* presented = nullptr;

    // path batch buffers
    size_t batchSlots = 0;
    int batchNee = -1;
    int batchDepth = -1;
    std::vector<void*> batchAllocs;
    PathState ps;
    QueueCtl ctl;
    // the wavefront pipelines' dense state (SplitState, tn_kernels.h); the split pipeline's hit / shadow-ray arrays only when
    // that is the pipeline in force
    SplitState ss;
    int batchPipeline = -1;             // the pipeline the current batch buffers were allocated for
    size_t splitCap = 0;                // positions per SplitState array: the batch slots + one wave of padding per region
    uint32_t splitMaxRegions = 0;
    uint32_t* regionOrder = nullptr;    // region groups, longest first (k_region_order): by live paths, by shadow-ray bundles
    uint32_t* regionOrderNee = nullptr;
    uint32_t* walkList = nullptr;       // k_walk's work list (k_seg_expand) and the prefix of the regions' front counts behind it
    uint32_t* segPrefix = nullptr;
    BinPrims binPrims = { 0, { 0, 0, 0, 0, 0, 0, 0 } };
    BinPrims walkPrims = { 0, { 0, 0, 0, 0, 0, 0, 0 } };   // the subset of binPrims whose closest hits k_walk computes (large trees)
    int walkPrimMesh[7] = { 0, 0, 0, 0, 0, 0, 0 };         // DevScene::meshes index of each walked primitive
    float4* walkRec = nullptr;                          // k_walk's closest-hit records (tn_walk.h); batch-sized
    // A SECOND set of the dense state (render_impl's overlapped chunks: two halves of a batch on two streams, each chunk's accumulate
    // behind the other chunk's kernels).  The fields above are the set in use; lane_swap exchanges them with this one between ENQUEUES
    // (a launch has copied its pointers by the time it returns).
    struct DenseLane
    {
        SplitState ss;
        size_t splitCap = 0;
        uint32_t splitMaxRegions = 0;
        uint32_t *regionOrder = nullptr, *regionOrderNee = nullptr, *walkList = nullptr, *segPrefix = nullptr;
        float4* walkRec = nullptr;
        uint32_t* walkOverflow = nullptr;       // (allocated by launch_walk on first use; freed with the renderer, not with the batch)
        size_t walkOverflowCap = 0;
    } laneB;
    int batchLanes = 1;                 // dense-state sets allocated (1 or 2)
    size_t batchStateSlots = 0;         // path slots each set holds (batchSlots: what ps.rad holds)
    hipStream_t laneStream = nullptr;   // the second chunk's stream
    hipEvent_t laneFork = nullptr, laneJoin = nullptr, accDone[2] = { nullptr, nullptr };
    uint32_t* walkOverflow = nullptr;                   // k_walk's stack entries beyond the LDS ones (TINSEL_HIP_WALK_LDS_STACK)
    size_t walkOverflowCap = 0;
    bool walkEnabled = true;                            // TINSEL_HIP_NO_WALK: walk meshes inline in k_extend / k_shadow (A/B)
    unsigned long long* walkProf = nullptr;             // developer-only (-DTN_WALK_PROF builds): section counters of k_walk
    uint2* probeAlias = nullptr;                        // alias table of the probe (tinsel_hip_set_probe_sampling), built on first use
    int sharedMemLimit = 65536;
    uint32_t* passSeedsDev = nullptr;   // the table: the seeds of passes [passSeedsBase, passSeedsBase + passSeedsCount)
    size_t passSeedsCap = 0, passSeedsCount = 0;
    uint32_t passSeedsBase = 0;
    const uint32_t* passSeeds = nullptr;    // the current call's first seed, inside the table
    hipEvent_t passSeedsReady = nullptr;    // recorded behind the launch that wrote the table, on passSeedsStream
    hipStream_t passSeedsStream = nullptr;
    unsigned long long* statsDev = nullptr;

    size_t lastBatchSlots = 0;
    int lastPipeline = TINSEL_PIPELINE_WAVEFRONT;   // of the last batch (queue_counts)
    uint32_t lastRegions = 0;
    size_t maxBatchSlots = 8u << 20;
    bool batchSlotsExplicit = false;     // set by TINSEL_HIP_BATCH_PATHS / tinsel_hip_set_batch_paths
    int pipeline = TINSEL_PIPELINE_AUTO;
    int arith = TINSEL_ARITH_EXACT;     // which build of the path kernels runs (tinsel_hip_set_arithmetic)
    bool pathKernelsPrepared = false;
    int segPrefixLds = 0;               // dynamic LDS k_seg_prefix may ask for (prepare_path_kernels): one count per region
    bool countDetail = false;

    uint32_t passIndex = 0;
    Rng seedRng = Rng::seeded(1u);      // Random(1) advanced seedRngIndex times: the generator of the pass seeds
    uint32_t seedRngIndex = 0;
    int shardRank = 0, shardWorld = 1, shardTile = 32;

    // look-ahead (tinsel_hip_set_lookahead): the NEXT call's passes are traced speculatively into accumSpec while this
    // call's running sum travels to the host
    int lookahead = 0;                  // 0 off, 1 on, 2 on + the caller's output array page-locked in place (TINSEL_LOOKAHEAD_PIN_OUTPUT)
    FrameParams lastFp;                 // of the most recent batch (its paths' radiance is still in ps.rad)
    struct SpecShot { float4* buf; hipEvent_t ready; };
    std::vector<float4*> specFree;      // accumulator-sized buffers not in use
    std::deque<SpecShot> specQueue;     // specQueue[j] = accum + the passes of the next j+1 calls, in flight or finished on workStream
    uint32_t specNextPass = 0;          // pass index the next speculated call starts at
    tinsel_camera specCamera;
    tinsel_options specOptions;
    int specPasses = 0;
    int lookaheadDepth = 0;             // calls per speculated batch; 0 = chosen from the batch capacity 
    hipStream_t workStream = nullptr, copyStream = nullptr;
    void* pinnedPtr = nullptr;          // caller's output buffer, page-locked in place (hipHostRegister) for the D2H DMA
    size_t pinnedBytes = 0;

    bool timing = false;
    std::vector<TimedSpan> spans;
    std::vector<hipEvent_t> eventPool;
    double gpuSeconds = 0.0;
};

namespace {

void free_batch(tinsel_hip* r)
{
    for (void* p : r->batchAllocs)
        (void)hipFree(p);
    r->batchAllocs.clear();
    r->walkRec = nullptr;
    {
        tinsel_hip::DenseLane fresh;
        fresh.walkOverflow = r->laneB.walkOverflow;
        fresh.walkOverflowCap = r->laneB.walkOverflowCap;
        r->laneB = fresh;
    }
    r->batchPipeline = -1;
    r->batchSlots = 0;
    r->batchStateSlots = 0;
    r->batchLanes = 1;
    r->batchNee = -1;
    r->batchDepth = -1;
}

template <class T>
int batch_alloc(tinsel_hip* r, T** out, size_t count)
{
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, sizeof(T)*(count ? count : 1)));
    r->batchAllocs.push_back(d);
    *out = (T*)d;
    return 0;
}

// blocks per CU of the streaming kernels' fixed grid.  Swept 4..256 on every config: 32 is best everywhere (finer static
// ranges even out the tail; beyond 64 the per-block staging and the shorter ranges cost more than they give)
int grid_mult()
{
    static const int m = getenv("TINSEL_HIP_GRID_MULT") ? std::max(1, atoi(getenv("TINSEL_HIP_GRID_MULT"))) : 32;
    return m;
}

int resolve_pipeline(const tinsel_hip* r)
{
    if (r->pipeline != TINSEL_PIPELINE_AUTO)
        return r->pipeline;
    // A scene whose arena is staged whole into LDS runs the fused kernel, whatever its shadow rays per bounce (fused ->
    // split, Msamples/s: cornell 2550 -> 2055, gloss 6250 -> 4070, env_loft 3540 -> 2260, 4 rays: veach 1295 -> 1267, 9 rays:
    // features 690 -> 616, 10 rays: features + probe 589 -> 515; until the BSDF terms moved behind the shadow traces the
    // many-ray scenes were faster split); scenes with meshes or a scene BVH in HBM run the split pipeline.
    return r->scene.allInArena ? TINSEL_PIPELINE_WAVEFRONT : TINSEL_PIPELINE_WAVEFRONT_SPLIT;
}

// The wavefront pipelines' state (SplitState, tn_kernels.h): by POSITION, two buffers of everything a bounce rewrites; for the
// split pipeline also what its kernels hand to each other (hit, shadow rays and their results, k_walk's records and list)
int alloc_dense(tinsel_hip* r, size_t slots, int maxDepth, bool split)
{
    const size_t K = split ? (size_t)r->neePerPath : 0;
    // (half as many again as the widest grid: the short regions at the end of a batch, split_tail_regions)
    const size_t maxRegions = (size_t)r->numCUs*(size_t)grid_mult()*(kBlock/kWave)*3/2;
    const size_t cap = slots + maxRegions*kWave;        // a region is a whole number of waves long
    SplitState& ss = r->ss;
    memset(&ss, 0, sizeof(ss));
    for (int b = 0; b < 2; ++b)
        if (batch_alloc(r, &ss.rayO[b], cap) || batch_alloc(r, &ss.rayD[b], cap) || batch_alloc(r, &ss.thr[b], cap) ||
            batch_alloc(r, &ss.rad[b], cap) || batch_alloc(r, &ss.rngId[b], cap))
            return -1;
    if (batch_alloc(r, &ss.segFront, maxRegions*((size_t)maxDepth + 1)) || batch_alloc(r, &ss.segBack, maxRegions*((size_t)maxDepth + 1)) ||
        batch_alloc(r, &r->regionOrder, maxRegions/(kBlock/kWave)) || batch_alloc(r, &r->regionOrderNee, maxRegions/(kBlock/kWave)))
        return -1;
    ss.radOut = r->ps.rad;
    ss.capacity = (uint32_t)cap;
    r->splitCap = cap;
    r->splitMaxRegions = (uint32_t)maxRegions;
    r->walkRec = nullptr;
    r->walkList = nullptr;
    r->segPrefix = nullptr;
    if (!split)
        return 0;

    if (batch_alloc(r, &ss.hit, cap) || batch_alloc(r, &ss.hitPrim, cap) || batch_alloc(r, &ss.pathNee, K ? cap : 1) ||
        batch_alloc(r, &ss.neeRay, cap*K*2) || batch_alloc(r, &ss.neeSky, r->scene.probe.valid ? cap : 1) ||
        batch_alloc(r, &ss.neeTime, K ? cap : 1) || batch_alloc(r, &ss.neeRes, cap*K) ||
        batch_alloc(r, &ss.neeFront, maxRegions*(size_t)maxDepth) || batch_alloc(r, &ss.neeBack, maxRegions*(size_t)maxDepth))
        return -1;
    ss.neePerPath = (int32_t)K;
    // k_walk: one 32-B closest hit per (ray, walked primitive), by position; extension and shadow rays share the buffer
    if (r->walkPrims.count > 0 && r->walkEnabled && (double)cap*(K > 1 ? K : 1)*r->walkPrims.count < 2147483648.0)
        if (batch_alloc(r, &r->walkRec, cap*(size_t)(K > 1 ? K : 1)*(size_t)r->walkPrims.count*2) || batch_alloc(r, &r->walkList, cap) ||
            batch_alloc(r, &r->segPrefix, maxRegions + 1))
            return -1;
    // k_swalk's list (scenes the flat scan cannot take): the same two arrays (such scenes have no walked primitives)
    if (!r->walkList && !r->scene.flatScan)
        if (batch_alloc(r, &r->walkList, cap) || batch_alloc(r, &r->segPrefix, maxRegions + 1))
            return -1;
    return 0;
}

void lane_swap(tinsel_hip* r)
{
    tinsel_hip::DenseLane& b = r->laneB;
    std::swap(r->ss, b.ss);
    std::swap(r->splitCap, b.splitCap);
    std::swap(r->splitMaxRegions, b.splitMaxRegions);
    std::swap(r->regionOrder, b.regionOrder);
    std::swap(r->regionOrderNee, b.regionOrderNee);
    std::swap(r->walkList, b.walkList);
    std::swap(r->segPrefix, b.segPrefix);
    std::swap(r->walkRec, b.walkRec);
    std::swap(r->walkOverflow, b.walkOverflow);
    std::swap(r->walkOverflowCap, b.walkOverflowCap);
}

// slots: paths whose radiance ps.rad holds (a whole batch); stateSlots: paths each set of dense state holds (a chunk of the batch
// where chunks overlap, render_impl; 0: the whole batch); lanes: how many sets
int ensure_batch(tinsel_hip* r, size_t slots, int maxDepth, size_t stateSlots = 0, int lanes = 1)
{
    const int K = r->neePerPath;
    const int pipeline = resolve_pipeline(r);
    if (stateSlots == 0 || stateSlots > slots)
        stateSlots = slots;
    if (r->batchSlots >= slots && r->batchStateSlots >= stateSlots && r->batchLanes >= lanes && r->batchNee == K && r->batchDepth >= maxDepth &&
        r->batchPipeline == pipeline)
        return 0;
    free_batch(r);

    // the radiance of finished paths by slot is what every pipeline hands to the accumulate kernels
    PathState& ps = r->ps;
    memset(&ps, 0, sizeof(ps));
    if (batch_alloc(r, &ps.rad, slots))
        return -1;
    if (pipeline != TINSEL_PIPELINE_MEGAKERNEL)
        for (int lane = lanes; lane-- > 0; )
        {
            if (alloc_dense(r, stateSlots, maxDepth, pipeline == TINSEL_PIPELINE_WAVEFRONT_SPLIT))
                return -1;
            if (lane > 0)
                lane_swap(r);           // the set just made becomes laneB
        }

    // slots of other shards are never written (gen_slot): keep their radiance at zero for the test hook
    HIP_TRY(hipMemset(ps.rad, 0, sizeof(float4)*slots));
    // (a memset of device memory only enqueues on the null stream, and the kernels that follow may run on a NON-BLOCKING stream
    // -- a group member's, the look-ahead's -- which the null stream does not order: without this wait the zeroes could land on
    // radiance a kernel had already written.  Allocation path only.)
    HIP_TRY(hipStreamSynchronize(nullptr));

    r->ctl.stats = r->statsDev;
    r->batchSlots = slots;
    r->batchStateSlots = stateSlots;
    r->batchLanes = pipeline != TINSEL_PIPELINE_MEGAKERNEL ? lanes : 1;
    r->batchNee = K;
    r->batchDepth = maxDepth;
    r->batchPipeline = pipeline;
    return 0;
}

// CameraSampler constructor (util.h:45-71) + Mat44(Transform) (maths.h:841-849), host side, once per call
void make_camera(const tinsel_camera& c, int width, int height, CameraParams& out)
{
    // Mat33(Quat): columns are q*e_k (maths.h:654-663); Mat44(Transform): cols*s, translation p*s with s == 1
    Q4 q = { c.rotation.x, c.rotation.y, c.rotation.z, c.rotation.w };
    const float s = 1.0f;
    V3 c0 = qrotate(q, V3(1.0f, 0.0f, 0.0f))*s;
    V3 c1 = qrotate(q, V3(0.0f, 1.0f, 0.0f))*s;
    V3 c2 = qrotate(q, V3(0.0f, 0.0f, 1.0f))*s;
    V3 c3 = V3(c.position.x, c.position.y, c.position.z)*s;

    // column-major 4x4s
    float c2w[16] = { c0.x, c0.y, c0.z, 0.0f, c1.x, c1.y, c1.z, 0.0f, c2.x, c2.y, c2.z, 0.0f, c3.x, c3.y, c3.z, 1.0f };

    // rasterToScreen given row-wise in the reference constructor (maths.h:801-829)
    float r2s[16] = { 2.0f/width, 0.0f, 0.0f, 0.0f,
                      0.0f, -2.0f/height, 0.0f, 0.0f,
                      0.0f, 0.0f, 1.0f, 0.0f,
                      -1.0f, 1.0f, 1.0f, 1.0f };

    float f = tanf(c.fov*0.5f);
    float aspect = float(width)/height;

    float s2c[16] = { f*aspect, 0.0f, 0.0f, 0.0f,
                      0.0f, f, 0.0f, 0.0f,
                      0.0f, 0.0f, -1.0f, 0.0f,
                      0.0f, 0.0f, 0.0f, 1.0f };

    // MatrixMultiply<4,4,4> (maths.h:83-99): result[i+j*4] = sum_k a[i+k*4]*b[k+j*4], k ascending from t = 0
    auto mul = [](float* result, const float* a, const float* b) {
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
            {
                float t = 0.0f;
                for (int k = 0; k < 4; ++k)
                    t += a[i + k*4]*b[k + j*4];
                result[i + j*4] = t;
            }
    };

    float tmp[16];
    mul(tmp, c2w, s2c);             // cameraToWorld*screenToCamera
    mul(out.r2w, tmp, r2s);         // ... *rasterToScreen
    out.ox = c2w[12]; out.oy = c2w[13]; out.oz = c2w[14];
    out.shutterStart = c.shutter_start;
    out.shutterEnd = c.shutter_end;
}

hipEvent_t get_event(tinsel_hip* r)
{
    if (!r->eventPool.empty())
    {
        hipEvent_t e = r->eventPool.back();
        r->eventPool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

struct ScopedTimer
{
    tinsel_hip* r;
    hipStream_t stream;
    TimedSpan span;
    bool on;
    ScopedTimer(tinsel_hip* r_, int kernel, hipStream_t s) : r(r_), stream(s), on(r_->timing)
    {
        if (on)
        {
            span.kernel = kernel;
            span.start = get_event(r);
            span.stop = get_event(r);
            (void)hipEventRecord(span.start, stream);
        }
    }
    ~ScopedTimer()
    {
        if (on)
        {
            (void)hipEventRecord(span.stop, stream);
            r->spans.push_back(span);
        }
    }
};

int pick_stack(int need)
{
    const int sizes[] = { 8, 12, 16, 24, 32, 48, 64, 96, 128, 156 };
    for (int s : sizes)
        if (need <= s)
            return s;
    return -1;
}

size_t stack_bytes(const tinsel_hip* r) { return ((size_t)r->stackNeed*kBlock + kScanWords)*sizeof(uint32_t) + r->scene.arenaLdsBytes; }

// The path kernels exist twice (tn_launch.h): this translation unit's, bit-identical to the CPU oracle, and
// tinsel_fast.hip's, built under the tolerance contract.  tinsel_hip_set_arithmetic picks the arm.
extern "C" void tinsel_fast_launch_path_kernel(int which, const void* launchArgs, void* stream);
extern "C" int tinsel_fast_prepare_path_kernels(int sharedMemLimit, const char** first);
extern "C" unsigned tinsel_fast_launch_args_size(void);

// Raises the dynamic-LDS limit of every kernel that needs more than the default launch limit, for both arithmetic arms (tn_launch.h).  Called by
// tinsel_hip_create, which refuses the device when the runtime refuses a kernel (r->prepRefused: "kernel name (arm)").
void prepare_kernels_once(tinsel_hip* r)
{
    if (r->pathKernelsPrepared)
        return;
    const PrepReport rep = prepare_path_kernels(r->sharedMemLimit);
    r->segPrefixLds = rep.segPrefixLds;
    const char* fastFirst = nullptr;
    const int fastRefused = tinsel_fast_prepare_path_kernels(r->sharedMemLimit, &fastFirst);
    if (rep.refused)
        r->prepRefused = std::string(rep.first ? rep.first : "?") + " (parity arm; " + std::to_string(rep.refused + fastRefused) + " kernels in all)";
    else if (fastRefused)
        r->prepRefused = std::string(fastFirst ? fastFirst : "?") + " (tolerance arm; " + std::to_string(fastRefused) + " kernels in all)";
    r->pathKernelsPrepared = true;
}

void launch_path(tinsel_hip* r, int which, const LaunchArgs& a, hipStream_t st)
{
    prepare_kernels_once(r);
    if (r->arith == TINSEL_ARITH_FAST)
        tinsel_fast_launch_path_kernel(which, &a, st);
    else
        launch_path_kernel(which, a, st);
}

// k_walk's records are used by the scan kernels unless the detail counters are on (those count the inline walk)
const float4* walk_records(const tinsel_hip* r) { return r->countDetail ? nullptr : r->walkRec; }

// what every launch of a batch shares
LaunchArgs batch_args(tinsel_hip* r, const CameraParams& cam, const FrameParams& fp)
{
    LaunchArgs a;
    memset(&a, 0, sizeof(a));
    a.scene = r->scene;
    a.ps = r->ps;
    a.ctl = r->ctl;
    a.cam = cam;
    a.fp = fp;
    a.passSeeds = r->passSeeds;
    a.walkRec = walk_records(r);
    a.walkPrims = (uint32_t)r->walkPrims.count;
    a.bins = r->binPrims;
    a.stackEntries = r->stackNeed;
    a.countDetail = r->countDetail ? 1 : 0;
    a.ldsBytes = (uint32_t)stack_bytes(r);
    return a;
}

// k_seg_prefix stages one count per region in dynamic LDS beside 64 B of static: what a launch may ask for (prepare_path_kernels raises
// the kernel's limit to the device's sharedMemLimit - 1024)
uint32_t seg_prefix_max_regions(tinsel_hip* r)
{
    prepare_kernels_once(r);
    return (uint32_t)std::max(0, r->segPrefixLds/4);
}

// k_walk (tn_walk.h): closest hits of the front rays of `queue` against the large meshes in HBM, ahead of the scan kernel.
// One 1024-thread workgroup per CU whose LDS holds the traversal stacks and, in what is left of the 160 KB, the top of
// the walked trees; trees too deep for that (a device-built LBVH of 524k triangles: 48 entries per lane) run 256-thread
// workgroups without a staged top.
int launch_walk(tinsel_hip* r, hipStream_t st, LaunchArgs a, const uint32_t* regionCounts, bool shadowRays)
{
    // (measured and settled, profiles/EXPERIMENTS.md: one resident set of workgroups; a refill once 24 lanes idle; a triangle phase once 8 wait)
    const int gridMult = 1, refillMin = 24, leafMin = 8;
    static const int forceBlock = getenv("TINSEL_HIP_WALK_BLOCK") ? atoi(getenv("TINSEL_HIP_WALK_BLOCK")) : 0;
    prepare_kernels_once(r);
    // the work list: the front entries of every region (paths / shadow-ray bundles whose ray enters a walked mesh's box)
    const SplitState& ss = a.ss;
    // the list visits the regions a golden-section step apart (TINSEL_HIP_WALK_LIST_STEP=1: in order)
    uint32_t step = (uint32_t)(ss.numRegions*0.6180339887) | 1u;
    {
        auto gcd = [](uint32_t x, uint32_t y) { while (y) { const uint32_t t = x % y; x = y; y = t; } return x; };
        while (step > 1 && gcd(step, ss.numRegions) != 1)
            step -= 1;
        if (step >= ss.numRegions || ss.numRegions > 65535u)       // k_seg_prefix multiplies in 32 bits
            step = 1;
    }
    {
        ScopedTimer t(r, KN_SEG, st);
        if (ss.numRegions > seg_prefix_max_regions(r))
            return fail("k_seg_prefix: " + std::to_string(ss.numRegions) + " regions do not fit its LDS (TINSEL_HIP_GRID_MULT too large for this device)");
        hipLaunchKernelGGL(k_seg_prefix, dim3(1), dim3(kSegBlock), ss.numRegions*sizeof(uint32_t), st, regionCounts, (const uint32_t*)nullptr, ss.numRegions, step, r->segPrefix);
        hipLaunchKernelGGL(k_seg_expand, dim3((unsigned)std::max(1, a.grid)), dim3(kBlock), 0, st, regionCounts, (const uint32_t*)r->segPrefix, ss, r->walkList);
    }
    WalkJob& job = a.walk;
    job.queue = r->walkList;
    job.frontCount = r->segPrefix + ss.numRegions;
    job.rayO = ss.rayO[a.bounce & 1];
    job.rayD = ss.rayD[a.bounce & 1];
    job.nee = ss.neeRay;
    job.neeStride = ss.capacity;
    job.neeTime = ss.neeTime;
    job.rec = r->walkRec;
    job.neePerPath = shadowRays ? r->neePerPath : 0;
    job.numPrims = r->walkPrims.count;
    int entries = 1;
    for (int k = 0; k < kWalkMaxPrims; ++k)
    {
        job.prim[k] = k < r->walkPrims.count ? r->walkPrims.prim[k] : 0;
        job.topCount[k] = 0;
        job.triCount[k] = 0;
        if (k < r->walkPrims.count)
            entries = std::max(entries, r->meshesNow[(size_t)r->walkPrimMesh[k]].stackNeed);
    }
    job.stackEntries = entries;
    job.prof = r->walkProf;
    job.refillMin = std::min(64, std::max(1, refillMin));
    job.leafMin = std::min(64, std::max(1, leafMin));
#ifdef TN_TUNE_ENV
    // (developer builds only -- scratch/build_variant.sh NAME -DTN_TUNE_ENV: the two thresholds from the environment for a sweep)
    if (getenv("TN_TUNE_WALK_REFILL")) job.refillMin = std::min(64, std::max(1, atoi(getenv("TN_TUNE_WALK_REFILL"))));
    if (getenv("TN_TUNE_WALK_LEAFMIN")) job.leafMin = std::min(64, std::max(1, atoi(getenv("TN_TUNE_WALK_LEAFMIN"))));
#endif
    // ONE walked primitive: its tree as kernel-argument scalars (TINSEL_HIP_WALK_SINGLE=0: per-lane pointers as for several; tests)
    {
        const char* singleEnv = getenv("TINSEL_HIP_WALK_SINGLE");
        a.walkSingle = (r->walkPrims.count == 1 && !(singleEnv && atoi(singleEnv) == 0)) ? 1 : 0;
    }

    const size_t ctl = kWalkCtlWords*sizeof(uint32_t);
    // n stack entries per lane in LDS (TINSEL_HIP_WALK_LDS_STACK, default 8; 0: the deepest tree's need, one workgroup per CU), the
    // rest of the deepest tree's need in HBM, and TWO 1024-thread workgroups per CU (8 waves per SIMD at 64 VGPRs) sharing the CU's
    // LDS: the 524k-triangle config's k_walk 19.0 -> 16.6 ms per 32 passes (2042 -> 2199 Msamples/s; 6 entries 17.2, 12 entries 16.7),
    // glass 10.4 -> 9.9; results unchanged (a stack entry is a stack entry wherever it lives)
    static const int ldsStackEnv = getenv("TINSEL_HIP_WALK_LDS_STACK") ? atoi(getenv("TINSEL_HIP_WALK_LDS_STACK")) : 8;
    // THE WHOLE MESH IN LDS (k_walk's kWalkLdsTris, tn_walk.h) where every walked tree is numbered breadth-first to its last node and all of
    // them, with their triangles' vertices (36 B each), fit beside the stacks of ONE 1024-thread workgroup per CU with at least four stack
    // entries per lane in LDS: glass.tin's sphere + cube (1290 nodes, 1292 triangles: 126 KB).  TINSEL_HIP_WALK_LDS_MESH=0: off (A/B, tests).
    int ldsMeshEntries = 0;
    {
        const char* meshEnv = getenv("TINSEL_HIP_WALK_LDS_MESH");
        size_t bytes = ctl;
        bool whole = !(meshEnv && atoi(meshEnv) == 0) && !forceBlock && r->walkPrims.count > 0;
        for (int k = 0; k < r->walkPrims.count && whole; ++k)
        {
            const DevMesh& dm = r->meshesNow[(size_t)r->walkPrimMesh[k]];
            whole = dm.topCount == dm.numInternal && dm.numInternal > 0;
            bytes += (size_t)dm.numInternal*sizeof(Node64) + (size_t)dm.numTris*36u;
        }
        if (whole)
            for (int e = std::min(entries, 8); e >= std::min(entries, 4) && !ldsMeshEntries; --e)
                if (bytes + (size_t)(e + kWalkLaneRows)*1024*sizeof(uint32_t) <= (size_t)r->sharedMemLimit)
                    ldsMeshEntries = e;
    }
    const bool twoPerCU = ldsStackEnv > 0 && !forceBlock && !ldsMeshEntries;
    const int ldsEntries = ldsMeshEntries ? ldsMeshEntries : twoPerCU ? std::min(entries, std::max(1, ldsStackEnv)) : entries;
    job.stackEntries = ldsEntries;
    job.overflow = nullptr;
    job.overflowEntries = 0;
    const size_t stackBig = (size_t)(ldsEntries + kWalkLaneRows)*1024*sizeof(uint32_t);     // (+ the per-lane rows, tn_walk.h)
    const size_t ldsBudget = twoPerCU ? (size_t)r->sharedMemLimit/2 : (size_t)r->sharedMemLimit;
    const bool big = forceBlock ? forceBlock == 1024 : stackBig + ctl + 16384 <= ldsBudget;
    const int block = big ? 1024 : 256;
    size_t lds = (size_t)(ldsEntries + kWalkLaneRows)*block*sizeof(uint32_t) + ctl;
    if (big)
    {
        // what is left of the CU's LDS goes to the tree tops, in primitive order
        size_t room = (ldsBudget - lds)/sizeof(Node64);
        for (int k = 0; k < r->walkPrims.count && room > 0; ++k)
        {
            const int n = (int)std::min<size_t>(room, (size_t)r->meshesNow[(size_t)r->walkPrimMesh[k]].topCount);
            job.topCount[k] = n;
            room -= (size_t)n;
            lds += (size_t)n*sizeof(Node64);
        }
        if (ldsMeshEntries)
            for (int k = 0; k < r->walkPrims.count; ++k)
            {
                job.triCount[k] = r->meshesNow[(size_t)r->walkPrimMesh[k]].numTris;
                lds += (size_t)job.triCount[k]*36u;
            }
    }
    a.walkLdsMesh = (big && ldsMeshEntries) ? 1 : 0;
    const size_t items = r->lastBatchSlots*(size_t)(shadowRays && r->neePerPath > 1 ? r->neePerPath : 1)*(size_t)r->walkPrims.count;
    const int perCU = big ? gridMult*(twoPerCU ? 2 : 1) : gridMult*4;
    a.grid = (int)std::max<size_t>(1, std::min<size_t>((items + block - 1)/block, (size_t)r->numCUs*(size_t)perCU));
    a.walkBig = big ? (twoPerCU ? 2 : 1) : 0;
    a.ldsBytes = (uint32_t)lds;
    if (ldsEntries < entries)
    {
        // the overflow columns are sized for the WIDEST grid this function launches (numCUs x perCU workgroups), once: nothing is freed or
        // allocated between the launches of a batch
        const size_t need = (size_t)r->numCUs*(size_t)perCU*(size_t)block*(size_t)(entries - ldsEntries);
        if (r->walkOverflowCap < need)
        {
            if (r->walkOverflow)
            {
                (void)hipDeviceSynchronize();       // (another stream's launch may still use the old columns)
                (void)hipFree(r->walkOverflow);
            }
            r->walkOverflow = nullptr;
            r->walkOverflowCap = 0;
            if (hipMalloc((void**)&r->walkOverflow, need*sizeof(uint32_t)) == hipSuccess)
                r->walkOverflowCap = need;
        }
        job.overflow = r->walkOverflow;
        job.overflowEntries = entries - ldsEntries;
        if (!job.overflow)
            return fail("k_walk: no memory for the stack overflow");
    }
    ScopedTimer t(r, KN_WALK, st);
    launch_path(r, PK_WALK, a, st);
    return 0;
}

// k_swalk (tn_swalk.h): the scene-level walk with ray replacement, for scenes the flat scan cannot take.  The list: every live
// entry of every region (front and back), regions in index order -- the workgroups' static ranges are image patches, coherent rays.
int launch_swalk(tinsel_hip* r, hipStream_t st, LaunchArgs a, const uint32_t* front, const uint32_t* back, bool shadowRays)
{
    const int refillMin = 32, leafMin = 16;         // (settled: profiles/r03_d_ab_swalk.txt, r03_e_ab_swalk.txt)
    static const bool noLds = getenv("TINSEL_HIP_SWALK_NO_LDS") != nullptr;
    const SplitState& ss = a.ss;
    // the list visits the regions a golden-section step apart: every workgroup's static range gets the same mix of rays
    // (k_walk's lesson; in index order a 256-thread grid of 8 workgroups per CU took 14.4 ms where 32 per CU took 9.2)
    uint32_t step = (uint32_t)(ss.numRegions*0.6180339887) | 1u;
    {
        auto gcd = [](uint32_t x, uint32_t y) { while (y) { const uint32_t t = x % y; x = y; y = t; } return x; };
        while (step > 1 && gcd(step, ss.numRegions) != 1)
            step -= 1;
        if (step >= ss.numRegions || ss.numRegions > 65535u)
            step = 1;
    }
    {
        ScopedTimer t(r, KN_SEG, st);
        if (ss.numRegions > seg_prefix_max_regions(r))
            return fail("k_seg_prefix: " + std::to_string(ss.numRegions) + " regions do not fit its LDS");
        hipLaunchKernelGGL(k_seg_prefix, dim3(1), dim3(kSegBlock), ss.numRegions*sizeof(uint32_t), st, front, back, ss.numRegions, step, r->segPrefix);
        hipLaunchKernelGGL(k_seg_expand_all, dim3((unsigned)std::max(1, a.grid)), dim3(kBlock), 0, st, front, back, (const uint32_t*)r->segPrefix, ss, r->walkList);
    }
    SwalkJob& job = a.swalk;
    job.list = r->walkList;
    job.count = r->segPrefix + ss.numRegions;
    job.neePerPath = shadowRays ? r->neePerPath : 0;
    job.stackEntries = r->stackNeed;
    job.refillMin = std::min(64, std::max(1, refillMin));
    job.leafMin = std::min(64, std::max(1, leafMin));
    // the whole arena beside the stacks of a 1024-thread workgroup?
    const size_t bigLds = ((size_t)r->stackNeed*1024 + kSwalkCtlWords)*sizeof(uint32_t) + r->scene.arenaBytes;
    const bool big = !noLds && bigLds <= (size_t)r->sharedMemLimit;
    bool allInArena = true;
    for (const DevMesh& dm : r->meshesNow)
        allInArena = allInArena && dm.inArena;
    a.swalkMode = big ? (allInArena ? 1 : 2) : 0;
    const int block = big ? 1024 : kBlock;
    const int gridMult = big ? 1 : 32;
    const size_t items = r->lastBatchSlots*(size_t)(shadowRays && r->neePerPath > 1 ? r->neePerPath : 1);
    a.grid = (int)std::max<size_t>(1, std::min<size_t>((items + block - 1)/block, (size_t)r->numCUs*(size_t)gridMult));
    if (big)
    {
        a.scene.arenaLdsBytes = a.scene.arenaBytes;
        a.ldsBytes = (uint32_t)bigLds;
    }
    else
        a.ldsBytes = (uint32_t)(((size_t)r->stackNeed*kBlock + kSwalkCtlWords)*sizeof(uint32_t) + r->scene.arenaLdsBytes);
    ScopedTimer t(r, shadowRays ? KN_SHADOW : KN_EXTEND, st);
    launch_path(r, shadowRays ? PK_SWALK_SHADOW : PK_SWALK_EXTEND, a, st);
    return 0;
}

void launch_normals(tinsel_hip* r, hipStream_t st, int grid, const CameraParams& cam, const FrameParams& fp)
{
    if (r->scene.allInArena)
        hipLaunchKernelGGL((k_normals<true>), dim3(grid), dim3(kBlock), stack_bytes(r), st, r->scene, cam, fp, r->accum, r->stackNeed);
    else
        hipLaunchKernelGGL((k_normals<false>), dim3(grid), dim3(kBlock), stack_bytes(r), st, r->scene, cam, fp, r->accum, r->stackNeed);
}

// Accumulate tiles (16x16 pixels + filter halo) that contain at least one pixel owned by this shard; cached per
// (frame, shard, halo).  Ownership is a function of the pixel only (pixel_owned, tn_kernels.h).
int accumulate_tile_list(tinsel_hip* r, const FrameParams& fp)
{
    const int reachLo = 1 + (int)floorf(fp.filterWidth), reachHi = (int)ceilf(fp.filterWidth);
    const int key[6] = { fp.width, fp.height, fp.shardRank, fp.shardWorld, fp.shardTile, reachLo*16 + reachHi };
    if (r->accTilesDev && memcmp(key, r->accTilesKey, sizeof(key)) == 0)
        return 0;
    const int tilesX = (fp.width + kAccTile - 1)/kAccTile, tilesY = (fp.height + kAccTile - 1)/kAccTile;
    const int shardX = (fp.width + fp.shardTile - 1)/fp.shardTile;
    std::vector<int> list;
    for (int ty = 0; ty < tilesY; ++ty)
    {
        for (int tx = 0; tx < tilesX; ++tx)
        {
            // candidate paths of this tile are generated at pixels [x0, x1] x [y0, y1]
            const int x0 = std::max(0, tx*kAccTile - reachLo), x1 = std::min(fp.width - 1, tx*kAccTile + kAccTile - 1 + reachHi);
            const int y0 = std::max(0, ty*kAccTile - reachLo), y1 = std::min(fp.height - 1, ty*kAccTile + kAccTile - 1 + reachHi);
            bool mine = false;
            for (int sy = y0/fp.shardTile; sy <= y1/fp.shardTile && !mine; ++sy)
                for (int sx = x0/fp.shardTile; sx <= x1/fp.shardTile && !mine; ++sx)
                    mine = ((sy*shardX + sx) % fp.shardWorld) == fp.shardRank;
            if (mine)
                list.push_back(ty*tilesX + tx);
        }
    }
    if (r->accTilesDev)
    {
        HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(r->accTilesDev);
        r->accTilesDev = nullptr;
    }
    HIP_TRY(hipMalloc((void**)&r->accTilesDev, sizeof(int)*(list.empty() ? 1 : list.size())));
    if (!list.empty())
        HIP_TRY(hipMemcpy(r->accTilesDev, list.data(), sizeof(int)*list.size(), hipMemcpyHostToDevice));
    r->accTilesCount = (int)list.size();
    memcpy(r->accTilesKey, key, sizeof(key));
    return 0;
}

// Paths resident per batch: 64 Mi for the wavefront pipelines (11 GB of path state of 288), 8 Mi for the megakernel arm.
// The split pipeline's trace launches end in a long tail (the slowest block of a deep traversal) and fewer, larger launches
// amortise it -- 524k-triangle config 765 / 941 / 992 Msamples/s at 8 / 32 / 64 Mi (round 1); the fused kernel wants its
// regions long (set_regions): cornell 2302 / 2803 Msamples/s at 8 / 64 Mi.  An explicit setting always wins.
size_t batch_slots(const tinsel_hip* r)
{
    if (!r->batchSlotsExplicit && r->pipeline != TINSEL_PIPELINE_MEGAKERNEL)
        return (size_t)64u << 20;
    return r->maxBatchSlots;
}

// Path slots one pass of this renderer's shard occupies: W*H for one shard, else its own tiles padded to full size.
size_t slots_per_pass(const tinsel_hip* r, int width, int height, int* tilesXOut = nullptr, int* ownedOut = nullptr)
{
    const int tilesX = (width + r->shardTile - 1)/r->shardTile;
    const int numTiles = tilesX*((height + r->shardTile - 1)/r->shardTile);
    const int owned = r->shardRank < numTiles ? (numTiles - r->shardRank + r->shardWorld - 1)/r->shardWorld : 0;
    if (tilesXOut) *tilesXOut = tilesX;
    if (ownedOut) *ownedOut = owned;
    if (r->shardWorld <= 1)
        return (size_t)width*height;
    return std::max<size_t>(1, (size_t)owned*r->shardTile*r->shardTile);
}

// The accumulate stage of a traced batch: adds the batch passes [fp.accBegin, fp.accEnd) to `target`.
int launch_accumulate(tinsel_hip* r, hipStream_t st, const FrameParams& fp, float4* target)
{
    const size_t npix = (size_t)fp.width*fp.height;
    ScopedTimer t(r, KN_ACCUMULATE, st);
    const int halo = 1 + (int)floorf(fp.filterWidth) + (int)ceilf(fp.filterWidth);
    if (halo <= kAccMaxHalo && fp.filterWidth >= 0.0f && fp.width < 65536 && fp.height < 65536)
    {
        int tiles = ((fp.width + kAccTile - 1)/kAccTile)*((fp.height + kAccTile - 1)/kAccTile);
        const int* tileList = nullptr;
        if (fp.shardWorld > 1)
        {
            if (accumulate_tile_list(r, fp))
                return -1;
            tileList = r->accTilesDev;
            tiles = r->accTilesCount;
        }
        if (tiles > 0)
        {
            const int span = 1 + (int)floorf(fp.filterWidth) + (int)ceilf(fp.filterWidth) + 1;     // reachLo + reachHi + 1
            // few tiles (a wave per SIMD or less): 512-thread workgroups, the second half only stages (tn_kernels.h; profiles/r03_y_ab_acc_wide.md)
            const bool wide = tiles <= r->numCUs*4;
            if (span == 3 && wide)
                hipLaunchKernelGGL((k_accumulate_tiled<3, 2*kBlock>), dim3(tiles), dim3(2*kBlock), 0, st, r->ps, fp, target, r->passSeeds, tileList);
            else if (span == 4 && wide)
                hipLaunchKernelGGL((k_accumulate_tiled<4, 2*kBlock>), dim3(tiles), dim3(2*kBlock), 0, st, r->ps, fp, target, r->passSeeds, tileList);
            else if (span == 3)
                hipLaunchKernelGGL((k_accumulate_tiled<3>), dim3(tiles), dim3(kBlock), 0, st, r->ps, fp, target, r->passSeeds, tileList);
            else if (span == 4)
                hipLaunchKernelGGL((k_accumulate_tiled<4>), dim3(tiles), dim3(kBlock), 0, st, r->ps, fp, target, r->passSeeds, tileList);
            else
                hipLaunchKernelGGL((k_accumulate_tiled<0>), dim3(tiles), dim3(kBlock), 0, st, r->ps, fp, target, r->passSeeds, tileList);
        }
    }
    else
    {
        const int gridPix = (int)((npix + kBlock - 1)/kBlock);
        hipLaunchKernelGGL(k_accumulate, dim3(gridPix), dim3(kBlock), 0, st, r->ps, fp, target, r->passSeeds);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// Blocks of the streaming kernels' grid = a quarter of the regions the batch is cut into (one region per wave, SplitState,
// tn_kernels.h).  A wave works through its region 64 entries at a time and a round is as long as its slowest lane, so
// regions should stay long as paths die (the last round of a region is the ragged one), yet there must be enough of them to
// balance: ~1024 positions per region, between 2 and 32 blocks per CU.  Fused kernel, cornell: a 1 M-path batch 1417 / 1520 /
// 1655 / 1747 Msamples/s at 16 / 8 / 4 / 2 blocks per CU (regions of 64 ... 512); a 64 Mi batch 2644 / 2735 / 2803 at 8 / 16 /
// 32 (regions of 8192 / 4096 / 2048).
int streaming_grid(const tinsel_hip* r, size_t slots, int pipeline)
{
    // (where the fused kernel's waves share their workgroup's regions -- three or more shadow rays per bounce, k_bounce -- the
    // regions may be twice as long: features 707 -> 740, features + probe 595 -> 622, veach +-0)
    const size_t regionTarget = (pipeline == TINSEL_PIPELINE_WAVEFRONT && r->neePerPath >= 3) ? 2048 : 1024;
    const size_t perBlock = regionTarget*(kBlock/kWave);
    const size_t blocks = (slots + perBlock - 1)/perBlock;
    // (at least as many workgroups as the chip holds at once -- TINSEL_HIP_GRID_MIN per CU, default 3: k_bounce and k_shade run three
    // waves per SIMD -- where the batch has that many 256-path pieces: a 1 M-path batch would otherwise leave the third wave slot empty)
    const int gridMin = pipeline == TINSEL_PIPELINE_WAVEFRONT ? r->bounceWaves : 3;
    const size_t lo = std::min<size_t>((size_t)r->numCUs*(size_t)gridMin, (slots + kBlock - 1)/kBlock), hi = (size_t)r->numCUs*(size_t)grid_mult();
    size_t grid = std::max<size_t>(1, std::min(hi, std::max(lo, blocks)));
    // The workgroups that HAVE work (regions are a whole number of waves long, so fewer than the grid may) as close to a whole number
    // of resident sets (gridMin per CU) as the region length allows within +-25 %: the last set of a launch is then full instead
    // of, say, two thirds empty.  Glass at 20 passes per batch 1262 -> 1291 Msamples/s, the 524k-triangle config 2024 -> 2034, the
    // fused configs +-0 (profiles/r03_s_ab_grid_round.txt); TINSEL_HIP_GRID_ROUND=0: off (A/B)
    const size_t resident = (size_t)r->numCUs*(size_t)gridMin;
    if (grid > 2*resident)
    {
        auto busy = [&](size_t g) {         // workgroups with work for a grid of g (set_regions' region length)
            const size_t regions = g*(kBlock/kWave);
            const size_t len = ((slots + regions - 1)/regions + kWave - 1)/kWave*kWave;
            return (slots + len*(kBlock/kWave) - 1)/(len*(kBlock/kWave));
        };
        size_t best = grid;
        double bestWaste = 2.0;
        for (size_t g = std::max(grid*3/4, 2*resident); g <= std::min(hi, grid*5/4); g += std::max<size_t>(1, resident/16))
        {
            const size_t b = busy(g);
            const double sets = (double)b/(double)resident;
            const double waste = std::ceil(sets) - sets;        // empty fraction of the last resident set
            if (waste < bestWaste - 1e-9)
            {
                bestWaste = waste;
                best = g;
            }
        }
        grid = best;
    }
    return (int)grid;
}

int set_regions(tinsel_hip* r, LaunchArgs& a, size_t slots, int gridPersist)
{
    a.ss = r->ss;
    a.ss.numRegions = (uint32_t)gridPersist*(kBlock/kWave);
    a.ss.regionLen = (uint32_t)(((slots + a.ss.numRegions - 1)/a.ss.numRegions + kWave - 1)/kWave*kWave);
    a.ss.bigRegions = a.ss.numRegions;
    a.ss.shortLen = a.ss.regionLen;
    if (a.ss.numRegions > r->splitMaxRegions || (size_t)a.ss.numRegions*a.ss.regionLen > r->splitCap)
        return fail("render: path buffers too small for this batch");
    r->lastRegions = a.ss.numRegions;
    return 0;
}

// The last `tailShare` of the batch's positions in regions 1/divide as long (SplitState::bigRegions / shortLen, tn_kernels.h): what the chip
// works on when a launch runs out.  `maxRegions` bounds their number (the region arrays; k_seg_prefix's LDS): where the uniform cut is
// already at the bound the long regions get longer.  Leaves `a` as set_regions made it when the batch is too small for any of that.
void split_tail_regions(tinsel_hip* r, LaunchArgs& a, size_t slots, double tailShare, int divide, size_t maxRegions)
{
    const uint32_t per = kBlock/kWave;                        // regions per group
    maxRegions = std::min<size_t>(maxRegions, r->splitMaxRegions);
    if (a.ss.regionLen < (uint32_t)(kWave*divide*2) || a.ss.numRegions < 64u*per || maxRegions < 128u*per)
        return;
    const double factor = (1.0 - tailShare) + tailShare*(double)divide;
    uint32_t L = a.ss.regionLen;
    if ((double)slots*factor/(double)L + 2.0*per > (double)maxRegions)
        L = (uint32_t)(((size_t)((double)slots*factor/(double)(maxRegions - 2*per)) + kWave)/kWave*kWave);
    const uint32_t S = L/(uint32_t)divide/kWave*kWave;        // short regions: a multiple of 64 positions
    const uint32_t big = (uint32_t)((double)slots*(1.0 - tailShare)/(double)L)/per*per;
    const size_t covered = (size_t)big*L;
    if (S < (uint32_t)kWave || big < per || covered >= slots)
        return;
    const size_t rest = slots - covered;
    const uint32_t small = (uint32_t)((rest + (size_t)S*per - 1)/((size_t)S*per))*per;
    if (big + small > maxRegions || (size_t)big*L + (size_t)small*S > r->splitCap)
        return;
    a.ss.regionLen = L;
    a.ss.bigRegions = big;
    a.ss.shortLen = S;
    a.ss.numRegions = big + small;
    r->lastRegions = a.ss.numRegions;
}

// A batch that ONE resident set of workgroups takes whole (cfg1: 256^2 x 16 passes = 683 region groups for 768 slots): two thirds of the CUs
// get three groups, a third gets two, and the launch lasts as long as three.  Cut so that every CU gets TWO long groups -- three quarters of
// the batch -- and the rest in groups a third as long, which the dispatcher deals out as slots come free: every CU ends up with the same
// work.  cornell 256^2 x 16 passes 2402 -> 2670 Msamples/s, 512^2 x 4 2600 -> 2919, veach 256^2 x 16 1176 -> 1324 (call Z10).
bool split_one_set(tinsel_hip* r, LaunchArgs& a, size_t slots)
{
    const uint32_t per = kBlock/kWave;
    const size_t cus = (size_t)r->numCUs;
    // (W workgroups resident per CU: W - 1 long groups per CU hold W/(W + 1) of the batch -- three waves: two groups, three quarters)
    const size_t W = (size_t)r->bounceWaves;
    const uint32_t L = (uint32_t)((slots*W/(W + 1))/((W - 1)*cus*per)/kWave*kWave);
    if (L < 3u*kWave)
        return false;
    const uint32_t S = L/3/kWave*kWave;
    const uint32_t big = (uint32_t)((W - 1)*cus)*per;
    const size_t covered = (size_t)big*L;
    if (covered >= slots)
        return false;
    const uint32_t small = (uint32_t)((slots - covered + (size_t)S*per - 1)/((size_t)S*per))*per;
    if (big + small > r->splitMaxRegions || (size_t)big*L + (size_t)small*S > r->splitCap)
        return false;
    a.ss.regionLen = L;
    a.ss.bigRegions = big;
    a.ss.shortLen = S;
    a.ss.numRegions = big + small;
    r->lastRegions = a.ss.numRegions;
    return true;
}

// set_regions + the short regions at the end.  The last eighth or so of the positions in regions a quarter as long: a workgroup's region group
// is 0.75 ms of a 5 ms launch (cornell, 20 passes) and a launch ends when its last workgroup does.  k_bounce alone (round 3, call Z5):
// cornell 1024^2 x 20 passes 3878 -> 4012 Msamples/s, x 8 3539 -> 3685, 512^2 x 16 2812 -> 3021, features 1183 -> 1292, veach 1080p
// 2414 -> 2610 (profiles/r03_z5_ab_tail_split.md).  TINSEL_HIP_TAIL_SPLIT="share,divide" (A/B; "0": off).  On return *grid is the number
// of region groups = the workgroups of a launch that gives every group its own.
int cut_regions(tinsel_hip* r, LaunchArgs& a, size_t slots, int* grid, size_t maxRegions)
{
    if (set_regions(r, a, slots, *grid))
        return -1;
    const char* tailEnv = getenv("TINSEL_HIP_TAIL_SPLIT");
    double share = -0.5;
    int divide = 4;
    if (tailEnv)
        sscanf(tailEnv, "%lf,%d", &share, &divide);
    // (three workgroups per CU are resident: k_bounce)
    if (share < 0.0 && maxRegions > 0 && (size_t)*grid <= (size_t)r->numCUs*r->bounceWaves && (size_t)*grid > (size_t)r->numCUs*(r->bounceWaves - 1) && split_one_set(r, a, slots))
    {
        *grid = (int)(a.ss.numRegions/(kBlock/kWave));
        return 0;
    }
    if (share < 0.0)
    {
        // a negative share: that multiple of ONE resident set's part of the batch (three workgroups per CU: k_bounce).  The default, half a
        // set's part, against a fixed eighth: cornell x 20 passes 4036 -> 4059, x 64 4203 -> 4221, features 1289 -> 1298, veach 1080p
        // 2610 -> 2621, gloss 10570 -> 10530 (call Z8)
        const double sets = (double)*grid/(double)(r->bounceWaves*r->numCUs);
        share = std::min(0.25, std::max(0.03, -share/std::max(1.0, sets)));
    }
    if (share > 0.0 && share < 0.9 && divide >= 2)
        split_tail_regions(r, a, slots, share, divide, maxRegions);
    *grid = (int)(a.ss.numRegions/(kBlock/kWave));
    return 0;
}

// k_bounce's LDS plan for the scene: does it close ranks through the waves' shading pools (returned), and how many waves per SIMD = workgroups
// per CU does it run at (r->bounceWaves: what the grid and the region cut are sized by).
//   * Pools (25 KB of LDS per workgroup) where rays can LEAVE the scene -- veach 1515 -> 1866 Msamples/s, features 755 -> 865, env_loft 3598 ->
//     3793, gloss 7584 -> 7934; between two facing planes every ray hits something and the pools only cost (cornell 2919 -> 2894) -- and
//     where they do not cost the third resident workgroup (features' 32-KB arena + pools would leave two).
//   * FOUR waves per SIMD (the 128-VGPR variant: 36 registers of loop invariants spilled in the prologue) where four workgroups' LDS fits
//     the CU, i.e. without the pools: cornell 4503 -> 4989 Msamples/s, cfg1 2961 -> 3100, features 1020 -> 1055; with the pools three
//     workgroups fit and the spills only cost (veach 2929 -> 2812, gloss 11 771 -> 11 190, env_loft 5064 -> 4915: profiles/r05_a_ab_waves4.md).
//     TINSEL_HIP_BOUNCE_WAVES=3 / 4 forces either (A/B, tests).
bool plan_bounce(tinsel_hip* r)
{
    static const char* repackEnv = getenv("TINSEL_HIP_REPACK");          // 0 / 1: never / always (A/B); default: open scenes
    const char* wavesEnv = getenv("TINSEL_HIP_BOUNCE_WAVES");            // (read per call: tests switch it)
    const size_t perCU = 160u*1024u;
    const size_t lds = stack_bytes(r), withPool = lds + kPoolWords*sizeof(uint32_t);
    const bool want = repackEnv ? atoi(repackEnv) != 0 : !r->sceneEnclosed;
    const bool repack = want && withPool*3 <= perCU && withPool <= (size_t)r->sharedMemLimit;
    const size_t perGroup = repack ? withPool : lds;
    r->bounceWaves = wavesEnv ? (atoi(wavesEnv) >= 4 ? 4 : 3) : (perGroup*4 <= perCU && !r->countDetail) ? 4 : 3;
    return repack;
}

int render_batch(tinsel_hip* r, hipStream_t st, const CameraParams& cam, FrameParams fp, bool accumulate = true)
{
    // path slots of this shard per pass and per batch (slot_pixel / slot_of, tn_kernels.h): rank-local numbering
    const size_t perPass = slots_per_pass(r, fp.width, fp.height, &fp.shardTilesX, &fp.shardOwnedTiles);
    const size_t slots = perPass*(size_t)fp.numPasses;
    if (slots >= (size_t)0xffffffffu)
        return fail("render: batch too large");
    fp.shardPerPass = (uint32_t)perPass;
    {
        auto magic = [](uint32_t d) { return 0xffffffffu/std::max(1u, d); };      // (tn_kernels.h div_magic)
        fp.perPassM = magic(fp.shardPerPass);
        fp.tileSqM = magic((uint32_t)fp.shardTile*(uint32_t)fp.shardTile);
        fp.tileM = magic((uint32_t)fp.shardTile);
        fp.tilesXM = magic((uint32_t)fp.shardTilesX);
    }
    fp.genCount = (uint32_t)slots;
    fp.accBegin = 0;
    fp.accEnd = fp.numPasses;
    fp.rrStart = r->rrStart;
    fp.repack = 0;
    fp.share = 0;
    const int gridFlat = (int)std::max<size_t>(1, (slots + kBlock - 1)/kBlock);
    const bool repackPlan = resolve_pipeline(r) == TINSEL_PIPELINE_WAVEFRONT && plan_bounce(r);
    int gridPersist = streaming_grid(r, slots, resolve_pipeline(r));
    // the trace kernels stride over the regions: by default one block per four regions like the others
    int gridTrace = gridPersist;
    r->lastBatchSlots = slots;

    const int pipeline = resolve_pipeline(r);
    if (pipeline != r->batchPipeline)
        return fail("render: path buffers were reserved for another pipeline");
    r->lastPipeline = pipeline;

    LaunchArgs a = batch_args(r, cam, fp);
    if (pipeline == TINSEL_PIPELINE_MEGAKERNEL)
    {
        ScopedTimer t(r, KN_MEGA, st);
        a.grid = gridFlat;
        launch_path(r, PK_MEGA, a, st);
    }
    else if (pipeline == TINSEL_PIPELINE_WAVEFRONT)
    {
        if (cut_regions(r, a, slots, &gridPersist, r->splitMaxRegions))
            return -1;
        a.grid = gridPersist;
        // (the shading pools and the waves per SIMD were planned before the batch was cut: plan_bounce)
        if (repackPlan)
        {
            a.fp.repack = 1;
            a.ldsBytes += (uint32_t)(kPoolWords*sizeof(uint32_t));
        }
        a.bounceWaves = r->bounceWaves;
        // bounces > 0: a workgroup's four regions as ONE stream dealt to its waves -- where a round is long (three or more shadow rays)
        // and where the regions are short (a small batch: the ragged last round of every region and bounce weighs more)
        {
            const char* shareEnv = getenv("TINSEL_HIP_BOUNCE_SHARE");            // 0 / 1: never / always (A/B, tests: read per call)
            const int shareLen = 512;       // cornell 256^2 x 16 passes (regions of 384): 2258 -> 2311 Msamples/s; 1024^2 x 20 (regions of 2048) +0.3 %
            a.fp.share = shareEnv ? (atoi(shareEnv) != 0) : (r->neePerPath >= 3 || (int)a.ss.regionLen <= shareLen);
        }
        // ONE launch takes every region through all the bounces (k_bounce, tn_kernels.h).  (Workgroup b takes region group b: a golden-section
        // step, which k_walk's static ranges need, loses here -- the dispatcher already hands workgroups out dynamically: veach 1970 -> 1904
        // Msamples/s, features 898 -> 880, cornell 2999 -> 2979.  The per-bounce launches of rounds 1-2, regions longest first, went in round 5.)
        a.bounce = 0;
        a.bounceEnd = fp.maxDepth;
        a.order = nullptr;
        ScopedTimer t(r, KN_BOUNCE, st);
        launch_path(r, PK_BOUNCE, a, st);
    }
    else
    {
        const bool walk = walk_records(r) != nullptr;
        // every mesh primitive walked by k_walk: the scan kernels run their lean variants with the scene-level stack only
        int meshPrims = 0;
        for (int m : r->primMesh)
            meshPrims += m >= 0 ? 1 : 0;
        const bool walkedOnly = walk && meshPrims == r->walkPrims.count && !r->scene.allInArena;
        const int stackScan = walkedOnly ? std::max(1, pick_stack(r->sceneStackNeed)) : r->stackNeed;
        const uint32_t ldsTrace = walkedOnly ? (uint32_t)(((size_t)stackScan*kBlock + kScanWords)*sizeof(uint32_t) + r->scene.arenaLdsBytes) : a.ldsBytes;
        // k_shade has no traversal stacks in LDS and reads a material per path: an arena too large to sit beside the stacks of the
        // trace kernels (32 KB) is still staged by it up to 60 KB (many_spheres, 39 KB of primitive and material records: k_shade
        // 7.8 -> 6.7 ms; staged in the trace kernels too it costs them their fourth wave per SIMD, 1380 -> 1280 Msamples/s, and
        // k_lights reads too little of it to repay the copy, 2.7 -> 3.1 ms)
        const uint32_t arenaLdsTrace = r->scene.arenaLdsBytes;
        const uint32_t arenaLdsShade = (arenaLdsTrace == 0 && r->scene.arenaBytes <= 61440u && !getenv("TINSEL_HIP_NO_LDS_SCENE")) ? r->scene.arenaBytes : arenaLdsTrace;
        const uint32_t ldsShade = r->scene.allInArena ? r->scene.arenaBytes : arenaLdsShade;
        a.walkedOnly = walkedOnly ? 1 : 0;
        static const bool noSceneWalkEarly = getenv("TINSEL_HIP_NO_SCENE_WALK") != nullptr;
        // the lean k_extend draws the light samples itself (tn_launch.h launches it when walkedOnly and not counting)
        // ... and so does the variant for a staged arena with meshes in HBM (glass): without the SLP vectoriser it fits 128 VGPRs and
        // saves k_lights' pass over the path state (k_extend 5.7 + k_lights 7.1 -> 11.2 ms per 32 passes, glass 1366 -> 1425 Msamples/s;
        // round 2, 170 VGPRs: 29.1 apart, 31.5 together).  TINSEL_HIP_LIGHTS_IN_EXTEND=0: k_lights as a kernel of its own (A/B)
        const bool mixedArena = !r->scene.allInArena && r->scene.arenaLdsBytes != 0 && r->scene.arenaLdsBytes == r->scene.arenaBytes;
        const bool lightsInMixed = !walkedOnly && !r->countDetail && mixedArena && !(!noSceneWalkEarly && !r->scene.flatScan);
        a.lightsInExtend = lightsInMixed ? 1 : 0;
        const bool lightsInExtend = (walkedOnly && !r->countDetail) || lightsInMixed;
        // No short regions at the end by default here (TINSEL_HIP_TAIL_SPLIT_SPLIT=1: A/B): the launches are many and short, k_walk cuts its
        // own list into static ranges, and more regions cost k_seg_prefix / k_walk more than the other kernels' tails gain -- the 524k-triangle
        // config 2319 -> 2254 Msamples/s, many_spheres 2108 -> 2082, glass +-0 (profiles/r03_z5_ab_tail_split.md).  (k_seg_prefix stages
        // the regions' counts in LDS: (sharedMemLimit - 1024)/4 of them at most where a walk list is built.)
        // (k_seg_prefix stages one count per region in LDS: where a walk list is built the grid is clamped to what fits)
        if (r->walkList != nullptr)
            gridPersist = std::max(1, std::min(gridPersist, (int)(seg_prefix_max_regions(r)/(kBlock/kWave))));
        if (cut_regions(r, a, slots, &gridPersist, (size_t)0))
            return -1;
        gridTrace = gridPersist;
        const size_t W = a.ss.numRegions;
        {
            ScopedTimer t(r, KN_GENERATE, st);
            a.grid = gridPersist;
            launch_path(r, PK_GENERATE, a, st);
        }
        // (not where k_walk does the walking: what is left for the scan kernels is too short for the two extra launches per
        // bounce to pay -- glass 1087 -> 1077, config 3 1891 -> 1881; many_spheres, scene BVH walked inline, 1168 -> 1290)
        // scenes the flat scan cannot take (more than 64 primitives): the scene-level walk with ray replacement (k_swalk, tn_swalk.h)
        // in the place of k_extend / k_shadow; the detail counters count the inline walks
        static const bool noSceneWalk = getenv("TINSEL_HIP_NO_SCENE_WALK") != nullptr;
        const bool sceneWalk = !noSceneWalk && !r->scene.flatScan && !r->countDetail && r->walkList != nullptr && !walk;
        const bool ordered = !walk && gridPersist > r->numCUs*2;
        auto order_regions = [&](const uint32_t* front, const uint32_t* back, uint32_t* out) {
            ScopedTimer t(r, KN_SEG, st);
            hipLaunchKernelGGL(k_region_order, dim3(1), dim3(kOrderBlock), 0, st, front, back, a.ss.numRegions, out);
        };
        for (int bounce = 0; bounce < fp.maxDepth; ++bounce)
        {
            a.bounce = bounce;
            // longest regions first (k_region_order, tn_kernels.h): the paths' order serves k_extend, k_lights and k_shade,
            // the shadow-ray bundles' order k_shadow; bounce 0's regions are all full
            a.order = nullptr;
            if (ordered && bounce > 0)
            {
                order_regions(r->ss.segFront + (size_t)bounce*W, r->ss.segBack + (size_t)bounce*W, r->regionOrder);
                a.order = r->regionOrder;
            }
            const uint32_t* const pathOrder = a.order;
            if (walk)
            {
                a.grid = gridPersist;
                if (launch_walk(r, st, a, r->ss.segFront + (size_t)bounce*W, false))
                    return -1;
            }
            if (sceneWalk)
            {
                a.grid = gridPersist;
                if (launch_swalk(r, st, a, r->ss.segFront + (size_t)bounce*W, r->ss.segBack + (size_t)bounce*W, false))
                    return -1;
            }
            else
            {
                ScopedTimer t(r, KN_EXTEND, st);
                a.grid = gridTrace;
                a.ldsBytes = ldsTrace;
                a.stackEntries = stackScan;
                launch_path(r, PK_EXTEND, a, st);
            }
            if (r->neePerPath > 0)
            {
                if (!lightsInExtend)
                {
                    ScopedTimer t(r, KN_LIGHTS, st);
                    a.grid = gridPersist;
                    a.ldsBytes = r->scene.allInArena ? r->scene.arenaBytes : arenaLdsTrace;
                    launch_path(r, PK_LIGHTS, a, st);
                }
                if (walk)
                {
                    a.grid = gridPersist;
                    if (launch_walk(r, st, a, r->ss.neeFront + (size_t)bounce*W, true))
                        return -1;
                }
                if (sceneWalk)
                {
                    a.grid = gridPersist;
                    if (launch_swalk(r, st, a, r->ss.neeFront + (size_t)bounce*W, r->ss.neeBack + (size_t)bounce*W, true))
                        return -1;
                }
                else
                {
                    if (ordered && bounce > 0)
                    {
                        order_regions(r->ss.neeFront + (size_t)bounce*W, r->ss.neeBack + (size_t)bounce*W, r->regionOrderNee);
                        a.order = r->regionOrderNee;
                    }
                    ScopedTimer t(r, KN_SHADOW, st);
                    a.grid = gridTrace;
                    a.ldsBytes = ldsTrace;
                    a.stackEntries = stackScan;
                    launch_path(r, PK_SHADOW, a, st);
                    a.order = pathOrder;
                }
            }
            {
                ScopedTimer t(r, KN_SHADE, st);
                // k_shade_sorted takes a region's paths class by class (tn_kernels.h): chosen PER SCENE.  It pays where a good share of a
                // bounce's paths are rays that LEFT the scene (a cheap class that otherwise idles through its wave-mates' shading) and the
                // path state is not already ordered by k_walk's front / back split: many_spheres 2087 -> 2122 Msamples/s, and in the split
                // pipeline veach 1902 -> 2003, features 1047 -> 1093; it loses in an enclosed scene (glass: no ray leaves, 17.5 -> 19.3 ms) and
                // where k_walk runs (the 524k-triangle config 6.47 -> 6.86 ms) (profiles/r03_g_ab_shade_sorted.txt, r04_e_rates.md).
                // TINSEL_HIP_SHADE_SORTED=0 / 1 forces either arm (A/B, tests).
                static const char* sortedEnv = getenv("TINSEL_HIP_SHADE_SORTED");
                const bool shadeSorted = sortedEnv ? atoi(sortedEnv) != 0 : (!r->sceneEnclosed && !walk);
                a.grid = gridPersist;
                a.shadeSorted = shadeSorted ? 1 : 0;
                a.ldsBytes = ldsShade + (shadeSorted ? (uint32_t)(kShadeListWords*sizeof(uint32_t)) : 0u);
                a.stackEntries = stackScan;
                a.scene.arenaLdsBytes = arenaLdsShade;
                launch_path(r, PK_SHADE, a, st);
                a.scene.arenaLdsBytes = arenaLdsTrace;
            }
        }
    }

    r->lastFp = fp;
    if (!accumulate)
        return 0;
    return launch_accumulate(r, st, fp, r->accum);
}

// traceOnly: the passes must fit ONE batch; their paths are traced (radiance left in ps.rad, r->lastFp set) but not
// accumulated -- the caller adds them pass range by pass range (launch_accumulate) into buffers of its choice (look-ahead).
int render_impl(tinsel_hip* r, const tinsel_camera* camera, const tinsel_options* options, int passes, hipStream_t st, bool traceOnly = false)
{
    if (!r || !camera || !options)
        return fail("render: null argument");
    if (!r->accum || options->width != r->width || options->height != r->height)
        return fail("render: options.width/height do not match the last tinsel_hip_init");
    if (passes < 1)
        return fail("render: passes must be >= 1");
    if (r->sceneDirty)
        return fail("render: a primitive was moved (tinsel_hip_set_primitive_transform): call tinsel_hip_rebuild_scene first");
    HIP_TRY(hipSetDevice(r->device));

    // return finished timing events to the pool
    for (TimedSpan& s : r->spans)
    {
        r->eventPool.push_back(s.start);
        r->eventPool.push_back(s.stop);
    }
    r->spans.clear();

    CameraParams cam;
    make_camera(*camera, options->width, options->height, cam);

    FrameParams fp;
    fp.width = options->width;
    fp.height = options->height;
    fp.npixM = 0xffffffffu/(uint32_t)std::max(1, options->width*options->height);
    fp.widthM = 0xffffffffu/(uint32_t)std::max(1, options->width);
    fp.maxDepth = options->max_depth;
    fp.shardRank = r->shardRank;
    fp.shardWorld = r->shardWorld;
    fp.shardTile = r->shardTile;
    fp.filterType = options->filter.type;
    fp.filterWidth = options->filter.width;
    fp.filterFalloff = options->filter.falloff;
    fp.filterOffset = options->filter.offset;
    fp.clampLen = options->clamp;
    fp.passBase = 0;
    fp.numPasses = 1;

    const size_t npix = (size_t)fp.width*fp.height;
    const int gridPix = (int)((npix + kBlock - 1)/kBlock);

    if (options->mode == TINSEL_MODE_NORMALS)
    {
        ScopedTimer t(r, KN_NORMALS, st);
        launch_normals(r, st, gridPix, cam, fp);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (options->mode != TINSEL_MODE_PATHTRACE)
        return 0;       // eComplexity is a no-op in the reference too (render.cpp:516-519)
    if (fp.maxDepth < 1)
        return 0;

    // pass seeds: passSeed[s] = (passIndex+s+1)-th output of Random(1).Rand().  The device keeps a TABLE of them, produced there by
    // one thread (k_pass_seeds) from the generator state the host keeps: this call's and the next thousand passes', so that a call
    // launches its path kernels and nothing else (a 256^2 x 16-pass batch is 0.43 ms of kernels: a third launch per call was 1.5 % of it).
    // The table is rewritten only when a call leaves it (every 1024 passes, or a rewind: tinsel_hip_set_pass_index) -- after a
    // device-wide wait, because kernels of another stream (look-ahead) may still read it; a stream other than the one that wrote
    // it waits for the writer's event.
    constexpr size_t kSeedsAhead = 1024;
    const bool covered = r->passSeedsCount > 0 && r->passIndex >= r->passSeedsBase &&
                         (size_t)(r->passIndex - r->passSeedsBase) + (size_t)passes <= r->passSeedsCount;
    if (!covered)
    {
        if (r->seedRngIndex > r->passIndex)
        {
            r->seedRng = Rng::seeded(1u);
            r->seedRngIndex = 0;
        }
        for (; r->seedRngIndex < r->passIndex; ++r->seedRngIndex)
            (void)r->seedRng.rand();
        if (r->passSeedsDev)
            HIP_TRY(hipDeviceSynchronize());
        const size_t want = (size_t)passes + kSeedsAhead;
        if (r->passSeedsCap < want)
        {
            if (r->passSeedsDev)
                (void)hipFree(r->passSeedsDev);
            r->passSeedsDev = nullptr;
            r->passSeedsCap = r->passSeedsCount = 0;
            HIP_TRY(hipMalloc((void**)&r->passSeedsDev, sizeof(uint32_t)*want));
            r->passSeedsCap = want;
        }
        if (!r->passSeedsReady)
            HIP_TRY(hipEventCreateWithFlags(&r->passSeedsReady, hipEventDisableTiming));
        hipLaunchKernelGGL(k_pass_seeds, dim3(1), dim3(1), 0, st, r->seedRng.s1, r->seedRng.s2, (int)want, r->passSeedsDev);
        HIP_TRY(hipEventRecord(r->passSeedsReady, st));
        r->passSeedsBase = r->passIndex;
        r->passSeedsCount = want;
        r->passSeedsStream = st;
    }
    else if (st != r->passSeedsStream)
        HIP_TRY(hipStreamWaitEvent(st, r->passSeedsReady, 0));
    r->passSeeds = r->passSeedsDev + (r->passIndex - r->passSeedsBase);

    const size_t perPass = slots_per_pass(r, fp.width, fp.height);
    int perBatch = (int)std::max<size_t>(1, batch_slots(r)/perPass);
    if (perBatch > passes)
        perBatch = passes;

    // Overlapped chunks.  A batch of several passes can be traced as TWO chunks of passes on two streams, each with its own dense state
    // (ensure_batch's lanes), both writing their slots of the one radiance array; a chunk's accumulate kernel follows its own trace
    // on its own stream and the previous chunk's accumulate by an event -- the framebuffer adds keep the reference's pass order, so no
    // bit changes (tests/test_gpu_switches.py) -- and the two chunks' kernels fill each other's tails.  Measured built in
    // (profiles/r04_k_ab_overlap.md, Msamples/s off -> on): it pays where a bounce is MANY SHORT launches that leave the chip half empty
    // at their ends -- the scene-level walk of scenes beyond the flat scan (k_seg_* + k_swalk twice a bounce: many_spheres 2118 -> 2304)
    // -- and nowhere else: the fused kernel is one launch that already ends in short regions (cornell 1024^2 x 20 passes 4173 -> 4145,
    // x 256 4338 -> 4268, veach 4K 2838 -> 2846, gloss 10972 -> 10588, a 1 M-path batch 2748 -> 2406: two launches, two tails);
    // k_walk's workgroups take a CU's whole LDS and gain nothing from a neighbour (the 524k-triangle config 2223 -> 2114, glass 1426 ->
    // 1418).  (Two RENDERERS on two streams had looked like +4 % on cornell, profiles/r04_j_two_streams.txt: that was the host's
    // share of a call overlapping, not the device's.)  Default: scenes whose scene level is walked by k_swalk, batches of 8 Mi paths
    // or more.  TINSEL_HIP_OVERLAP=0 / 1: never / wherever a batch has two passes (A/B, tests); TINSEL_HIP_OVERLAP_MIN_PATHS: the floor.
    int chunkPasses = perBatch;
    int lanes = 1;
    {
        const char* overlapEnv = getenv("TINSEL_HIP_OVERLAP");         // (read per call: tests switch it)
        const size_t minPaths = (size_t)8u << 20;
        const int pipeline = resolve_pipeline(r);
        const bool can = !traceOnly && perBatch >= 2 && pipeline != TINSEL_PIPELINE_MEGAKERNEL;
        static const bool noSceneWalk = getenv("TINSEL_HIP_NO_SCENE_WALK") != nullptr;
        const bool sceneWalked = pipeline == TINSEL_PIPELINE_WAVEFRONT_SPLIT && !r->scene.flatScan && !noSceneWalk && !r->countDetail &&
                                 !(r->walkPrims.count > 0 && r->walkEnabled);
        const bool want = overlapEnv ? atoi(overlapEnv) != 0 : (sceneWalked && perPass*(size_t)perBatch >= minPaths);
        if (can && want)
        {
            chunkPasses = (perBatch + 1)/2;
            lanes = 2;
        }
    }
    if (ensure_batch(r, perPass*(size_t)perBatch, fp.maxDepth, perPass*(size_t)chunkPasses, lanes))
        return -1;

    if (traceOnly && perBatch < passes)
        return fail("render: look-ahead batch does not fit");
    if (lanes == 2 && !r->laneStream)
    {
        HIP_TRY(hipStreamCreateWithFlags(&r->laneStream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&r->laneFork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&r->laneJoin, hipEventDisableTiming));
        for (int k = 0; k < 2; ++k)
            HIP_TRY(hipEventCreateWithFlags(&r->accDone[k], hipEventDisableTiming));
    }
    for (int done = 0; done < passes; done += perBatch)
    {
        const int n = std::min(perBatch, passes - done);
        if (lanes == 1 || n < 2)
        {
            fp.passBase = done;
            fp.numPasses = n;
            if (render_batch(r, st, cam, fp, !traceOnly))
                return -1;
            continue;
        }
        float4* const radBase = r->ps.rad;
        const int first = std::min(chunkPasses, (n + 1)/2);
        HIP_TRY(hipEventRecord(r->laneFork, st));                       // whatever the caller's stream holds comes first
        HIP_TRY(hipStreamWaitEvent(r->laneStream, r->laneFork, 0));
        int rc = 0;
        for (int c = 0; c < 2 && !rc; ++c)
        {
            hipStream_t s = c ? r->laneStream : st;
            if (c)
                lane_swap(r);
            fp.passBase = done + (c ? first : 0);
            fp.numPasses = c ? n - first : first;
            r->ps.rad = radBase + (size_t)(c ? first : 0)*perPass;
            r->ss.radOut = r->ps.rad;
            rc = render_batch(r, s, cam, fp, false);
            if (!rc && c)
                rc = hipStreamWaitEvent(s, r->accDone[0], 0) == hipSuccess ? 0 : fail("render: hipStreamWaitEvent");
            if (!rc)
                rc = launch_accumulate(r, s, r->lastFp, r->accum);
            if (!rc)
                rc = hipEventRecord(c ? r->laneJoin : r->accDone[0], s) == hipSuccess ? 0 : fail("render: hipEventRecord");
            r->ps.rad = radBase;
            r->ss.radOut = radBase;
            if (c)
                lane_swap(r);
        }
        if (!rc)
            rc = hipStreamWaitEvent(st, r->laneJoin, 0) == hipSuccess ? 0 : fail("render: hipStreamWaitEvent");
        if (rc)
        {
            // kernels already enqueued on the second stream still run over the path state and the accumulator: nothing the caller does
            // next (another render on another stream, init, destroy) may overtake them (lookahead_cancel does the same)
            (void)hipStreamSynchronize(r->laneStream);
            return -1;
        }
        // the test hooks read a whole batch (tinsel_hip_read_batch_radiance; queue_counts reports the second chunk's regions)
        r->lastBatchSlots = perPass*(size_t)n;
        r->lastFp.passBase = done;
        r->lastFp.numPasses = n;
        r->lastFp.genCount = (uint32_t)(perPass*(size_t)n);
        r->lastFp.accBegin = 0;
        r->lastFp.accEnd = n;
    }
    r->passIndex += (uint32_t)passes;
    return 0;
}


// ---------------------------------------------------------------------------
// Look-ahead for the reference's call pattern (main.cpp:246-250: Render() = ONE pass + the full-frame running sum to
// the host, 16 times per displayed frame, render.cu:1099-1102).  A call cannot return before its own pass has been
// copied out, and the copy cannot start before the pass is done -- inside one call there is nothing to overlap.  Across
// calls there is: while call k's image crosses PCIe, the passes call k+1 will most probably ask for (same camera, same
// options: the caller's loop) are already being traced into a SECOND accumulator, accumSpec = accum + those passes.  If
// the next call matches, the buffers swap and only the copy is left to do; if it does not (or any other entry point
// intervenes), the speculation is dropped -- accum itself was never touched by it.  Results are bit-identical to the
// plain path (same seeds, same adds in the same order); only the statistics counters run one call ahead.

void lookahead_cancel(tinsel_hip* r)
{
    if (!r || (!r->workStream && r->specQueue.empty()))
        return;
    // The work stream is waited for WHENEVER it exists, not only when shots are queued: a speculation that failed half-way
    // (lookahead_extend after render_impl had enqueued its kernels) leaves the queue empty and kernels in flight over the path
    // buffers the next plain render -- on another non-blocking stream -- is about to reuse.
    (void)hipSetDevice(r->device);
    if (r->workStream)
        (void)hipStreamSynchronize(r->workStream);
    for (tinsel_hip::SpecShot& shot : r->specQueue)
    {
        r->specFree.push_back(shot.buf);
        r->eventPool.push_back(shot.ready);
    }
    r->specQueue.clear();
}

void lookahead_release(tinsel_hip* r)
{
    lookahead_cancel(r);
    for (float4* b : r->specFree)
        (void)hipFree(b);
    r->specFree.clear();
    if (r->pinnedPtr) { (void)hipHostUnregister(r->pinnedPtr); r->pinnedPtr = nullptr; r->pinnedBytes = 0; }
}

// Speculate `depth` more calls: ONE batch of depth x passes passes is traced (as efficient as the resident path's batches),
// then each call's passes are added to a buffer of their own, chained: shot j = shot j-1 + call j's passes.
int lookahead_extend(tinsel_hip* r, const tinsel_camera* camera, const tinsel_options* options, int passes, int depth)
{
    const size_t bytes = sizeof(float4)*(size_t)r->width*r->height;
    const uint32_t committed = r->passIndex;
    r->passIndex = r->specNextPass;
    const int rc = render_impl(r, camera, options, passes*depth, r->workStream, true);
    r->passIndex = committed;
    if (rc)
        return -1;
    const float4* src = r->specQueue.empty() ? r->accum : r->specQueue.back().buf;
    for (int j = 0; j < depth; ++j)
    {
        float4* dst = nullptr;
        if (!r->specFree.empty())
        {
            dst = r->specFree.back();
            r->specFree.pop_back();
        }
        else
            HIP_TRY(hipMalloc((void**)&dst, bytes));
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, r->workStream));
        FrameParams fp = r->lastFp;
        fp.accBegin = j*passes;
        fp.accEnd = (j + 1)*passes;
        if (launch_accumulate(r, r->workStream, fp, dst))
            return -1;
        tinsel_hip::SpecShot shot = { dst, get_event(r) };
        HIP_TRY(hipEventRecord(shot.ready, r->workStream));
        r->specQueue.push_back(shot);
        src = dst;
    }
    r->specNextPass += (uint32_t)(passes*depth);
    return 0;
}

// calls per speculated batch: half a batch per speculation (two are in flight), at most 16 calls -- 4 at 1024^2, 16 for the
// small interactive frames; 0: one call's passes do not fit a batch
int lookahead_depth(const tinsel_hip* r, int passes)
{
    const size_t perPass = slots_per_pass(r, r->width, r->height);
    if (batch_slots(r) < perPass*(size_t)passes)
        return 0;
    const int fit = (int)std::max<size_t>(1, batch_slots(r)/(perPass*(size_t)passes));
    return r->lookaheadDepth > 0 ? std::max(1, std::min(r->lookaheadDepth, fit)) : std::max(1, std::min(16, fit/2));
}

int lookahead_streams(tinsel_hip* r)
{
    HIP_TRY(hipSetDevice(r->device));
    if (!r->workStream)
    {
        HIP_TRY(hipStreamCreateWithFlags(&r->workStream, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&r->copyStream, hipStreamNonBlocking));
    }
    return 0;
}

// the front of the speculation queue becomes the running sum (the caller has checked that it is this call's)
int lookahead_commit(tinsel_hip* r, int passes)
{
    tinsel_hip::SpecShot shot = r->specQueue.front();
    r->specQueue.pop_front();
    HIP_TRY(hipEventSynchronize(shot.ready));
    r->eventPool.push_back(shot.ready);
    r->specFree.push_back(r->accum);        // the previous running sum: copied out by the previous call, copied from by this shot
    r->accum = shot.buf;
    r->passIndex += (uint32_t)passes;
    return 0;
}

int lookahead_render(tinsel_hip* r, const tinsel_camera* camera, const tinsel_options* options, float* out_rgba, int passes)
{
    if (lookahead_streams(r))
        return -1;
    const size_t bytes = sizeof(float4)*(size_t)r->width*r->height;

    // 1. this call's passes: already traced (the front of the speculation queue) or traced now
    const bool hit = !r->specQueue.empty() && passes == r->specPasses && memcmp(camera, &r->specCamera, sizeof(*camera)) == 0 &&
                     memcmp(options, &r->specOptions, sizeof(*options)) == 0;
    if (hit)
    {
        if (lookahead_commit(r, passes))
            return -1;
    }
    else
    {
        lookahead_cancel(r);
        if (render_impl(r, camera, options, passes, r->workStream))
            return -1;
        HIP_TRY(hipStreamSynchronize(r->workStream));
        r->specNextPass = r->passIndex;
    }

    // 2. / 3. the running sum travels to the host while the speculation queue is kept between `depth` and 2 x depth calls deep
    //    (a batch of `depth` calls is traced while the previous batch's running sums are copied out one call at a time).
    //    The caller's array is NOT page-locked by default: the reference's caller frees and re-allocates it on every reshape
    //    (main.cpp:73-87: delete[] g_pixels, then Renderer::Init), and a registration must not outlive the memory it names.
    //    A copy to pageable memory blocks this thread while it runs, so the next batch is launched FIRST (it is needed `depth`
    //    calls from now; the launches cost the copy ~0.1 ms of delay every `depth` calls).  TINSEL_LOOKAHEAD_PIN_OUTPUT (the
    //    caller guarantees the array outlives the renderer or the next Init): registered in place, the copy is asynchronous
    //    and starts first.
    const bool pin = r->lookahead == TINSEL_LOOKAHEAD_PIN_OUTPUT;
    if (r->pinnedPtr && (!pin || r->pinnedPtr != (void*)out_rgba || r->pinnedBytes != bytes))
    {
        (void)hipHostUnregister(r->pinnedPtr);
        r->pinnedPtr = nullptr;
        r->pinnedBytes = 0;
    }
    if (pin && !r->pinnedPtr)
    {
        if (hipHostRegister(out_rgba, bytes, hipHostRegisterDefault) == hipSuccess)
        {
            r->pinnedPtr = out_rgba;
            r->pinnedBytes = bytes;
        }
        else
            (void)hipGetLastError();        // pageable copy below: still correct
    }
    const bool asyncCopy = r->pinnedPtr != nullptr;
    if (asyncCopy)
        HIP_TRY(hipMemcpyAsync(out_rgba, r->accum, bytes, hipMemcpyDeviceToHost, r->copyStream));

    if (options->mode == TINSEL_MODE_PATHTRACE && options->max_depth >= 1)
    {
        const int depth = lookahead_depth(r, passes);
        if (depth > 0 && (int)r->specQueue.size() <= depth)
        {
            r->specCamera = *camera;
            r->specOptions = *options;
            r->specPasses = passes;
            if (lookahead_extend(r, camera, options, passes, depth))
                lookahead_cancel(r);            // could not speculate: the plain path still works
        }
    }

    if (!asyncCopy)
        HIP_TRY(hipMemcpyAsync(out_rgba, r->accum, bytes, hipMemcpyDeviceToHost, r->copyStream));
    HIP_TRY(hipStreamSynchronize(r->copyStream));
    return 0;
}

} // namespace

// ===========================================================================
// C-ABI

// ---------------------------------------------------------------------------
// device-side mesh BVH build (tn_lbvh.h); the reference trees stay the default and the parity path

namespace {

// one device allocation carved into aligned pieces (hipMalloc / hipFree dominate a small build otherwise)
struct ScratchPool
{
    unsigned char* base = nullptr;
    size_t size = 0, used = 0;
    ~ScratchPool() { if (base) (void)hipFree(base); }
    static size_t padded(size_t bytes) { return (bytes + 255) & ~size_t(255); }
    bool reserve(size_t bytes) { size = bytes; return hipMalloc((void**)&base, bytes ? bytes : 1) == hipSuccess; }
    template <class T> T* get(size_t count)
    {
        T* p = (T*)(base + used);
        used += padded(sizeof(T)*count);
        return used <= size ? p : nullptr;
    }
};

// Builds a BVH over mesh `dm`'s triangles on the device -- TINSEL_BVH_LBVH: Karras' hierarchy over the Morton order + box fitting level by
// level; TINSEL_BVH_PLOC: agglomerative clustering over the same order (tn_lbvh.h) -- and emits it with its top numbered breadth-first
// (k_walk stages a prefix of the node array into LDS).  On success fills nodes / root / stackNeed / topCount of `out`.
int build_device_bvh(tinsel_hip* r, const DevMesh& dm, DevMesh& out, int mode)
{
    const int n = dm.numTris;
    // scratch of the library's own sort and scan (tn_sort.h), in bytes
    const size_t sortBytes = sort_scratch_ints((size_t)n)*sizeof(int), scanBytes = scan_scratch_ints((size_t)n)*sizeof(int);
    const size_t N = (size_t)n;
    ScratchPool tmp;
    if (!tmp.reserve(ScratchPool::padded(6*4) + 2*ScratchPool::padded(N*8) + ScratchPool::padded((N - 1)*8) + 2*ScratchPool::padded((2*N - 1)*4) +
                     ScratchPool::padded((2*N - 1)*24) + 7*ScratchPool::padded(N*4) + ScratchPool::padded(sortBytes) + ScratchPool::padded(scanBytes) +
                     ScratchPool::padded(kWalkTopNodes*4) + 256))
        return fail("build_mesh_bvh: device allocation failed");
    uint32_t* bounds = tmp.get<uint32_t>(6);
    unsigned long long* keys = tmp.get<unsigned long long>(N);
    unsigned long long* sorted = tmp.get<unsigned long long>(N);
    int2* children = tmp.get<int2>(N - 1);
    int* parent = tmp.get<int>(2*N - 1);
    float* boxes = tmp.get<float>((2*N - 1)*6);
    int* height = tmp.get<int>(2*N - 1);
    int* visits = tmp.get<int>(N);          // LBVH: the fitting passes' generations; PLOC: nearest neighbours
    int* clustersA = tmp.get<int>(N);
    int* clustersB = tmp.get<int>(N);
    int* keep = tmp.get<int>(N);
    int* offsets = tmp.get<int>(N);
    int* isTop = tmp.get<int>(N);
    int* perm = tmp.get<int>(N);
    unsigned char* sortTmp = tmp.get<unsigned char>(sortBytes);
    unsigned char* scanTmp = tmp.get<unsigned char>(scanBytes);
    int* topIds = tmp.get<int>(kWalkTopNodes);
    int* nextId = tmp.get<int>(2);          // [0] the next internal node id, [1] clusters left after a round
    Node64* nodes = nullptr;
    if (!bounds || !keys || !sorted || !children || !parent || !boxes || !height || !visits || !clustersA || !clustersB || !keep || !offsets || !isTop ||
        !perm || !sortTmp || !scanTmp || !topIds || !nextId || hipMalloc((void**)&nodes, sizeof(Node64)*(N - 1)) != hipSuccess)
        return fail("build_mesh_bvh: device allocation failed");

    const uint32_t init[6] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u };
    const unsigned grid = (unsigned)((n + 255)/256);
    int rc = 0;
    do
    {
        if (hipMemcpyAsync(bounds, init, sizeof(init), hipMemcpyHostToDevice, nullptr) != hipSuccess ||
            hipMemsetAsync(visits, 0, sizeof(int)*(size_t)n, nullptr) != hipSuccess) { rc = fail("build_mesh_bvh: init failed"); break; }
        hipLaunchKernelGGL(k_lbvh_bounds, dim3(grid < 256u ? grid : 256u), dim3(256), 0, nullptr, dm.tris, n, bounds);
        hipLaunchKernelGGL(k_lbvh_keys, dim3(grid), dim3(256), 0, nullptr, dm.tris, n, bounds, keys);
        // keys = Morton code << 32 | triangle index, written in index order: a STABLE sort by the code's bytes (bits 32..63) is the sort by the
        // whole key
        radix_sort_keys(keys, sorted, (size_t)n, 32, 64, reinterpret_cast<int*>(sortTmp), nullptr);
        hipLaunchKernelGGL(k_lbvh_leaves, dim3(grid), dim3(256), 0, nullptr, dm.tris, sorted, n, boxes, height);
        if (mode == TINSEL_BVH_PLOC)
        {
            // agglomerative rounds over the Morton order; the host reads the number of clusters left after every round (8 B)
            const int firstId = n - 2;
            if (hipMemcpyAsync(nextId, &firstId, sizeof(int), hipMemcpyHostToDevice, nullptr) != hipSuccess) { rc = fail("build_mesh_bvh: init failed"); break; }
            hipLaunchKernelGGL(k_ploc_init, dim3(grid), dim3(256), 0, nullptr, n, clustersA);
            int c = n;
            int* cur = clustersA;
            int* nxt = clustersB;
            int rounds = 0;
            while (c > 1 && !rc)
            {
                const unsigned g = (unsigned)((c + 255)/256);
                hipLaunchKernelGGL(k_ploc_nearest, dim3(g), dim3(256), 0, nullptr, (const int*)cur, c, (const float*)boxes, visits);
                hipLaunchKernelGGL(k_ploc_merge, dim3(g), dim3(256), 0, nullptr, cur, c, (const int*)visits, boxes, children, height, nextId, keep);
                exclusive_scan(keep, offsets, (size_t)c, reinterpret_cast<int*>(scanTmp), nullptr);
                hipLaunchKernelGGL(k_ploc_compact, dim3(g), dim3(256), 0, nullptr, (const int*)cur, c, (const int*)keep, (const int*)offsets, nxt, nextId + 1);
                int left = 0;
                if (hipMemcpy(&left, nextId + 1, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail("build_mesh_bvh: read-back failed"); break; }
                if (left >= c || left < 1 || ++rounds > 4096) { rc = fail("build_mesh_bvh: clustering made no progress"); break; }
                c = left;
                std::swap(cur, nxt);
            }
            if (rc)
                break;
        }
        else
        {
            hipLaunchKernelGGL(k_lbvh_hierarchy, dim3(grid), dim3(256), 0, nullptr, sorted, n, children, parent);
            // one pass per tree level (<= 63 for 62-bit keys); look at the root every 16 passes
            int rootGen = 0;
            for (int pass = 2; pass <= 66 && !rootGen; )
            {
                for (int k = 0; k < 16; ++k, ++pass)
                    hipLaunchKernelGGL(k_lbvh_fit_pass, dim3(grid), dim3(256), 0, nullptr, n, pass, children, boxes, height, visits);
                if (hipMemcpy(&rootGen, visits, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
                    break;
            }
            if (!rootGen) { rc = fail("build_mesh_bvh: box fitting did not reach the root"); break; }
        }

        // the tree's top, breadth-first: the host walks the first kWalkTopNodes internal nodes (children: 8 B per node)
        std::vector<int2> hostChildren(N - 1);
        if (hipMemcpy(hostChildren.data(), children, sizeof(int2)*(N - 1), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail("build_mesh_bvh: read-back failed"); break; }
        std::vector<int> topOrder;
        topOrder.reserve(kWalkTopNodes);
        {
            std::vector<int> frontier(1, 0);
            size_t head = 0;
            while (head < frontier.size() && (int)topOrder.size() < kWalkTopNodes)
            {
                const int id = frontier[head++];
                topOrder.push_back(id);
                const int2 ch = hostChildren[(size_t)id];
                if (ch.x < n - 1) frontier.push_back(ch.x);
                if (ch.y < n - 1) frontier.push_back(ch.y);
            }
        }
        const int top = (int)topOrder.size();
        if (hipMemsetAsync(isTop, 0, sizeof(int)*(N - 1), nullptr) != hipSuccess ||
            hipMemcpy(topIds, topOrder.data(), sizeof(int)*(size_t)top, hipMemcpyHostToDevice) != hipSuccess) { rc = fail("build_mesh_bvh: upload failed"); break; }
        int* rank = keep;           // (free again)
        hipLaunchKernelGGL(k_bfs_mark, dim3((unsigned)((top + 255)/256)), dim3(256), 0, nullptr, (const int*)topIds, top, isTop, rank);
        exclusive_scan(isTop, offsets, N - 1, reinterpret_cast<int*>(scanTmp), nullptr);
        hipLaunchKernelGGL(k_bfs_perm, dim3(grid), dim3(256), 0, nullptr, n - 1, top, (const int*)isTop, (const int*)rank, (const int*)offsets, perm);
        hipLaunchKernelGGL(k_lbvh_emit_perm, dim3(grid), dim3(256), 0, nullptr, sorted, n, children, boxes, (const int*)perm, nodes);
        int rootHeight = 0;
        if (hipGetLastError() != hipSuccess || hipMemcpy(&rootHeight, height, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail("build_mesh_bvh: kernels failed"); break; }
        out = dm;
        out.nodes = nodes;
        out.root = 0;                   // perm[0] == 0: the root is the first node of the breadth-first walk
        out.stackNeed = rootHeight + 1;
        out.topCount = top;
        out.numInternal = n - 1;
    } while (false);
    if (rc)
        (void)hipFree(nodes);
    else
        r->lbvhAllocs.push_back(nodes);
    return rc;
}

} // namespace

extern "C" {

const char* tinsel_hip_last_error(void) { return g_error.c_str(); }

tinsel_hip* tinsel_hip_create(const tinsel_scene_desc* desc, int device_index)
{
    if (!desc || !desc->primitives || desc->num_primitives <= 0 || !desc->bvh_nodes || desc->num_bvh_nodes <= 0)
    {
        fail("create: empty scene (Scene::Build must have run)");
        return nullptr;
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    {
        fail("create: no HIP device visible -- this library has no CPU fallback");
        return nullptr;
    }
    if (device_index < 0 || device_index >= ndev)
    {
        fail("create: bad device index");
        return nullptr;
    }
    HIP_TRY_NULL(hipSetDevice(device_index));

    tinsel_hip* r = new tinsel_hip();
    r->device = device_index;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_index) == hipSuccess)
    {
        r->numCUs = prop.multiProcessorCount;
        r->sharedMemLimit = (int)prop.sharedMemPerBlock;
    }
    // every kernel's dynamic-LDS limit is raised here, once, and the results are checked: a device that grants less than it reports is
    // refused now, by the kernel's name, not at some later launch with a generic error
    prepare_kernels_once(r);
    if (!r->prepRefused.empty())
    {
        fail("create: the device refused " + std::to_string(r->sharedMemLimit) + " B of dynamic LDS for " + r->prepRefused);
        delete r;
        return nullptr;
    }

    if (const char* e = getenv("TINSEL_HIP_BATCH_PATHS"))
    {
        long long v = atoll(e);
        if (v >= 65536)
        {
            r->maxBatchSlots = (size_t)v;
            r->batchSlotsExplicit = true;
        }
    }

    DevScene& sc = r->scene;
    memset(&sc, 0, sizeof(sc));

    const int P = desc->num_primitives;
    std::vector<Prim64> prims((size_t)P);
    std::vector<Mat128> mats((size_t)P);
    std::vector<Moving64> moving((size_t)P);
    std::vector<DevMesh> meshes;
    std::vector<int32_t> lights;
    std::map<uint64_t, uint32_t> meshIndex;     // MeshGeometry::id (util.h:20) -> DevScene::meshes index
    ArenaBuilder arena;
    int maxMeshNeed = 0;
    int totalLightSamples = 0;

    bool ok = true;

    // Where meshes live.  A scene all of whose meshes are small (<= 4 KB each; cornell's boxes) keeps them in the arena,
    // is staged whole into LDS and runs the fused kernel.  Once ONE mesh has to live in HBM the scene runs the split
    // pipeline, and there only meshes of a few triangles are worth walking inline from the arena: the others go to HBM
    // too and are walked by k_walk (glass.tin's 12-triangle cube: k_extend + k_shadow + k_walk 34.7 -> 31.3 ms; its
    // 2-triangle lamp stays inline -- every shadow ray enters its box).
    auto mesh_bytes_estimate = [](const tinsel_mesh_geometry& g) {
        return (size_t)g.num_nodes*32 + (size_t)(g.num_indices/3)*52 + (size_t)g.num_vertices*12;
    };
    bool sceneHasBigMesh = false;
    for (int i = 0; i < P; ++i)
        if (desc->primitives[i].type == TINSEL_GEOM_MESH && mesh_bytes_estimate(desc->primitives[i].geo.mesh) > kSmallMeshBytes)
            sceneHasBigMesh = true;
    const int inlineMaxTris = getenv("TINSEL_HIP_INLINE_MAX_TRIS") ? atoi(getenv("TINSEL_HIP_INLINE_MAX_TRIS")) : kInlineMaxTris;
    auto lives_in_arena = [&](size_t meshBytes, int numTris) {
        // TINSEL_HIP_SMALL_MESH_BYTES: test / A-B knob (0 = every mesh lives in HBM, so the queue sort and k_walk see them all)
        if (getenv("TINSEL_HIP_SMALL_MESH_BYTES"))
            return meshBytes <= (size_t)atoll(getenv("TINSEL_HIP_SMALL_MESH_BYTES"));
        return sceneHasBigMesh ? numTris <= inlineMaxTris && meshBytes <= kSmallMeshBytes : meshBytes <= kSmallMeshBytes;
    };

    for (int i = 0; i < P && ok; ++i)
    {
        const tinsel_primitive& p = desc->primitives[i];
        Prim64& o = prims[(size_t)i];
        memset(&o, 0, sizeof(o));

        make_material(p, mats[(size_t)i]);
        r->primEndScale.push_back(p.end_transform.s);
        if (p.light_samples > 0)
        {
            if (p.type == TINSEL_GEOM_PLANE)
            {
                fail("create: a plane cannot be a light (PrimitiveSample asserts, intersection.h:871-875)");
                ok = false;
                break;
            }
            lights.push_back(i);
            totalLightSamples += p.light_samples;
        }

        const bool isStatic = memcmp(&p.start_transform, &p.end_transform, sizeof(tinsel_transform)) == 0;
        const Xform xs = to_xform(p.start_transform), xe = to_xform(p.end_transform);
        r->primStart.push_back(xs);
        r->primEnd.push_back(xe);
        // InterpolateTransform(a, a, t) is t-independent: static primitives get it evaluated once, with the same function
        set_prim_pose(o, xs, xe, isStatic);
        // (a Moving64 slot for EVERY primitive, its own index: a static one may start to move, tinsel_hip_set_primitive_transform)
        o.moving = (uint32_t)i;
        moving[(size_t)i] = make_moving(xs, xe);

        if (p.type == TINSEL_GEOM_SPHERE)
        {
            o.type = kPrimSphere;
            o.g0 = p.geo.sphere.radius;
        }
        else if (p.type == TINSEL_GEOM_PLANE)
        {
            o.type = kPrimPlane;
            o.g0 = p.geo.plane.plane[0]; o.g1 = p.geo.plane.plane[1]; o.g2 = p.geo.plane.plane[2]; o.g3 = p.geo.plane.plane[3];
        }
        else if (p.type == TINSEL_GEOM_MESH)
        {
            o.type = kPrimMesh;
            const tinsel_mesh_geometry& g = p.geo.mesh;
            // Key on MeshGeometry::id and rewrite EVERY instance (the reference forgets both: render.cu:1000-1011)
            auto it = meshIndex.find(g.id);
            if (it != meshIndex.end())
            {
                o.mesh = it->second;
            }
            else
            {
                const int numTris = g.num_indices/3;
                if (numTris <= 0 || !g.positions || !g.normals || !g.indices || !g.nodes || !g.cdf)
                {
                    fail("create: mesh primitive with missing arrays");
                    ok = false;
                    break;
                }
                ConvertedBvh cb;
                // meshes that will live in HBM: the upper levels breadth-first (k_walk's LDS-resident top, tn_walk.h)
                if (!convert_bvh(g.nodes, g.num_nodes, numTris, lives_in_arena(mesh_bytes_estimate(g), numTris) ? 0 : kWalkTopNodes, cb))
                {
                    fail("create: malformed mesh BVH");
                    ok = false;
                    break;
                }
                std::vector<Tri48> tris((size_t)numTris);
                for (int t = 0; t < numTris; ++t)
                {
                    const int i0 = g.indices[t*3 + 0], i1 = g.indices[t*3 + 1], i2 = g.indices[t*3 + 2];
                    if (i0 < 0 || i1 < 0 || i2 < 0 || i0 >= g.num_vertices || i1 >= g.num_vertices || i2 >= g.num_vertices)
                    {
                        fail("create: mesh index out of range");
                        ok = false;
                        break;
                    }
                    Tri48& T = tris[(size_t)t];
                    T.ax = g.positions[i0].x; T.ay = g.positions[i0].y; T.az = g.positions[i0].z; T.i0 = i0;
                    T.bx = g.positions[i1].x; T.by = g.positions[i1].y; T.bz = g.positions[i1].z; T.i1 = i1;
                    T.cx = g.positions[i2].x; T.cy = g.positions[i2].y; T.cz = g.positions[i2].z; T.i2 = i2;
                }
                if (!ok)
                    break;

                DevMesh dm;
                memset(&dm, 0, sizeof(dm));
                dm.root = cb.root;
                dm.numTris = numTris;
                dm.stackNeed = cb.maxLeafDepth + 1;
                dm.topCount = cb.topCount;
                dm.numInternal = (int32_t)cb.nodes.size();
                // one internal node over two one-triangle leaves (a quad): walked without stack or loop (ray_mesh_two_leaves)
                dm.twoLeaves = (cb.nodes.size() == 1 && !(cb.root & kLeafBit) && (cb.nodes[0].left & kLeafBit) && (cb.nodes[0].right & kLeafBit)) ? 1 : 0;
                const size_t meshBytes = cb.nodes.size()*sizeof(Node64) + tris.size()*sizeof(Tri48) + (size_t)g.num_vertices*12 + (size_t)numTris*4;
                if (lives_in_arena(meshBytes, numTris))
                {
                    // offsets for now; turned into pointers once the arena has its device address
                    dm.inArena = 1;
                    dm.offNodes = (uint32_t)arena.add(cb.nodes.data(), cb.nodes.size());
                    dm.offTris = (uint32_t)arena.add(tris.data(), tris.size());
                    dm.offNormals = (uint32_t)arena.add(&g.normals[0].x, (size_t)g.num_vertices*3);
                    dm.offCdf = (uint32_t)arena.add(g.cdf, (size_t)numTris);
                }
                else
                {
                    dm.nodes = r->sceneMem.upload(cb.nodes.data(), cb.nodes.size());
                    dm.tris = r->sceneMem.upload(tris.data(), tris.size());
                    dm.normals = r->sceneMem.upload(&g.normals[0].x, (size_t)g.num_vertices*3);
                    dm.cdf = r->sceneMem.upload(g.cdf, (size_t)numTris);
                    if ((!cb.nodes.empty() && !dm.nodes) || !dm.tris || !dm.normals || !dm.cdf)
                    {
                        fail("create: device allocation failed (mesh)");
                        ok = false;
                        break;
                    }
                }
                if (dm.stackNeed > maxMeshNeed)
                    maxMeshNeed = dm.stackNeed;
                r->meshNumVertices.push_back(g.num_vertices);
                r->meshIndices.emplace_back(g.indices, g.indices + (size_t)numTris*3);
                r->meshRootLo.push_back(V3(g.nodes[0].lower.x, g.nodes[0].lower.y, g.nodes[0].lower.z));      // PrimitiveBounds reads nodes[0].bounds (intersection.h:928)
                r->meshRootHi.push_back(V3(g.nodes[0].upper.x, g.nodes[0].upper.y, g.nodes[0].upper.z));
                r->meshArea.push_back(g.area);
                o.mesh = (uint32_t)meshes.size();
                meshIndex[g.id] = o.mesh;
                meshes.push_back(dm);
            }
        }
        else
        {
            fail("create: unknown primitive type");
            ok = false;
        }
    }

    r->primMesh.assign((size_t)P, -1);
    for (int i = 0; i < P && ok; ++i)
    {
        set_prim_derived(prims[(size_t)i]);
        if (prims[(size_t)i].type == kPrimMesh)
            r->primMesh[(size_t)i] = (int)prims[(size_t)i].mesh;
    }

    ConvertedBvh sceneBvh;
    if (ok && !convert_bvh(desc->bvh_nodes, desc->num_bvh_nodes, P, 0, sceneBvh))
    {
        fail("create: malformed scene BVH");
        ok = false;
    }

    if (ok)
    {
        const int need = sceneBvh.maxLeafDepth + 1 + maxMeshNeed;
        r->stackNeed = pick_stack(need);
        if (r->stackNeed < 0)
        {
            fail("create: BVH too deep for the 156-entry LDS traversal stack");
            ok = false;
        }
    }

    if (ok)
    {
        // leaf boxes of the scene BVH, by primitive index (flat scene-level scan)
        std::vector<PrimBox> boxes((size_t)P);
        std::vector<char> seen((size_t)P, 0);
        for (int k = 0; k < desc->num_bvh_nodes; ++k)
        {
            const tinsel_bvh_node& nd = desc->bvh_nodes[k];
            if (!ref_is_leaf(nd) || nd.left_index >= (uint32_t)P)
                continue;
            boxes[nd.left_index] = make_prim_box(nd);
            seen[nd.left_index] = 1;
        }
        bool everyPrimHasALeaf = true;
        for (int k = 0; k < P; ++k)
            everyPrimHasALeaf = everyPrimHasALeaf && seen[(size_t)k];
        const bool flatScan = everyPrimHasALeaf && P <= 64 && !getenv("TINSEL_HIP_NO_FLAT_SCAN");

        // primitives whose mesh lives in HBM (flat-scan scenes, the first 7): their leaf-box test sorts the ray queues
        // (k_generate, k_shade), and they are walked by k_walk ahead of the scan kernels (tn_walk.h).  That includes trees
        // that stay in L1/L2 (glass.tin's 1280-triangle sphere: 80 KB of nodes; its 12-triangle cube): what the lean kernel
        // buys there is ray replacement for incoherent bounces (glass, maxDepth 12: 924 -> 1001 Msamples/s with the sphere,
        // 1050 with the cube too; with k_walk's work list in image order the sphere had lost, 732 inline vs 657-690).
        const int walkMinTris = getenv("TINSEL_HIP_WALK_MIN_TRIS") ? atoi(getenv("TINSEL_HIP_WALK_MIN_TRIS")) : kInlineMaxTris + 1;
        r->binPrims.count = 0;
        r->walkPrims.count = 0;
        if (flatScan)
            for (int k = 0; k < P && r->binPrims.count < 7; ++k)
                if (prims[(size_t)k].type == kPrimMesh && !meshes[prims[(size_t)k].mesh].inArena)
                {
                    r->binPrims.prim[r->binPrims.count++] = k;
                    if (meshes[prims[(size_t)k].mesh].numTris >= walkMinTris)
                    {
                        prims[(size_t)k].flags |= kPrimWalked | ((uint32_t)r->walkPrims.count << kPrimWalkLaneShift);
                        r->walkPrimMesh[r->walkPrims.count] = (int)prims[(size_t)k].mesh;
                        r->walkPrims.prim[r->walkPrims.count++] = k;
                    }
                }
        r->walkEnabled = !getenv("TINSEL_HIP_NO_WALK");
#ifdef TN_WALK_PROF
        if (hipMalloc((void**)&r->walkProf, 16*sizeof(unsigned long long)) == hipSuccess)
            (void)hipMemset(r->walkProf, 0, 16*sizeof(unsigned long long));
#endif

        // one contiguous arena: scene BVH, Prim64, Mat128, moving poses, lights, mesh table (+ small meshes, added above)
        const size_t offNodes = arena.add(sceneBvh.nodes.data(), sceneBvh.nodes.size());
        const size_t offPrims = arena.add(prims.data(), prims.size());
        const size_t offMats = arena.add(mats.data(), mats.size());
        const size_t offMoving = arena.add(moving.data(), moving.size());
        const size_t offLights = arena.add(lights.data(), lights.size());
        const size_t offMeshes = arena.add(meshes.data(), meshes.size());

        // the always-hit planes once more, four by four, for the flat scan (trace_flat; TINSEL_HIP_NO_PLANE_TABLE: A/B)
        std::vector<float> planeEq;
        std::vector<int32_t> planeIdx;
        if (flatScan)
        {
            for (int k = 0; k < P; ++k)
                if (prims[(size_t)k].type == kPrimPlane && boxes[(size_t)k].alwaysHit)
                {
                    const Prim64& pp = prims[(size_t)k];
                    planeEq.insert(planeEq.end(), { pp.g0, pp.g1, pp.g2, pp.g3 });
                    planeIdx.push_back(k);
                    boxes[(size_t)k].alwaysHit = 2u;
                }
            r->planeTablePrims = planeIdx;
            while (planeIdx.size() % 4)
            {
                planeEq.insert(planeEq.end(), { 0.0f, 0.0f, 0.0f, 0.0f });      // d == 0: IntersectRayPlane's own "no hit"
                planeIdx.push_back(0);
            }
        }
        const size_t offBoxes = arena.add(boxes.data(), boxes.size());
        const size_t offPlaneEq = arena.add(planeEq.data(), planeEq.size());
        const size_t offPlaneIdx = arena.add(planeIdx.data(), planeIdx.size());
        arena.bytes.resize((arena.bytes.size() + 127) & ~size_t(127), 0);

        unsigned char* arenaDev = r->sceneMem.upload(arena.bytes.data(), arena.bytes.size());
        if (arenaDev)
        {
            // small meshes: offsets -> device pointers, in the host image of the mesh table, then upload once more
            DevMesh* hm = reinterpret_cast<DevMesh*>(&arena.bytes[offMeshes]);
            for (size_t m = 0; m < meshes.size(); ++m)
            {
                if (hm[m].inArena)
                {
                    hm[m].nodes = reinterpret_cast<const Node64*>(arenaDev + hm[m].offNodes);
                    hm[m].tris = reinterpret_cast<const Tri48*>(arenaDev + hm[m].offTris);
                    hm[m].normals = reinterpret_cast<const float*>(arenaDev + hm[m].offNormals);
                    hm[m].cdf = reinterpret_cast<const float*>(arenaDev + hm[m].offCdf);
                }
            }
            if (!meshes.empty() && hipMemcpy(arenaDev + offMeshes, hm, sizeof(DevMesh)*meshes.size(), hipMemcpyHostToDevice) != hipSuccess)
                arenaDev = nullptr;
            r->meshesRef.assign(hm, hm + meshes.size());
            r->meshesNow = r->meshesRef;
        }
        if (!arenaDev)
        {
            fail("create: device allocation failed (scene arena)");
            ok = false;
        }
        else
        {
            sc.arena = arenaDev;
            sc.arenaBytes = (uint32_t)arena.bytes.size();
            const size_t ldsLimit = getenv("TINSEL_HIP_ARENA_LDS_LIMIT") ? (size_t)atoll(getenv("TINSEL_HIP_ARENA_LDS_LIMIT")) : kArenaLdsLimit;
            sc.arenaLdsBytes = (arena.bytes.size() <= ldsLimit && !getenv("TINSEL_HIP_NO_LDS_SCENE")) ? sc.arenaBytes : 0u;
            sc.nodes = reinterpret_cast<const Node64*>(arenaDev + offNodes);
            sc.prims = reinterpret_cast<const Prim64*>(arenaDev + offPrims);
            sc.mats = reinterpret_cast<const Mat128*>(arenaDev + offMats);
            sc.moving = reinterpret_cast<const Moving64*>(arenaDev + offMoving);
            sc.lights = reinterpret_cast<const int32_t*>(arenaDev + offLights);
            sc.meshes = reinterpret_cast<const DevMesh*>(arenaDev + offMeshes);
            sc.numMeshes = (int)meshes.size();
            r->sceneStackNeed = sceneBvh.maxLeafDepth + 1;
            sc.primBoxes = reinterpret_cast<const PrimBox*>(arenaDev + offBoxes);
            sc.planeEq = reinterpret_cast<const float4*>(arenaDev + offPlaneEq);
            sc.planeIdx = reinterpret_cast<const int32_t*>(arenaDev + offPlaneIdx);
            sc.numPlanes = (int32_t)r->planeTablePrims.size();
            r->sceneBvhHost.assign(desc->bvh_nodes, desc->bvh_nodes + desc->num_bvh_nodes);
            r->arenaOffNodes = offNodes;
            r->arenaOffBoxes = offBoxes;
            r->arenaOffPrims = offPrims;
            r->arenaOffMoving = offMoving;
            r->arenaOffMats = offMats;
            r->primsHost = prims;
            sc.hasMedia = 0;
            for (const Mat128& mm : mats)
                if (mm.absorption[0] != 0.0f || mm.absorption[1] != 0.0f || mm.absorption[2] != 0.0f)
                    sc.hasMedia = 1;
            sc.flatScan = flatScan ? 1 : 0;
            {
                int meshPrimCount = 0;
                for (int k = 0; k < P; ++k)
                    meshPrimCount += prims[(size_t)k].type == kPrimMesh ? 1 : 0;
                sc.deferMeshes = (meshPrimCount >= 2) ? 1 : 0;
            }
            // Fused kernel: sort the next bounce's queue by "meets the box of a bounded primitive" (tn_isect.h) when the
            // scene is open.  Measured (cornell-sized frames, fused kernel): env_loft (1 plane) +16 %, gloss (1 plane) +4 %;
            // the closed boxes cornell / cornell+probe (5 planes, every NEE ray aimed at the light mesh) -4 %: the test and
            // the second append cost more than the plane-only waves save.
            {
                int planes = 0;
                for (int k = 0; k < P; ++k)
                    planes += boxes[(size_t)k].alwaysHit ? 1 : 0;
                sc.sortQueues = (sc.flatScan && planes <= 2 && planes < P) ? 1 : 0;
            }
            // two infinite planes with opposite normals (a floor and a ceiling): every ray between them that is not parallel to
            // them hits one -- a scene no ray leaves, whatever else is in it (cornell.tin, glass.tin)
            for (int i = 0; i < P && !r->sceneEnclosed; ++i)
                for (int j = i + 1; j < P && !r->sceneEnclosed; ++j)
                    if (prims[(size_t)i].type == kPrimPlane && prims[(size_t)j].type == kPrimPlane)
                    {
                        const Prim64 &a = prims[(size_t)i], &b = prims[(size_t)j];
                        const float d = a.g0*b.g0 + a.g1*b.g1 + a.g2*b.g2;
                        const float la = sqrtf(a.g0*a.g0 + a.g1*a.g1 + a.g2*a.g2), lb = sqrtf(b.g0*b.g0 + b.g1*b.g1 + b.g2*b.g2);
                        r->sceneEnclosed = la > 0.0f && lb > 0.0f && d < -0.99f*la*lb;
                    }
            bool all = sc.arenaLdsBytes != 0;
            for (const DevMesh& dmesh : meshes)
                all = all && dmesh.inArena;
            sc.allInArena = all ? 1 : 0;
        }
        sc.root = sceneBvh.root;
        sc.numPrims = P;
        sc.numLights = (int)lights.size();
        sc.horizon[0] = desc->sky_horizon.x; sc.horizon[1] = desc->sky_horizon.y; sc.horizon[2] = desc->sky_horizon.z;
        sc.zenith[0] = desc->sky_zenith.x; sc.zenith[1] = desc->sky_zenith.y; sc.zenith[2] = desc->sky_zenith.z;

    }

    if (ok && desc->probe_valid)
    {
        const size_t n = (size_t)desc->probe_width*desc->probe_height;
        if (!desc->probe_data || !desc->probe_pdf_x || !desc->probe_cdf_x || !desc->probe_pdf_y || !desc->probe_cdf_y || n == 0)
        {
            fail("create: probe marked valid but arrays missing");
            ok = false;
        }
        else
        {
            sc.probe.data = (const float4*)r->sceneMem.upload((const float*)desc->probe_data, n*4);
            sc.probe.pdfX = r->sceneMem.upload(desc->probe_pdf_x, n);
            sc.probe.cdfX = r->sceneMem.upload(desc->probe_cdf_x, n);
            sc.probe.pdfY = r->sceneMem.upload(desc->probe_pdf_y, (size_t)desc->probe_height);
            sc.probe.cdfY = r->sceneMem.upload(desc->probe_cdf_y, (size_t)desc->probe_height);
            sc.probe.width = desc->probe_width;
            sc.probe.height = desc->probe_height;
            sc.probe.valid = 1;
            if (!sc.probe.data || !sc.probe.pdfX || !sc.probe.cdfX || !sc.probe.pdfY || !sc.probe.cdfY)
            {
                fail("create: device allocation failed (probe)");
                ok = false;
            }
        }
    }

    if (ok)
    {
        r->neePerPath = totalLightSamples + (sc.probe.valid ? 1 : 0);
        sc.totalLightSamples = r->neePerPath;
        if (hipMalloc((void**)&r->statsDev, sizeof(unsigned long long)*kStatShards*kStatWords) != hipSuccess ||
            hipMemset(r->statsDev, 0, sizeof(unsigned long long)*kStatShards*kStatWords) != hipSuccess)
        {
            fail("create: device allocation failed (stats)");
            ok = false;
        }
    }

    // the traversal stacks (+ the staged arena) must fit a workgroup's LDS: give the arena up first, then refuse
    if (ok && stack_bytes(r) > (size_t)r->sharedMemLimit && r->scene.arenaLdsBytes)
    {
        r->scene.arenaLdsBytes = 0;
        r->scene.allInArena = 0;
    }
    if (ok && stack_bytes(r) > (size_t)r->sharedMemLimit)
    {
        fail("create: the traversal stacks of this scene need " + std::to_string(stack_bytes(r)) + " B of LDS per workgroup, the device offers " + std::to_string(r->sharedMemLimit));
        ok = false;
    }

    if (!ok)
    {
        r->sceneMem.release();
        if (r->statsDev) (void)hipFree(r->statsDev);
        if (r->walkProf) (void)hipFree(r->walkProf);
        delete r;
        return nullptr;
    }
    return r;
}

void tinsel_hip_destroy(tinsel_hip* r)
{
    if (!r)
        return;
    (void)hipSetDevice(r->device);
    lookahead_release(r);
    (void)hipDeviceSynchronize();
    if (r->laneStream) (void)hipStreamDestroy(r->laneStream);
    if (r->laneFork) (void)hipEventDestroy(r->laneFork);
    if (r->laneJoin) (void)hipEventDestroy(r->laneJoin);
    for (int k = 0; k < 2; ++k)
        if (r->accDone[k]) (void)hipEventDestroy(r->accDone[k]);
    if (r->workStream) (void)hipStreamDestroy(r->workStream);
    if (r->copyStream) (void)hipStreamDestroy(r->copyStream);
    if (r->probeAlias) (void)hipFree(r->probeAlias);
    if (r->walkOverflow) (void)hipFree(r->walkOverflow);
    if (r->laneB.walkOverflow) (void)hipFree(r->laneB.walkOverflow);
    if (r->walkProf)
    {
        unsigned long long wp[16] = { 0 };
        (void)hipMemcpy(wp, r->walkProf, sizeof(wp), hipMemcpyDeviceToHost);
        const double tot = (double)(wp[0] + wp[1] + wp[2] + wp[3] + wp[4]);
        fprintf(stderr, "k_walk profile: cycles refill %.1f%% node %.1f%% tri %.1f%% pop %.1f%% loop %.1f%% | iterations %llu refills %llu node-phases %llu tri-phases %llu | "
                "lanes/node-phase %.1f lanes/tri-phase %.1f lanes/refill %.1f | waves %llu cycles/wave %.0f cycles/iteration %.0f cycles/refill %.0f cycles/node-phase %.0f cycles/tri-phase %.0f\n",
                100.0*wp[0]/tot, 100.0*wp[1]/tot, 100.0*wp[2]/tot, 100.0*wp[3]/tot, 100.0*wp[4]/tot, wp[5], wp[6], wp[7], wp[8],
                (double)wp[9]/std::max(1ull, wp[7]), (double)wp[10]/std::max(1ull, wp[8]), (double)wp[11]/std::max(1ull, wp[6]),
                wp[12], (double)wp[13]/std::max(1ull, wp[12]), tot/std::max(1ull, wp[5]), (double)wp[0]/std::max(1ull, wp[6]),
                (double)wp[1]/std::max(1ull, wp[7]), (double)wp[2]/std::max(1ull, wp[8]));
        (void)hipFree(r->walkProf);
    }
    free_batch(r);
    r->sceneMem.release();
    if (r->accum && r->accumOwned) (void)hipFree(r->accum);
    for (float4* d : r->display)
        if (d) (void)hipFree(d);
    for (void* p : r->lbvhAllocs)
        (void)hipFree(p);
    if (r->accTilesDev) (void)hipFree(r->accTilesDev);
    if (r->passSeedsDev) (void)hipFree(r->passSeedsDev);
    if (r->passSeedsReady) (void)hipEventDestroy(r->passSeedsReady);
    if (r->statsDev) (void)hipFree(r->statsDev);
    for (TimedSpan& s : r->spans)
    {
        (void)hipEventDestroy(s.start);
        (void)hipEventDestroy(s.stop);
    }
    for (hipEvent_t e : r->eventPool)
        (void)hipEventDestroy(e);
    delete r;
}

int tinsel_hip_init(tinsel_hip* r, int width, int height)
{
    if (r)
        lookahead_release(r);
    if (!r || width <= 0 || height <= 0)
        return fail("init: bad arguments");
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipDeviceSynchronize());
    if (r->accum && r->accumOwned)
        (void)hipFree(r->accum);
    r->accum = nullptr;
    r->accumOwned = true;
    HIP_TRY(hipMalloc((void**)&r->accum, sizeof(float4)*(size_t)width*height));
    HIP_TRY(hipMemset(r->accum, 0, sizeof(float4)*(size_t)width*height));
    HIP_TRY(hipStreamSynchronize(nullptr));     // before anything is accumulated on another (non-blocking) stream: see ensure_batch
    r->width = width;
    r->height = height;
    return 0;
}

int tinsel_hip_init_external(tinsel_hip* r, int width, int height, float* device_accum)
{
    if (r)
        lookahead_release(r);
    if (!r || width <= 0 || height <= 0 || !device_accum)
        return fail("init_external: bad arguments");
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipDeviceSynchronize());
    if (r->accum && r->accumOwned)
        (void)hipFree(r->accum);
    r->accum = (float4*)device_accum;
    r->accumOwned = false;
    HIP_TRY(hipMemset(r->accum, 0, sizeof(float4)*(size_t)width*height));
    HIP_TRY(hipStreamSynchronize(nullptr));
    r->width = width;
    r->height = height;
    return 0;
}

int tinsel_hip_render_async(tinsel_hip* r, const tinsel_camera* camera, const tinsel_options* options, int passes, void* stream)
{
    lookahead_cancel(r);
    return render_impl(r, camera, options, passes, (hipStream_t)stream);
}

int tinsel_hip_set_lookahead(tinsel_hip* r, int enable)
{
    if (!r)
        return fail("set_lookahead: null");
    if (!enable)
        lookahead_cancel(r);
    if (enable != TINSEL_LOOKAHEAD_PIN_OUTPUT && r->pinnedPtr)
    {
        (void)hipSetDevice(r->device);
        if (r->copyStream)
            (void)hipStreamSynchronize(r->copyStream);
        (void)hipHostUnregister(r->pinnedPtr);
        r->pinnedPtr = nullptr;
        r->pinnedBytes = 0;
    }
    r->lookahead = enable == TINSEL_LOOKAHEAD_PIN_OUTPUT ? TINSEL_LOOKAHEAD_PIN_OUTPUT : (enable ? TINSEL_LOOKAHEAD_ON : TINSEL_LOOKAHEAD_OFF);
    return 0;
}

int tinsel_hip_render(tinsel_hip* r, const tinsel_camera* camera, const tinsel_options* options, float* out_rgba, int passes)
{
    if (r && r->lookahead && out_rgba && camera && options && r->accum && r->accumOwned && passes >= 1 &&
        options->width == r->width && options->height == r->height)
        return lookahead_render(r, camera, options, out_rgba, passes);
    lookahead_cancel(r);
    if (render_impl(r, camera, options, passes, nullptr))
        return -1;
    if (out_rgba)
        return tinsel_hip_read_accum(r, out_rgba);
    HIP_TRY(hipStreamSynchronize(nullptr));
    return 0;
}

float* tinsel_hip_accum_device_ptr(tinsel_hip* r) { return r ? (float*)r->accum : nullptr; }

int tinsel_hip_read_accum(tinsel_hip* r, float* out_rgba)
{
    if (!r || !r->accum || !out_rgba)
        return fail("read_accum: bad arguments");
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out_rgba, r->accum, sizeof(float4)*(size_t)r->width*r->height, hipMemcpyDeviceToHost));
    return 0;
}

// The display stage of the reference's frame loop (main.cpp:258-282) on the device accumulator.
int tinsel_hip_present_async(tinsel_hip* r, const tinsel_options* options, int nlm_width, float nlm_falloff, void* stream)
{
    if (!r || !r->accum || !options || nlm_width < 0)
        return fail("present: bad arguments (Init and Render first)");
    HIP_TRY(hipSetDevice(r->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)r->width*r->height;
    if (options->mode != TINSEL_MODE_PATHTRACE)
    {
        r->presented = r->accum;        // main.cpp:258: the other modes present the raw pixels
        return 0;
    }
    if (r->displayPixels != n)
    {
        HIP_TRY(hipDeviceSynchronize());
        for (float4*& d : r->display)
        {
            if (d) (void)hipFree(d);
            d = nullptr;
        }
        r->displayPixels = 0;
    }
    const int needed = nlm_width ? 3 : 1;
    for (int i = 0; i < needed; ++i)
        if (!r->display[i])
            HIP_TRY(hipMalloc((void**)&r->display[i], sizeof(float4)*n));
    r->displayPixels = n;

    {
        ScopedTimer t(r, KN_PRESENT, st);
        hipLaunchKernelGGL(k_present, dim3((unsigned)((n + 255)/256)), dim3(256), 0, st, r->accum, r->display[0], (int)n,
                           options->exposure, options->limit);
    }
    r->presented = r->display[0];
    if (nlm_width)
    {
        const dim3 grid((r->width + 15)/16, (r->height + 15)/16);
        {
            ScopedTimer t(r, KN_NLM_MEANS, st);
            hipLaunchKernelGGL(k_nlm_means, grid, dim3(256), 0, st, r->display[0], r->display[1], r->width, r->height, nlm_width);
        }
        {
            ScopedTimer t(r, KN_NLM, st);
            hipLaunchKernelGGL(k_nlm, grid, dim3(256), 0, st, r->display[0], r->display[1], r->display[2], r->width, r->height,
                               nlm_falloff, nlm_width);
        }
        r->presented = r->display[2];
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int tinsel_hip_present(tinsel_hip* r, const tinsel_options* options, int nlm_width, float nlm_falloff, float* out_rgba)
{
    if (tinsel_hip_present_async(r, options, nlm_width, nlm_falloff, nullptr))
        return -1;
    HIP_TRY(hipStreamSynchronize(nullptr));
    if (out_rgba)
        HIP_TRY(hipMemcpy(out_rgba, r->presented, sizeof(float4)*(size_t)r->width*r->height, hipMemcpyDeviceToHost));
    return 0;
}

const float* tinsel_hip_present_device_ptr(tinsel_hip* r) { return r ? (const float*)r->presented : nullptr; }

// WritePng's 8-bit quantisation (png.cpp:323-343): one serial default-seeded Random stream dithers every
// channel (two Randf per channel), all in double until the narrowing at the Quantize(float) call.  Host code:
// the generator is a nonlinear recurrence (no skip-ahead), 6 draws per pixel.
int tinsel_image_quantize_rgb8(const float* rgba, int width, int height, unsigned char* rgb)
{
    if (!rgba || !rgb || width <= 0 || height <= 0)
        return fail("quantize: bad arguments");
    Rng rand = Rng::seeded(0u);
    const size_t n = (size_t)width*height;
    for (size_t i = 0; i < n; ++i)
    {
        for (int c = 0; c < 3; ++c)
        {
            const double a = (double)rgba[i*4 + c]*255.0;
            const float r1 = rand.randf();
            const float r2 = rand.randf();
            const float x = (float)(((a + (double)r1) + (double)r2) - (double)0.5f);
            // Clamp = Min(Max(x, 0), 255) with Max(a,b) = (a < b) ? b : a, Min(a,b) = (a < b) ? a : b  (maths.h:55-64)
            const float lo = (x < 0.0f) ? 0.0f : x;
            const float cl = (lo < 255.0f) ? lo : 255.0f;
            rgb[i*3 + c] = (unsigned char)(int)cl;
        }
    }
    return 0;
}

int tinsel_hip_set_mesh_bvh(tinsel_hip* r, int mode, double* build_ms)
{
    lookahead_cancel(r);
    if (!r || (mode != TINSEL_BVH_REFERENCE && mode != TINSEL_BVH_LBVH && mode != TINSEL_BVH_PLOC))
        return fail("set_mesh_bvh: bad arguments");
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipDeviceSynchronize());
    if (build_ms)
        *build_ms = 0.0;

    std::vector<DevMesh> next = r->meshesRef;
    const size_t prevAllocs = r->lbvhAllocs.size();
    if (mode != TINSEL_BVH_REFERENCE)
    {
        hipEvent_t e0, e1;
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        (void)hipEventRecord(e0, nullptr);
        int rc = 0;
        for (size_t m = 0; m < next.size() && !rc; ++m)
            if (!next[m].inArena && next[m].numTris >= 2)       // LDS-resident meshes keep their (tiny) reference trees
                rc = build_device_bvh(r, r->meshesRef[m], next[m], mode);
        (void)hipEventRecord(e1, nullptr);
        (void)hipEventSynchronize(e1);
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        if (rc)
            return -1;
        if (build_ms)
            *build_ms = ms;
    }

    int maxMeshNeed = 0;
    for (const DevMesh& dm : next)
        if (dm.stackNeed > maxMeshNeed)
            maxMeshNeed = dm.stackNeed;
    const int stack = pick_stack(r->sceneStackNeed + maxMeshNeed);
    const size_t ldsNeed = stack < 0 ? 0 : ((size_t)stack*kBlock + kScanWords)*sizeof(uint32_t) + r->scene.arenaLdsBytes;
    if (stack < 0 || ldsNeed > (size_t)r->sharedMemLimit)
    {
        // keep what was there: drop the trees just built
        for (size_t k = prevAllocs; k < r->lbvhAllocs.size(); ++k)
            (void)hipFree(r->lbvhAllocs[k]);
        r->lbvhAllocs.resize(prevAllocs);
        return fail("set_mesh_bvh: tree too deep for the LDS traversal stack (previous trees kept)");
    }
    if (!next.empty())
        HIP_TRY(hipMemcpy((void*)r->scene.meshes, next.data(), sizeof(DevMesh)*next.size(), hipMemcpyHostToDevice));
    r->meshesNow = next;
    r->stackNeed = stack;
    r->bvhMode = mode;
    // the previous generation of device-built trees is unreachable now (a per-frame rebuild must not grow)
    for (size_t k = 0; k < prevAllocs; ++k)
        (void)hipFree(r->lbvhAllocs[k]);
    r->lbvhAllocs.erase(r->lbvhAllocs.begin(), r->lbvhAllocs.begin() + (long)prevAllocs);
    return 0;
}

// Refit of a deforming mesh: new vertex positions (and optionally normals), same topology, same tree shape.
int tinsel_hip_refit_mesh(tinsel_hip* r, int primitive, const float* positions_xyz, int num_vertices, const float* normals_xyz)
{
    lookahead_cancel(r);
    if (!r || !positions_xyz || primitive < 0 || primitive >= r->scene.numPrims || r->primMesh[(size_t)primitive] < 0)
        return fail("refit_mesh: bad arguments (a mesh primitive and its new positions)");
    const int mi = r->primMesh[(size_t)primitive];
    DevMesh& dm = r->meshesNow[(size_t)mi];
    // (a mesh of the LDS-staged arena is refitted in the arena's copy in HBM, which every launch stages from)
    if (num_vertices != r->meshNumVertices[(size_t)mi])
        return fail("refit_mesh: the topology must not change (vertex count differs)");
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipDeviceSynchronize());

    const int numTris = dm.numTris;
    const int numNodes = numTris - 1;           // one triangle per leaf: internal nodes
    float* posDev = nullptr;
    float* own = nullptr;
    int* gen = nullptr;
    int rc = 0;
    do {
        if (hipMalloc((void**)&posDev, sizeof(float)*3*(size_t)num_vertices) != hipSuccess ||
            hipMemcpy(posDev, positions_xyz, sizeof(float)*3*(size_t)num_vertices, hipMemcpyHostToDevice) != hipSuccess) { rc = fail("refit_mesh: upload failed"); break; }
        hipLaunchKernelGGL(k_refit_tris, dim3((unsigned)((numTris + 255)/256)), dim3(256), 0, nullptr, const_cast<Tri48*>(dm.tris), numTris, posDev);
        if (normals_xyz && hipMemcpy(const_cast<float*>(dm.normals), normals_xyz, sizeof(float)*3*(size_t)num_vertices, hipMemcpyHostToDevice) != hipSuccess) { rc = fail("refit_mesh: normals upload failed"); break; }
        if (numNodes > 0)
        {
            if (hipMalloc((void**)&own, sizeof(float)*6*(size_t)numNodes) != hipSuccess || hipMalloc((void**)&gen, sizeof(int)*(size_t)numNodes) != hipSuccess)
            { rc = fail("refit_mesh: device allocation failed"); break; }
            // the tree in use and, when a device-built one is, the reference's too (switching back must not find stale boxes)
            const DevMesh* trees[2] = { &dm, r->meshesRef[(size_t)mi].nodes != dm.nodes ? &r->meshesRef[(size_t)mi] : nullptr };
            for (const DevMesh* tree : trees)
            {
                if (!tree || rc)
                    continue;
                if (hipMemset(gen, 0, sizeof(int)*(size_t)numNodes) != hipSuccess) { rc = fail("refit_mesh: memset failed"); break; }
                // the root of a converted tree is node 0 (reference trees: convert_bvh; device-built ones: the Karras root)
                int rootGen = 0;
                for (int pass = 1; pass <= 4096 && !rootGen; )
                {
                    for (int k = 0; k < 16; ++k, ++pass)
                        hipLaunchKernelGGL(k_refit_pass, dim3((unsigned)((numNodes + 255)/256)), dim3(256), 0, nullptr, const_cast<Node64*>(tree->nodes), numNodes, dm.tris, own, gen, pass);
                    if (hipMemcpy(&rootGen, gen + (tree->root & ~kLeafBit), sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
                        break;
                }
                if (!rootGen)
                    rc = fail("refit_mesh: the refit did not reach the root");
            }
            if (rc)
                break;
        }
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { rc = fail("refit_mesh: kernels failed"); break; }
    } while (false);
    if (posDev) (void)hipFree(posDev);
    if (own) (void)hipFree(own);
    if (gen) (void)hipFree(gen);
    if (rc)
        return rc;

    // Mesh::RebuildCDF (mesh.cpp:340-368) in the reference's own serial fp32 order, then PrimitiveArea of every instance
    const std::vector<int32_t>& idx = r->meshIndices[(size_t)mi];
    std::vector<float> cdf((size_t)numTris);
    float totalArea = 0.0f;
    for (int t = 0; t < numTris; ++t)
    {
        const float* a = positions_xyz + (size_t)idx[(size_t)t*3 + 0]*3;
        const float* b = positions_xyz + (size_t)idx[(size_t)t*3 + 1]*3;
        const float* c = positions_xyz + (size_t)idx[(size_t)t*3 + 2]*3;
        const V3 ab(b[0] - a[0], b[1] - a[1], b[2] - a[2]), ac(c[0] - a[0], c[1] - a[1], c[2] - a[2]);
        const float area = 0.5f*length(cross(ab, ac));
        totalArea += area;
        cdf[(size_t)t] = totalArea;
    }
    for (int t = 0; t < numTris; ++t)
        cdf[(size_t)t] /= totalArea;
    HIP_TRY(hipMemcpy(const_cast<float*>(dm.cdf), cdf.data(), sizeof(float)*(size_t)numTris, hipMemcpyHostToDevice));
    for (int p = 0; p < r->scene.numPrims; ++p)
        if (r->primMesh[(size_t)p] == mi)
        {
            const float area = totalArea*r->primEndScale[(size_t)p];          // intersection.h:843-847
            const float rcpArea = 1.0f/area;
            HIP_TRY(hipMemcpy((void*)&r->scene.mats[p].area, &area, sizeof(float), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy((void*)&r->scene.mats[p].rcpArea, &rcpArea, sizeof(float), hipMemcpyHostToDevice));
        }

    // The scene level follows: Scene::Build (scene.cpp:4-16) gives the scene BVH builder PrimitiveBounds(p) (intersection.h:906-939)
    // = the mesh root's box under the start and end transforms.  The tree keeps its shape here too: the leaf box of every
    // instance is recomputed with the reference's expressions and its ancestors become the union of their children (what the
    // builder stores for that shape: min and max do not round).  Without this a deformation that leaves the old box is
    // clipped: the flat scan, the queue sort and k_walk's `enters` test all start from the leaf box.
    V3 lo(kFltMax, kFltMax, kFltMax), hi(-kFltMax, -kFltMax, -kFltMax);     // the root's box: the union of the triangles' (Bounds::AddPoint)
    for (size_t k = 0; k < (size_t)numTris*3; ++k)
    {
        const float* v = positions_xyz + (size_t)idx[k]*3;
        lo = V3(minT(lo.x, v[0]), minT(lo.y, v[1]), minT(lo.z, v[2]));
        hi = V3(maxT(hi.x, v[0]), maxT(hi.y, v[1]), maxT(hi.z, v[2]));
    }
    r->meshRootLo[(size_t)mi] = lo;
    r->meshRootHi[(size_t)mi] = hi;
    r->meshArea[(size_t)mi] = totalArea;
    std::vector<tinsel_bvh_node>& sb = r->sceneBvhHost;
    std::vector<int> leafOf((size_t)r->scene.numPrims, -1);
    for (size_t k = 0; k < sb.size(); ++k)
        if (ref_is_leaf(sb[k]) && sb[k].left_index < (uint32_t)r->scene.numPrims)
            leafOf[sb[k].left_index] = (int)k;
    std::vector<PrimBox> newBoxes;
    std::vector<int> newBoxPrim;
    for (int p = 0; p < r->scene.numPrims; ++p)
    {
        if (r->primMesh[(size_t)p] != mi || leafOf[(size_t)p] < 0)
            continue;
        V3 sl, su, el, eu;
        transform_bounds(r->primStart[(size_t)p], lo, hi, sl, su);
        transform_bounds(r->primEnd[(size_t)p], lo, hi, el, eu);
        tinsel_bvh_node& leaf = sb[(size_t)leafOf[(size_t)p]];
        leaf.lower.x = minT(sl.x, el.x); leaf.lower.y = minT(sl.y, el.y); leaf.lower.z = minT(sl.z, el.z);      // Union, maths.h:1023-1026
        leaf.upper.x = maxT(su.x, eu.x); leaf.upper.y = maxT(su.y, eu.y); leaf.upper.z = maxT(su.z, eu.z);
        newBoxes.push_back(make_prim_box(leaf));
        newBoxPrim.push_back(p);
    }
    {
        // ancestors: post-order over the reference's tree (validated acyclic by convert_bvh at create)
        std::vector<uint32_t> order, stack(1, 0u);
        while (!stack.empty())
        {
            const uint32_t k = stack.back();
            stack.pop_back();
            order.push_back(k);
            if (!ref_is_leaf(sb[k]))
            {
                stack.push_back(sb[k].left_index);
                stack.push_back(ref_right(sb[k]));
            }
        }
        for (size_t q = order.size(); q-- > 0; )
        {
            tinsel_bvh_node& n = sb[order[q]];
            if (ref_is_leaf(n))
                continue;
            const tinsel_bvh_node& a = sb[n.left_index];
            const tinsel_bvh_node& b = sb[ref_right(n)];
            n.lower.x = minT(a.lower.x, b.lower.x); n.lower.y = minT(a.lower.y, b.lower.y); n.lower.z = minT(a.lower.z, b.lower.z);
            n.upper.x = maxT(a.upper.x, b.upper.x); n.upper.y = maxT(a.upper.y, b.upper.y); n.upper.z = maxT(a.upper.z, b.upper.z);
        }
    }
    ConvertedBvh sceneBvh;
    if (!convert_bvh(sb.data(), (int)sb.size(), r->scene.numPrims, 0, sceneBvh))
        return fail("refit_mesh: the scene BVH could not be refitted");
    unsigned char* arenaDev = const_cast<unsigned char*>(r->scene.arena);
    if (!sceneBvh.nodes.empty())
        HIP_TRY(hipMemcpy(arenaDev + r->arenaOffNodes, sceneBvh.nodes.data(), sizeof(Node64)*sceneBvh.nodes.size(), hipMemcpyHostToDevice));
    for (size_t k = 0; k < newBoxes.size(); ++k)
        HIP_TRY(hipMemcpy(arenaDev + r->arenaOffBoxes + sizeof(PrimBox)*(size_t)newBoxPrim[k], &newBoxes[k], sizeof(PrimBox), hipMemcpyHostToDevice));
    return 0;
}

// A primitive moves (the reference mutates Scene::primitives[i].startTransform / endTransform and re-runs Scene::Build).
int tinsel_hip_set_primitive_transform(tinsel_hip* r, int index, const tinsel_transform* start, const tinsel_transform* end)
{
    lookahead_cancel(r);
    if (!r || !start || !end || index < 0 || index >= r->scene.numPrims)
        return fail("set_primitive_transform: bad arguments");
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipDeviceSynchronize());
    const Xform xs = to_xform(*start), xe = to_xform(*end);
    const bool isStatic = memcmp(start, end, sizeof(tinsel_transform)) == 0;
    r->primStart[(size_t)index] = xs;
    r->primEnd[(size_t)index] = xe;
    r->primEndScale[(size_t)index] = xe.s;
    Prim64& o = r->primsHost[(size_t)index];
    set_prim_pose(o, xs, xe, isStatic);
    set_prim_derived(o);
    const Moving64 mv = make_moving(xs, xe);
    unsigned char* arenaDev = const_cast<unsigned char*>(r->scene.arena);
    HIP_TRY(hipMemcpy(arenaDev + r->arenaOffPrims + sizeof(Prim64)*(size_t)index, &o, sizeof(Prim64), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(arenaDev + r->arenaOffMoving + sizeof(Moving64)*(size_t)index, &mv, sizeof(Moving64), hipMemcpyHostToDevice));
    if (o.type == kPrimMesh)
    {
        // PrimitiveArea of a mesh: area*endTransform.s (intersection.h:843-847)
        const float area = r->meshArea[(size_t)r->primMesh[(size_t)index]]*xe.s;
        const float rcpArea = 1.0f/area;
        HIP_TRY(hipMemcpy(arenaDev + r->arenaOffMats + sizeof(Mat128)*(size_t)index + offsetof(Mat128, area), &area, sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(arenaDev + r->arenaOffMats + sizeof(Mat128)*(size_t)index + offsetof(Mat128, rcpArea), &rcpArea, sizeof(float), hipMemcpyHostToDevice));
    }
    r->sceneDirty = true;
    return 0;
}

namespace {

// PrimitiveBounds (intersection.h:906-939) of primitive i as it is now: the local box (sphere: +-radius; plane: +-1e8; mesh: its root's)
// under the start and the end transform, TransformBounds (maths.h:1004-1021), united
void primitive_bounds(const tinsel_hip* r, int i, V3& lower, V3& upper)
{
    const Prim64& p = r->primsHost[(size_t)i];
    V3 lo, hi;
    if (p.type == kPrimSphere)      { lo = V3(-p.g0); hi = V3(p.g0); }
    else if (p.type == kPrimPlane)  { lo = V3(-1.e+8f); hi = V3(1.e+8f); }
    else                            { lo = r->meshRootLo[(size_t)r->primMesh[(size_t)i]]; hi = r->meshRootHi[(size_t)r->primMesh[(size_t)i]]; }
    V3 sl, su, el, eu;
    transform_bounds(r->primStart[(size_t)i], lo, hi, sl, su);
    transform_bounds(r->primEnd[(size_t)i], lo, hi, el, eu);
    lower = V3(minT(sl.x, el.x), minT(sl.y, el.y), minT(sl.z, el.z));      // Union, maths.h:1023-1026
    upper = V3(maxT(su.x, eu.x), maxT(su.y, eu.y), maxT(su.z, eu.z));
}

// a Node64 tree (as the device builders emit it) back into the reference's node array: what tinsel_hip_refit_mesh walks to refit
// the scene level, and what a later TINSEL_SCENE_BVH_NODES caller would hand in
void node64_to_reference(const std::vector<Node64>& nodes, uint32_t ref, float lminx, float lminy, float lminz, float lmaxx, float lmaxy, float lmaxz,
                         std::vector<tinsel_bvh_node>& out, uint32_t at)
{
    tinsel_bvh_node& me = out[at];
    me.lower.x = lminx; me.lower.y = lminy; me.lower.z = lminz;
    me.upper.x = lmaxx; me.upper.y = lmaxy; me.upper.z = lmaxz;
    if (ref & kLeafBit)
    {
        me.left_index = ref & ~kLeafBit;
        me.right_index_leaf = 0x80000000u;
        return;
    }
    const Node64 n = nodes[ref];
    const uint32_t l = (uint32_t)out.size();
    out.push_back(tinsel_bvh_node());
    out.push_back(tinsel_bvh_node());
    out[at].left_index = l;
    out[at].right_index_leaf = l + 1u;
    node64_to_reference(nodes, n.left, n.lminx, n.lminy, n.lminz, n.lmaxx, n.lmaxy, n.lmaxz, out, l);
    node64_to_reference(nodes, n.right, n.rminx, n.rminy, n.rminz, n.rmaxx, n.rmaxy, n.rmaxz, out, l + 1u);
}

} // namespace

int tinsel_hip_rebuild_scene(tinsel_hip* r, int mode, const tinsel_bvh_node* nodes, int num_nodes, double* build_ms)
{
    lookahead_cancel(r);
    if (!r || (mode != TINSEL_SCENE_BVH_NODES && mode != TINSEL_SCENE_BVH_DEVICE) || (mode == TINSEL_SCENE_BVH_NODES && (!nodes || num_nodes <= 0)))
        return fail("rebuild_scene: bad arguments");
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipDeviceSynchronize());
    if (build_ms)
        *build_ms = 0.0;
    const int P = r->scene.numPrims;

    std::vector<tinsel_bvh_node> ref;       // the new tree in the reference's format
    if (mode == TINSEL_SCENE_BVH_NODES)
        ref.assign(nodes, nodes + num_nodes);
    else
    {
        // leaf boxes: PrimitiveBounds of every primitive as it is now
        std::vector<V3> lo((size_t)P), hi((size_t)P);
        for (int i = 0; i < P; ++i)
            primitive_bounds(r, i, lo[(size_t)i], hi[(size_t)i]);
        if (P == 1)
        {
            ref.resize(1);
            ref[0].lower.x = lo[0].x; ref[0].lower.y = lo[0].y; ref[0].lower.z = lo[0].z;
            ref[0].upper.x = hi[0].x; ref[0].upper.y = hi[0].y; ref[0].upper.z = hi[0].z;
            ref[0].left_index = 0;
            ref[0].right_index_leaf = 0x80000000u;
        }
        else
        {
            // The mesh builders' kernels over the primitives' boxes: a box travels as a degenerate triangle record (a = c = lower, b = upper),
            // whose min / max IS the box; Morton order of the centroids, agglomerative clustering by surface area (tn_lbvh.h)
            std::vector<Tri48> items((size_t)P);
            for (int i = 0; i < P; ++i)
            {
                Tri48& T = items[(size_t)i];
                T.ax = lo[(size_t)i].x; T.ay = lo[(size_t)i].y; T.az = lo[(size_t)i].z; T.i0 = i;
                T.bx = hi[(size_t)i].x; T.by = hi[(size_t)i].y; T.bz = hi[(size_t)i].z; T.i1 = i;
                T.cx = lo[(size_t)i].x; T.cy = lo[(size_t)i].y; T.cz = lo[(size_t)i].z; T.i2 = i;
            }
            Tri48* itemsDev = nullptr;
            HIP_TRY(hipMalloc((void**)&itemsDev, sizeof(Tri48)*(size_t)P));
            int rc = 0;
            std::vector<Node64> built((size_t)P - 1);
            DevMesh fake, out;
            memset(&fake, 0, sizeof(fake));
            fake.tris = itemsDev;
            fake.numTris = P;
            fake.inArena = 1;           // (no bottom-level records for this one)
            const size_t allocsBefore = r->lbvhAllocs.size();
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (hipMemcpy(itemsDev, items.data(), sizeof(Tri48)*(size_t)P, hipMemcpyHostToDevice) != hipSuccess ||
                hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
                rc = fail("rebuild_scene: upload failed");
            if (!rc)
            {
                (void)hipEventRecord(e0, nullptr);
                rc = build_device_bvh(r, fake, out, TINSEL_BVH_PLOC);
                (void)hipEventRecord(e1, nullptr);
                (void)hipEventSynchronize(e1);
                float ms = 0.0f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (build_ms)
                    *build_ms = ms;
            }
            if (!rc && hipMemcpy(built.data(), out.nodes, sizeof(Node64)*((size_t)P - 1), hipMemcpyDeviceToHost) != hipSuccess)
                rc = fail("rebuild_scene: read-back failed");
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
            // the builder's own allocation for this tree: the arena takes a copy
            for (size_t k = allocsBefore; k < r->lbvhAllocs.size(); ++k)
                (void)hipFree(r->lbvhAllocs[k]);
            r->lbvhAllocs.resize(allocsBefore);
            (void)hipFree(itemsDev);
            if (rc)
                return rc;
            // root box: the union of its children's
            const Node64& rt = built[0];
            ref.reserve((size_t)2*P - 1);
            ref.push_back(tinsel_bvh_node());
            node64_to_reference(built, 0u, minT(rt.lminx, rt.rminx), minT(rt.lminy, rt.rminy), minT(rt.lminz, rt.rminz),
                                maxT(rt.lmaxx, rt.rmaxx), maxT(rt.lmaxy, rt.rmaxy), maxT(rt.lmaxz, rt.rmaxz), ref, 0u);
        }
    }

    // from here on as at create: validate, convert, leaf boxes by primitive, stack depth
    ConvertedBvh sceneBvh;
    if (!convert_bvh(ref.data(), (int)ref.size(), P, 0, sceneBvh))
        return fail("rebuild_scene: malformed scene BVH");
    if ((int)sceneBvh.nodes.size() != (P > 1 ? P - 1 : 0))
        return fail("rebuild_scene: the scene BVH must have one leaf per primitive");
    std::vector<PrimBox> boxes((size_t)P);
    std::vector<char> seen((size_t)P, 0);
    for (const tinsel_bvh_node& nd : ref)
        if (ref_is_leaf(nd) && nd.left_index < (uint32_t)P)
        {
            boxes[nd.left_index] = make_prim_box(nd);
            seen[nd.left_index] = 1;
        }
    for (int k = 0; k < P; ++k)
        if (!seen[(size_t)k])
            return fail("rebuild_scene: a primitive has no leaf in the scene BVH");
    int maxMeshNeed = 0;
    for (const DevMesh& dm : r->meshesNow)
        maxMeshNeed = std::max(maxMeshNeed, dm.stackNeed);
    const int stack = pick_stack(sceneBvh.maxLeafDepth + 1 + maxMeshNeed);
    if (stack < 0 || ((size_t)stack*kBlock + kScanWords)*sizeof(uint32_t) + r->scene.arenaLdsBytes > (size_t)r->sharedMemLimit)
        return fail("rebuild_scene: tree too deep for the LDS traversal stack (previous tree kept)");

    unsigned char* arenaDev = const_cast<unsigned char*>(r->scene.arena);
    if (!sceneBvh.nodes.empty())
        HIP_TRY(hipMemcpy(arenaDev + r->arenaOffNodes, sceneBvh.nodes.data(), sizeof(Node64)*sceneBvh.nodes.size(), hipMemcpyHostToDevice));
    // The plane table (flat scan of the split pipeline's kernels: the always-hit planes' equations, tested ahead of the loop) follows the new
    // boxes: a table plane whose leaf box is no longer "infinite" (the primitive was scaled below 0.1, or the caller's tree has a tighter
    // leaf) is box-tested in the loop like everything else -- its table entry becomes d == 0, IntersectRayPlane's own "no hit" -- and one
    // whose box is infinite again gets its equation back (the table used to be written at create only).
    if (!r->planeTablePrims.empty() && r->scene.planeEq)
    {
        std::vector<float> eq(r->planeTablePrims.size()*4, 0.0f);
        for (size_t t = 0; t < r->planeTablePrims.size(); ++t)
        {
            const int32_t k = r->planeTablePrims[t];
            if (boxes[(size_t)k].alwaysHit)
            {
                boxes[(size_t)k].alwaysHit = 2u;
                const Prim64& pp = r->primsHost[(size_t)k];
                eq[t*4 + 0] = pp.g0; eq[t*4 + 1] = pp.g1; eq[t*4 + 2] = pp.g2; eq[t*4 + 3] = pp.g3;
            }
        }
        HIP_TRY(hipMemcpy(const_cast<float4*>(r->scene.planeEq), eq.data(), eq.size()*sizeof(float), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemcpy(arenaDev + r->arenaOffBoxes, boxes.data(), sizeof(PrimBox)*(size_t)P, hipMemcpyHostToDevice));
    r->scene.root = sceneBvh.root;
    r->sceneStackNeed = sceneBvh.maxLeafDepth + 1;
    r->stackNeed = stack;
    r->sceneBvhHost = ref;
    r->sceneDirty = false;
    return 0;
}

// Probe importance sampling: the reference's two binary searches (default, sample-identical) or an alias table.
int tinsel_hip_set_probe_sampling(tinsel_hip* r, int mode)
{
    lookahead_cancel(r);
    if (!r || (mode != TINSEL_PROBE_CDF && mode != TINSEL_PROBE_ALIAS))
        return fail("set_probe_sampling: bad arguments");
    if (mode == TINSEL_PROBE_CDF || !r->scene.probe.valid)
    {
        r->scene.probe.alias = nullptr;
        return 0;
    }
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipDeviceSynchronize());
    if (!r->probeAlias)
    {
        // Vose's alias method over p(row, col) = pdfY[row]*pdfX[row, col] -- the probabilities ProbeSample's two searches
        // realise (probe.h:31-79 BuildCDF) -- in double on the host, once
        const int W = r->scene.probe.width, H = r->scene.probe.height;
        const size_t n = (size_t)W*H;
        std::vector<float> px(n), py((size_t)H);
        HIP_TRY(hipMemcpy(px.data(), r->scene.probe.pdfX, sizeof(float)*n, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(py.data(), r->scene.probe.pdfY, sizeof(float)*(size_t)H, hipMemcpyDeviceToHost));
        std::vector<double> scaled(n);
        double total = 0.0;
        for (int j = 0; j < H; ++j)
            for (int i = 0; i < W; ++i)
            {
                const double p = (double)py[(size_t)j]*(double)px[(size_t)j*W + i];
                scaled[(size_t)j*W + i] = p;
                total += p;
            }
        if (!(total > 0.0))
            return fail("set_probe_sampling: the probe has no energy");
        std::vector<uint32_t> small, large;
        small.reserve(n); large.reserve(n);
        for (size_t k = 0; k < n; ++k)
        {
            scaled[k] = scaled[k]/total*(double)n;
            (scaled[k] < 1.0 ? small : large).push_back((uint32_t)k);
        }
        std::vector<uint2> table(n);
        while (!small.empty() && !large.empty())
        {
            const uint32_t s = small.back(); small.pop_back();
            const uint32_t l = large.back();
            const float keep = (float)scaled[s];
            table[s] = make_uint2(__builtin_bit_cast(uint32_t, keep), l);
            scaled[l] = (scaled[l] + scaled[s]) - 1.0;
            if (scaled[l] < 1.0)
            {
                large.pop_back();
                small.push_back(l);
            }
        }
        const float one = 2.0f;         // r2 <= 1 < 2: always keep
        for (uint32_t k : large) table[k] = make_uint2(__builtin_bit_cast(uint32_t, one), k);
        for (uint32_t k : small) table[k] = make_uint2(__builtin_bit_cast(uint32_t, one), k);
        HIP_TRY(hipMalloc((void**)&r->probeAlias, sizeof(uint2)*n));
        HIP_TRY(hipMemcpy(r->probeAlias, table.data(), sizeof(uint2)*n, hipMemcpyHostToDevice));
    }
    r->scene.probe.alias = r->probeAlias;
    return 0;
}

int tinsel_hip_set_russian_roulette(tinsel_hip* r, int start_bounce)
{
    lookahead_cancel(r);
    if (!r || start_bounce < 0)
        return fail("set_russian_roulette: bad arguments");
    r->rrStart = start_bounce;
    return 0;
}

int tinsel_hip_write_accum(tinsel_hip* r, const float* rgba, uint32_t next_pass_index)
{
    lookahead_cancel(r);
    if (!r || !r->accum || !rgba)
        return fail("write_accum: bad arguments (Init first)");
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(r->accum, rgba, sizeof(float4)*(size_t)r->width*r->height, hipMemcpyHostToDevice));
    r->passIndex = next_pass_index;
    return 0;
}

int tinsel_hip_set_shard(tinsel_hip* r, int rank, int world, int tile)
{
    lookahead_cancel(r);
    if (!r || world < 1 || rank < 0 || rank >= world || tile < 1)
        return fail("set_shard: bad arguments");
    if (rank != r->shardRank || world != r->shardWorld || tile != r->shardTile)
    {
        HIP_TRY(hipSetDevice(r->device));
        HIP_TRY(hipDeviceSynchronize());
        free_batch(r);          // ownership changes: start from clean path buffers
    }
    r->shardRank = rank;
    r->shardWorld = world;
    r->shardTile = tile;
    return 0;
}

int tinsel_hip_set_arithmetic(tinsel_hip* r, int mode)
{
    lookahead_cancel(r);
    if (!r || (mode != TINSEL_ARITH_EXACT && mode != TINSEL_ARITH_FAST))
        return fail("set_arithmetic: bad arguments");
    if (tinsel_fast_launch_args_size() != sizeof(LaunchArgs))
        return fail("set_arithmetic: the two builds of the path kernels disagree on the launch record");
    r->arith = mode;
    return 0;
}

int tinsel_hip_get_arithmetic(tinsel_hip* r) { return r ? r->arith : TINSEL_ARITH_EXACT; }

int tinsel_hip_set_pipeline(tinsel_hip* r, int pipeline)
{
    lookahead_cancel(r);
    if (!r || pipeline < TINSEL_PIPELINE_WAVEFRONT || pipeline > TINSEL_PIPELINE_AUTO)
        return fail("set_pipeline: bad arguments");
    r->pipeline = pipeline;
    return 0;
}

int tinsel_hip_set_pass_index(tinsel_hip* r, uint32_t pass_index)
{
    lookahead_cancel(r);
    if (!r)
        return fail("set_pass_index: null");
    r->passIndex = pass_index;
    return 0;
}

uint32_t tinsel_hip_get_pass_index(tinsel_hip* r) { return r ? r->passIndex : 0; }

static int read_stats(tinsel_hip* r, unsigned long long* out8)
{
    std::vector<unsigned long long> shards((size_t)kStatShards*kStatWords);
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(shards.data(), r->statsDev, sizeof(unsigned long long)*shards.size(), hipMemcpyDeviceToHost));
    for (int w = 0; w < kStatWords; ++w)
        out8[w] = 0;
    for (int b = 0; b < kStatShards; ++b)
        for (int w = 0; w < kStatWords; ++w)
            out8[w] += shards[(size_t)b*kStatWords + w];
    return 0;
}

void tinsel_hip_stats(tinsel_hip* r, unsigned long long* rays, unsigned long long* samples, double* gpu_seconds)
{
    unsigned long long s[8] = { 0 };
    if (r && r->statsDev)
        (void)read_stats(r, s);
    if (rays) *rays = s[0];
    if (samples) *samples = s[1];
    if (gpu_seconds) *gpu_seconds = r ? r->gpuSeconds : 0.0;
}

/* extended counters: [0]=rays [1]=samples [2]=internal node visits [3]=triangle tests
 * [4]=primitive tests [5]=shadow rays ; [2..4] only count while detail counting is on */
int tinsel_hip_stats_detail(tinsel_hip* r, unsigned long long* out8)
{
    if (!r || !out8)
        return fail("stats_detail: bad arguments");
    return read_stats(r, out8);
}

int tinsel_hip_set_detail_counters(tinsel_hip* r, int enable)
{
    lookahead_cancel(r);
    if (!r)
        return fail("set_detail_counters: null");
    r->countDetail = enable != 0;
    return 0;
}

void tinsel_hip_reset_stats(tinsel_hip* r)
{
    lookahead_cancel(r);
    if (!r)
        return;
    (void)hipSetDevice(r->device);
    (void)hipDeviceSynchronize();
    (void)hipMemset(r->statsDev, 0, sizeof(unsigned long long)*kStatShards*kStatWords);
    (void)hipStreamSynchronize(nullptr);
    r->gpuSeconds = 0.0;
}

int tinsel_hip_enable_kernel_timing(tinsel_hip* r, int enable)
{
    lookahead_cancel(r);
    if (!r)
        return fail("enable_kernel_timing: null");
    r->timing = enable != 0;
    return 0;
}

int tinsel_hip_kernel_time_bytes(void) { return (int)sizeof(tinsel_kernel_time); }

int tinsel_hip_kernel_times(tinsel_hip* r, tinsel_kernel_time* out, int max_entries)
{
    lookahead_cancel(r);
    if (!r || !out)
        return fail("kernel_times: bad arguments");
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipDeviceSynchronize());
    float total[KN_COUNT] = { 0 };
    uint32_t launches[KN_COUNT] = { 0 };
    // busy time: the union of a kernel's launch intervals -- launches of one kernel on two streams overlap (render_impl's chunks), and
    // the sum of their durations counts the shared stretch twice
    std::vector<std::pair<float, float>> intervals[KN_COUNT];
    for (const TimedSpan& s : r->spans)
    {
        float ms = 0.0f, at = 0.0f;
        if (hipEventElapsedTime(&ms, s.start, s.stop) == hipSuccess)
        {
            total[s.kernel] += ms;
            launches[s.kernel]++;
            if (hipEventElapsedTime(&at, r->spans.front().start, s.start) == hipSuccess)
                intervals[s.kernel].push_back(std::make_pair(at, at + ms));
        }
    }
    float busy[KN_COUNT] = { 0 };
    for (int k = 0; k < KN_COUNT; ++k)
    {
        std::sort(intervals[k].begin(), intervals[k].end());
        float end = -1e30f;
        for (const auto& iv : intervals[k])
        {
            if (iv.second > end)
                busy[k] += iv.second - std::max(iv.first, end);
            end = std::max(end, iv.second);
        }
        if (intervals[k].size() != launches[k])
            busy[k] = total[k];
    }
    int n = 0;
    double sum = 0.0;
    for (int k = 0; k < KN_COUNT && n < max_entries; ++k)
    {
        if (!launches[k])
            continue;
        memset(&out[n], 0, sizeof(out[n]));
        strncpy(out[n].name, kKernelNames[k], sizeof(out[n].name) - 1);
        out[n].launches = launches[k];
        out[n].total_ms = total[k];
        out[n].busy_ms = busy[k];
        sum += total[k];
        ++n;
    }
    r->gpuSeconds += sum*1e-3;
    return n;
}

int tinsel_hip_reserve(tinsel_hip* r, int passes, int max_depth)
{
    lookahead_cancel(r);
    if (!r || !r->accum || passes < 1 || max_depth < 1)
        return fail("reserve: bad arguments (Init first)");
    HIP_TRY(hipSetDevice(r->device));
    const size_t perPass = slots_per_pass(r, r->width, r->height);
    int perBatch = (int)std::max<size_t>(1, batch_slots(r)/perPass);
    if (perBatch > passes)
        perBatch = passes;
    return ensure_batch(r, perPass*(size_t)perBatch, max_depth);
}

int tinsel_hip_set_batch_paths(tinsel_hip* r, unsigned long long max_paths)
{
    lookahead_cancel(r);
    if (!r || max_paths < 1024)
        return fail("set_batch_paths: bad arguments");
    r->maxBatchSlots = (size_t)max_paths;
    r->batchSlotsExplicit = true;
    return 0;
}

long long tinsel_hip_read_batch_radiance(tinsel_hip* r, float* out_rgbx, unsigned long long max_paths)
{
    lookahead_cancel(r);
    if (!r || !out_rgbx)
        return fail("read_batch_radiance: bad arguments");
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipDeviceSynchronize());
    const size_t n = std::min<size_t>((size_t)max_paths, r->lastBatchSlots);
    if (n)
        HIP_TRY(hipMemcpy(out_rgbx, r->ps.rad, sizeof(float4)*n, hipMemcpyDeviceToHost));
    return (long long)n;
}

int tinsel_hip_leaf(tinsel_hip* r, int op, int index, int n, const float* in, int in_stride, const uint32_t* seeds,
                    float* out, int out_stride, const tinsel_camera* camera, int width, int height)
{
    lookahead_cancel(r);
    if (!r || n <= 0 || !out || out_stride <= 0 || op < 0 || op > kLeafDisplay)
        return fail("leaf: bad arguments");
    if ((op == kLeafBsdfEval || op == kLeafBsdfSample || op == kLeafPrimIntersect || op == kLeafPrimSample) &&
        (index < 0 || index >= r->scene.numPrims))
        return fail("leaf: primitive index out of range");
    HIP_TRY(hipSetDevice(r->device));
    float* dIn = nullptr;
    uint32_t* dSeeds = nullptr;
    float* dOut = nullptr;
    int rc = 0;
    CameraParams cam;
    memset(&cam, 0, sizeof(cam));
    if (camera && width > 0 && height > 0)
        make_camera(*camera, width, height, cam);
    do {
        if (in && in_stride > 0)
        {
            if (hipMalloc((void**)&dIn, sizeof(float)*(size_t)n*in_stride) != hipSuccess ||
                hipMemcpy(dIn, in, sizeof(float)*(size_t)n*in_stride, hipMemcpyHostToDevice) != hipSuccess) { rc = fail("leaf: input upload failed"); break; }
        }
        if (seeds)
        {
            if (hipMalloc((void**)&dSeeds, sizeof(uint32_t)*(size_t)n) != hipSuccess ||
                hipMemcpy(dSeeds, seeds, sizeof(uint32_t)*(size_t)n, hipMemcpyHostToDevice) != hipSuccess) { rc = fail("leaf: seed upload failed"); break; }
        }
        if (hipMalloc((void**)&dOut, sizeof(float)*(size_t)n*out_stride) != hipSuccess) { rc = fail("leaf: output allocation failed"); break; }
        hipLaunchKernelGGL(k_leaf, dim3((n + kBlock - 1)/kBlock), dim3(kBlock), stack_bytes(r), nullptr, r->scene, op, index, n, dIn, in_stride,
                           dSeeds, dOut, out_stride, cam, r->stackNeed);
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { rc = fail("leaf: kernel failed"); break; }
        if (hipMemcpy(out, dOut, sizeof(float)*(size_t)n*out_stride, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail("leaf: download failed"); break; }
    } while (0);
    if (dIn) (void)hipFree(dIn);
    if (dSeeds) (void)hipFree(dSeeds);
    if (dOut) (void)hipFree(dOut);
    return rc;
}

int tinsel_hip_stack_entries(tinsel_hip* r) { return r ? r->stackNeed : 0; }
int tinsel_hip_walked_prims(tinsel_hip* r) { return (r && r->walkEnabled) ? r->walkPrims.count : 0; }
int tinsel_hip_nee_per_path(tinsel_hip* r) { return r ? r->neePerPath : 0; }

int tinsel_hip_queue_counts(tinsel_hip* r, uint32_t* out, int max_bounces)
{
    if (!r || !out || max_bounces < 1)
        return fail("queue_counts: bad arguments");
    if (r->batchPipeline < 0 || r->batchDepth < 1)
        return fail("queue_counts: nothing rendered yet");
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipDeviceSynchronize());
    const int n = std::min(max_bounces, std::min(r->batchDepth, r->lastFp.maxDepth));
    if (r->lastPipeline == TINSEL_PIPELINE_MEGAKERNEL || r->batchPipeline != r->lastPipeline)
        return fail("queue_counts: the last batch did not run a wavefront pipeline");
    // the counts are kept per region
    const bool split = r->lastPipeline == TINSEL_PIPELINE_WAVEFRONT_SPLIT && r->neePerPath > 0;
    const size_t W = r->lastRegions;
    std::vector<uint32_t> seg(W*(size_t)n*4, 0u);
    uint32_t* const src[4] = { r->ss.segFront, r->ss.segBack, r->ss.neeFront, r->ss.neeBack };
    for (int a = 0; a < (split ? 4 : 2); ++a)
        HIP_TRY(hipMemcpy(seg.data() + (size_t)a*W*n, src[a], W*(size_t)n*sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int b = 0; b < n; ++b)
    {
        unsigned long long live = 0, nee = 0;
        for (size_t g = 0; g < W; ++g)
        {
            live += seg[(size_t)b*W + g] + seg[W*n + (size_t)b*W + g];
            nee += seg[2*W*n + (size_t)b*W + g] + seg[3*W*n + (size_t)b*W + g];
        }
        // the fused kernel generates bounce 0's paths itself
        out[b] = (b == 0 && r->lastPipeline == TINSEL_PIPELINE_WAVEFRONT) ? r->lastFp.genCount : (uint32_t)live;
        out[max_bounces + b] = (uint32_t)nee;
    }
    return n;
}

// ---------------------------------------------------------------------------
// scene packs

// Yard-sticks on this GPU (tn_ubench.h): kind 0 = float4 stream copy of `bytes` bytes (units = bytes read + written),
// kinds 1..3 = dependent 64-B record chases through a table of `bytes` bytes rounded down to a power of two, `steps` visits
// per lane (units = records visited); the kind only names the kernel for the profiler (1 beyond the Infinity Cache, 2 the size
// of a walked tree, 3 inside one L2).  One warm-up launch, then one timed with HIP events.
// How a batch of `slots` path slots would be cut into regions on a device of `num_cus` CUs (streaming_grid + cut_regions): pure host
// arithmetic, no device needed -- tests/test_abi.py checks its invariants over the whole range of batch sizes on the CPU box
int tinsel_hip_plan_regions(unsigned long long slots, int num_cus, int nee_per_path, int fused, unsigned int* out)
{
    if (!out || slots == 0 || slots >= 0xffffffffull || num_cus < 1 || num_cus > 4096)
        return fail("plan_regions: bad arguments");
    tinsel_hip* r = new tinsel_hip();
    r->numCUs = num_cus;
    r->neePerPath = nee_per_path;
    // (alloc_dense's capacities)
    const size_t maxRegions = (size_t)num_cus*(size_t)grid_mult()*(kBlock/kWave)*3/2;
    r->splitMaxRegions = (uint32_t)maxRegions;
    r->splitCap = (size_t)slots + maxRegions*kWave;
    LaunchArgs a = {};
    int grid = streaming_grid(r, (size_t)slots, fused ? TINSEL_PIPELINE_WAVEFRONT : TINSEL_PIPELINE_WAVEFRONT_SPLIT);
    const int rc = cut_regions(r, a, (size_t)slots, &grid, fused ? (size_t)r->splitMaxRegions : (size_t)0);
    out[0] = a.ss.numRegions; out[1] = a.ss.regionLen; out[2] = a.ss.bigRegions; out[3] = a.ss.shortLen;
    out[4] = (unsigned int)grid; out[5] = r->splitMaxRegions;
    delete r;
    return rc;
}

int tinsel_hip_selftest_arith(int device_index, int op, int variant, unsigned long long* out_counts, unsigned int* out_first_bad)
{
    if (!out_counts || !out_first_bad || op < 0 || op > 2)
        return fail("selftest_arith: bad arguments");
    if (variant < 0)
        variant = op == 0 ? TN_RCP_VARIANT : op == 1 ? TN_SQRT_VARIANT : TN_RSQRT_VARIANT;      // what this library is built with
    HIP_TRY(hipSetDevice(device_index));
    unsigned long long* counts = nullptr;
    uint32_t* first = nullptr;
    HIP_TRY(hipMalloc((void**)&counts, 260*sizeof(unsigned long long)));
    if (hipMalloc((void**)&first, sizeof(uint32_t)) != hipSuccess)
    {
        (void)hipFree(counts);
        return fail("selftest_arith: allocation failed");
    }
    (void)hipMemset(counts, 0, 260*sizeof(unsigned long long));
    (void)hipMemset(first, 0xff, sizeof(uint32_t));
    bool known = true;
    switch (op*100 + variant)
    {
    case 0: launch_selftest_arith<0, 0>(counts, first); break;
    case 1: launch_selftest_arith<0, 1>(counts, first); break;
    case 11: launch_selftest_arith<0, 11>(counts, first); break;
    case 100: launch_selftest_arith<1, 0>(counts, first); break;
    case 101: launch_selftest_arith<1, 1>(counts, first); break;
    case 111: launch_selftest_arith<1, 11>(counts, first); break;
    case 121: launch_selftest_arith<1, 21>(counts, first); break;
    case 200: launch_selftest_arith<2, 0>(counts, first); break;
    case 201: launch_selftest_arith<2, 1>(counts, first); break;
    case 202: launch_selftest_arith<2, 2>(counts, first); break;
    case 203: launch_selftest_arith<2, 3>(counts, first); break;
    default: known = false; break;
    }
    int rc = 0;
    if (!known)
        rc = fail("selftest_arith: unknown variant");
    else if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess ||
             hipMemcpy(out_counts, counts, 260*sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess ||
             hipMemcpy(out_first_bad, first, sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail("selftest_arith: kernel failed");
    (void)hipFree(counts);
    (void)hipFree(first);
    return rc;
}

int tinsel_hip_ubench(int device_index, int kind, unsigned long long bytes, int steps, double* out_ms, double* out_units)
{
    if (kind < 0 || kind > 3 || bytes < 4096 || !out_ms || !out_units)
        return fail("ubench: bad arguments");
    HIP_TRY(hipSetDevice(device_index));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_index));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    float ms = 0.0f;
    int rc = 0;
    if (kind == 0)
    {
        const size_t n = (size_t)bytes/sizeof(float4);
        float4 *in = nullptr, *out = nullptr;
        if (hipMalloc((void**)&in, n*sizeof(float4)) != hipSuccess || hipMalloc((void**)&out, n*sizeof(float4)) != hipSuccess ||
            hipMemset(in, 0x3c, n*sizeof(float4)) != hipSuccess)
            rc = fail("ubench: allocation failed");
        else
        {
            // the best of a few shapes (workgroups per CU x interleaved / workgroup-contiguous x plain / non-temporal): what this chip sustains, not what one shape gets
            float best = 0.0f;
            for (int shape = 0; shape < 12 && !rc; ++shape)
            {
                const unsigned grid = (unsigned)prop.multiProcessorCount*(shape % 3 == 0 ? 8u : shape % 3 == 1 ? 16u : 32u);
                const bool contig = (shape/3) % 2 == 1 && n % ((size_t)grid*256*8) == 0;
                auto launch = [&] {
                    if (shape < 6)
                    {
                        if (contig) hipLaunchKernelGGL((k_ub_copy<false, true>), dim3(grid), dim3(256), 0, nullptr, (const float4*)in, out, n);
                        else hipLaunchKernelGGL((k_ub_copy<false, false>), dim3(grid), dim3(256), 0, nullptr, (const float4*)in, out, n);
                    }
                    else
                    {
                        if (contig) hipLaunchKernelGGL((k_ub_copy<true, true>), dim3(grid), dim3(256), 0, nullptr, (const float4*)in, out, n);
                        else hipLaunchKernelGGL((k_ub_copy<true, false>), dim3(grid), dim3(256), 0, nullptr, (const float4*)in, out, n);
                    }
                };
                launch();
                (void)hipEventRecord(e0, nullptr);
                launch();
                (void)hipEventRecord(e1, nullptr);
                if (hipEventSynchronize(e1) != hipSuccess || hipGetLastError() != hipSuccess)
                    rc = fail("ubench: copy kernel failed");
                float t = 0.0f;
                (void)hipEventElapsedTime(&t, e0, e1);
                if (best == 0.0f || t < best)
                    best = t;
            }
            ms = best;
            *out_units = 2.0*(double)(n*sizeof(float4));
        }
        if (in) (void)hipFree(in);
        if (out) (void)hipFree(out);
    }
    else
    {
        uint32_t nrec = 1;
        while ((unsigned long long)nrec*2ull*64ull <= bytes && nrec < (1u << 30))
            nrec *= 2u;
        if (steps < 1)
            steps = 64;
        const unsigned grid = (unsigned)prop.multiProcessorCount*16u;       // 4 workgroups x 4 waves per SIMD-quad: 16 waves per CU
        float4* recs = nullptr;
        float* out = nullptr;
        if (hipMalloc((void**)&recs, (size_t)nrec*64) != hipSuccess || hipMalloc((void**)&out, (size_t)grid*256*sizeof(float)) != hipSuccess)
            rc = fail("ubench: allocation failed");
        else
        {
            hipLaunchKernelGGL(k_ub_fill, dim3((unsigned)prop.multiProcessorCount*8u), dim3(256), 0, nullptr, recs, nrec);
            auto launch = [&] {
                if (kind == 1) hipLaunchKernelGGL((k_ub_gather<0>), dim3(grid), dim3(256), 0, nullptr, (const float4*)recs, nrec, steps, out);
                else if (kind == 2) hipLaunchKernelGGL((k_ub_gather<1>), dim3(grid), dim3(256), 0, nullptr, (const float4*)recs, nrec, steps, out);
                else hipLaunchKernelGGL((k_ub_gather<2>), dim3(grid), dim3(256), 0, nullptr, (const float4*)recs, nrec, steps, out);
            };
            launch();
            (void)hipEventRecord(e0, nullptr);
            launch();
            (void)hipEventRecord(e1, nullptr);
            if (hipEventSynchronize(e1) != hipSuccess || hipGetLastError() != hipSuccess)
                rc = fail("ubench: gather kernel failed");
            (void)hipEventElapsedTime(&ms, e0, e1);
            *out_units = (double)grid*256.0*(double)steps;
        }
        if (recs) (void)hipFree(recs);
        if (out) (void)hipFree(out);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *out_ms = (double)ms;
    return rc;
}

int tinsel_pack_open(void* blob, size_t size, tinsel_scene_desc* out_scene, tinsel_camera* out_camera, tinsel_options* out_options)
{
    if (!blob || size < sizeof(tinsel_pack_header) || !out_scene)
        return fail("pack_open: bad arguments");
    unsigned char* base = (unsigned char*)blob;
    tinsel_pack_header hdr;
    memcpy(&hdr, base, sizeof(hdr));
    if (memcmp(hdr.magic, TINSEL_PACK_MAGIC, 8) != 0 || hdr.version != 1)
        return fail("pack_open: not a TINPACK1 blob");
    if (hdr.total_bytes > size)
        return fail("pack_open: truncated blob");
    if (hdr.probe_width < 0 || hdr.probe_height < 0)
        return fail("pack_open: negative probe size");

    // written so that nothing can wrap: bytes <= total first, then off <= total - bytes
    auto in_range = [&](uint64_t off, uint64_t bytes) { return off >= sizeof(hdr) && bytes <= hdr.total_bytes && off <= hdr.total_bytes - bytes; };

    if (!in_range(hdr.off_primitives, (uint64_t)hdr.num_primitives*sizeof(tinsel_primitive)) ||
        !in_range(hdr.off_bvh_nodes, (uint64_t)hdr.num_bvh_nodes*sizeof(tinsel_bvh_node)))
        return fail("pack_open: section out of range");

    tinsel_primitive* prims = (tinsel_primitive*)(base + hdr.off_primitives);
    for (uint32_t i = 0; i < hdr.num_primitives; ++i)
    {
        tinsel_primitive& p = prims[i];
        if (p.type != TINSEL_GEOM_MESH)
            continue;
        tinsel_mesh_geometry& g = p.geo.mesh;
        if (g.num_vertices < 0 || g.num_indices < 0 || g.num_nodes < 0)
            return fail("pack_open: negative mesh counts");
        // offsets -> pointers, exactly once (a resolved pointer is far above total_bytes)
        const uint64_t offs[5] = { (uint64_t)(uintptr_t)g.positions, (uint64_t)(uintptr_t)g.normals, (uint64_t)(uintptr_t)g.indices,
                                   (uint64_t)(uintptr_t)g.nodes, (uint64_t)(uintptr_t)g.cdf };
        const uint64_t sizes[5] = { (uint64_t)g.num_vertices*12, (uint64_t)g.num_vertices*12, (uint64_t)g.num_indices*4,
                                    (uint64_t)g.num_nodes*32, (uint64_t)(g.num_indices/3)*4 };
        for (int k = 0; k < 5; ++k)
            if (!in_range(offs[k], sizes[k]))
                return fail("pack_open: mesh section out of range (or pack already opened)");
        g.positions = (const tinsel_vec3*)(base + offs[0]);
        g.normals = (const tinsel_vec3*)(base + offs[1]);
        g.indices = (const int32_t*)(base + offs[2]);
        g.nodes = (const tinsel_bvh_node*)(base + offs[3]);
        g.cdf = (const float*)(base + offs[4]);
    }

    memset(out_scene, 0, sizeof(*out_scene));
    out_scene->primitives = prims;
    out_scene->num_primitives = (int32_t)hdr.num_primitives;
    out_scene->bvh_nodes = (const tinsel_bvh_node*)(base + hdr.off_bvh_nodes);
    out_scene->num_bvh_nodes = (int32_t)hdr.num_bvh_nodes;
    out_scene->sky_horizon = hdr.sky_horizon;
    out_scene->sky_zenith = hdr.sky_zenith;
    if (hdr.off_probe_data)
    {
        const uint64_t n = (uint64_t)hdr.probe_width*hdr.probe_height;
        if (!in_range(hdr.off_probe_data, n*16) || !in_range(hdr.off_probe_pdf_x, n*4) || !in_range(hdr.off_probe_cdf_x, n*4) ||
            !in_range(hdr.off_probe_pdf_y, (uint64_t)hdr.probe_height*4) || !in_range(hdr.off_probe_cdf_y, (uint64_t)hdr.probe_height*4))
            return fail("pack_open: probe section out of range");
        out_scene->probe_valid = 1;
        out_scene->probe_width = hdr.probe_width;
        out_scene->probe_height = hdr.probe_height;
        out_scene->probe_data = (const tinsel_vec4*)(base + hdr.off_probe_data);
        out_scene->probe_pdf_x = (const float*)(base + hdr.off_probe_pdf_x);
        out_scene->probe_cdf_x = (const float*)(base + hdr.off_probe_cdf_x);
        out_scene->probe_pdf_y = (const float*)(base + hdr.off_probe_pdf_y);
        out_scene->probe_cdf_y = (const float*)(base + hdr.off_probe_cdf_y);
    }
    if (out_camera)
        *out_camera = hdr.camera;
    if (out_options)
        *out_options = hdr.options;
    return 0;
}

} // extern "C"

// ===========================================================================
// tinsel_hip_group: N devices of one node behind one Renderer (include/tinsel_hip.h).
//
// One worker thread per member drives that member's device (launches are asynchronous, but N x (3 maxDepth + 2)
// launches per batch from ONE thread would serialise the devices' queues at 1-pass-per-call rates); the caller's thread
// only posts a job and waits.  The members' accumulators hold each member's own partial sums since Init and are never
// written by the reduce: the sum goes to `total` on member 0, so calling Render twice cannot count a sample twice.

namespace {

struct RcclApi
{
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;

    bool load()
    {
        if (lib)
            return true;
        const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        for (const char* n : names)
            if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) != nullptr)
                break;
        if (!lib)
        {
            error = std::string("RCCL not found (librccl.so.1): ") + (dlerror() ? dlerror() : "?");
            return false;
        }
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        Reduce = (decltype(Reduce))dlsym(lib, "ncclReduce");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !Reduce || !GetErrorString)
        {
            error = "RCCL library lacks ncclCommInitAll / ncclReduce";
            lib = nullptr;
            return false;
        }
        return true;
    }
};

RcclApi g_rccl;

} // namespace

namespace tn {
// validation arm of the reduce (members sharing one device): total = sum over members in rank order
struct SumSources { const float4* src[16]; int n; };
__global__ void k_sum_accums(SumSources s, float4* __restrict__ total, size_t count)
{
    const size_t i = (size_t)blockIdx.x*blockDim.x + threadIdx.x;
    if (i >= count)
        return;
    float4 a = s.src[0][i];
    for (int k = 1; k < s.n; ++k)
    {
        const float4 b = s.src[k][i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    total[i] = a;
}
} // namespace tn

namespace {

enum { GJ_NONE = 0, GJ_INIT, GJ_RENDER, GJ_REDUCE, GJ_AHEAD, GJ_QUIT };

struct GroupMember
{
    tinsel_hip* r = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;
    std::thread thread;
    int rc = 0;
    std::string error;
};

} // namespace

struct tinsel_hip_group
{
    std::vector<GroupMember> members;
    bool oneDevice = false;         // validation: all members on device 0, device-local sum instead of RCCL
    bool solo = true;               // one member used directly: no threads, no reduce, total aliases its accumulator
    int width = 0, height = 0;
    float4* total = nullptr;        // on member 0's device; == member 0's accumulator when there is one member

    // Look-ahead for the reference's call pattern at N members (tinsel_hip_group_set_lookahead): after a read-back every
    // member keeps a queue of speculated calls (lookahead_extend: one batch of `depth` calls of ITS shard, one snapshot per
    // call) and the NEXT call's snapshots are reduced into `totalNext` while this call's `total` crosses PCIe.  A matching
    // call then only waits for that job, swaps the buffers and copies.
    int lookahead = TINSEL_LOOKAHEAD_OFF;
    float4* totalNext = nullptr;
    bool aheadInFlight = false;     // a GJ_AHEAD job has been posted and not yet waited for
    bool aheadValid = false;        // every member holds a snapshot of the call described below (and totalNext its reduced sum)
    tinsel_camera aheadCamera;
    tinsel_options aheadOptions;
    int aheadPasses = 0;
    hipStream_t copyStream = nullptr;   // on member 0's device
    void* pinnedPtr = nullptr;
    size_t pinnedBytes = 0;

    // job hand-off: the caller posts (job, epoch), every worker runs it for its member and reports
    std::mutex mu;
    std::condition_variable cvWork, cvDone;
    unsigned long long epoch = 0;
    int pending = 0;
    int job = GJ_NONE;
    tinsel_camera camera;
    tinsel_options options;
    int passes = 0;
};

namespace {

void group_worker(tinsel_hip_group* g, int rank)
{
    GroupMember& m = g->members[(size_t)rank];
    unsigned long long seen = 0;
    for (;;)
    {
        int job;
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cvWork.wait(lk, [&] { return g->epoch != seen; });
            seen = g->epoch;
            job = g->job;
        }
        int rc = 0;
        g_error.clear();
        if (job == GJ_INIT)
        {
            rc = tinsel_hip_init(m.r, g->width, g->height);
        }
        else if (job == GJ_RENDER)
        {
            rc = render_impl(m.r, &g->camera, &g->options, g->passes, m.stream);
            if (!rc && hipStreamSynchronize(m.stream) != hipSuccess)
                rc = fail("group: a member's render failed on the device");
        }
        else if (job == GJ_REDUCE)
        {
            // every member enters the collective from its own thread and stream: the ring runs over xGMI
            (void)hipSetDevice(m.device);
            const ncclResult_t e = g_rccl.Reduce(m.r->accum, g->total, (size_t)g->width*g->height*4, ncclFloat, ncclSum, 0, m.comm, m.stream);
            if (e != ncclSuccess)
                rc = fail(std::string("group: ncclReduce: ") + g_rccl.GetErrorString(e));
            else if (hipStreamSynchronize(m.stream) != hipSuccess)
                rc = fail("group: the reduce failed on the device");
        }
        else if (job == GJ_AHEAD)
        {
            // the NEXT call, speculated: keep this member's queue of traced calls deep enough, then reduce the snapshot
            // the next call will swap in (the members' queues advance in lockstep: the same calls, the same depth rule)
            tinsel_hip* r = m.r;
            rc = lookahead_streams(r);
            const int depth = rc ? 0 : lookahead_depth(r, g->aheadPasses);
            if (!rc && depth <= 0)
                rc = fail("group: one call does not fit a batch");
            if (!rc && (int)r->specQueue.size() <= depth)
            {
                if (r->specQueue.empty())
                    r->specNextPass = r->passIndex;
                r->specCamera = g->aheadCamera;
                r->specOptions = g->aheadOptions;
                r->specPasses = g->aheadPasses;
                rc = lookahead_extend(r, &g->aheadCamera, &g->aheadOptions, g->aheadPasses, depth);
                if (rc)
                    lookahead_cancel(r);        // kernels of the failed speculation may be in flight on the work stream: wait, drop the shots
            }
            if (!rc && !g->oneDevice)
            {
                const tinsel_hip::SpecShot& shot = r->specQueue.front();
                if (hipStreamWaitEvent(m.stream, shot.ready, 0) != hipSuccess)
                    rc = fail("group: look-ahead wait failed");
                else
                {
                    const ncclResult_t e = g_rccl.Reduce(shot.buf, g->totalNext, (size_t)g->width*g->height*4, ncclFloat, ncclSum, 0, m.comm, m.stream);
                    if (e != ncclSuccess)
                        rc = fail(std::string("group: ncclReduce: ") + g_rccl.GetErrorString(e));
                    else if (hipStreamSynchronize(m.stream) != hipSuccess)
                        rc = fail("group: the look-ahead reduce failed on the device");
                }
            }
        }
        m.rc = rc;
        m.error = rc ? g_error : std::string();
        {
            std::lock_guard<std::mutex> lk(g->mu);
            --g->pending;
        }
        g->cvDone.notify_all();
        if (job == GJ_QUIT)
            return;
    }
}

// posts `job` to every member's thread
void group_post(tinsel_hip_group* g, int job)
{
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->job = job;
        g->pending = (int)g->members.size();
        ++g->epoch;
    }
    g->cvWork.notify_all();
}

// waits for the posted job; 0 when every member succeeded
int group_wait(tinsel_hip_group* g)
{
    {
        std::unique_lock<std::mutex> lk(g->mu);
        g->cvDone.wait(lk, [&] { return g->pending == 0; });
    }
    for (size_t k = 0; k < g->members.size(); ++k)
        if (g->members[k].rc)
            return fail("member " + std::to_string(k) + ": " + g->members[k].error);
    return 0;
}

int group_run(tinsel_hip_group* g, int job)
{
    group_post(g, job);
    return group_wait(g);
}

// Look-ahead bookkeeping.  group_ahead_join: the job in flight (if any) has ended; the workers are idle afterwards and the
// caller's thread may touch the members.  group_ahead_drop: ... and nothing speculated survives (Init, another camera,
// look-ahead switched off).
void group_ahead_join(tinsel_hip_group* g)
{
    if (!g->aheadInFlight)
        return;
    g->aheadInFlight = false;
    if (group_wait(g))
        g->aheadValid = false;      // a member could not speculate: the plain path still works
}

void group_ahead_drop(tinsel_hip_group* g)
{
    group_ahead_join(g);
    g->aheadValid = false;
    if (!g->solo)
        for (GroupMember& m : g->members)
            lookahead_cancel(m.r);
}

void group_unpin(tinsel_hip_group* g)
{
    if (!g->pinnedPtr)
        return;
    (void)hipSetDevice(g->members[0].device);
    if (g->copyStream)
        (void)hipStreamSynchronize(g->copyStream);
    (void)hipHostUnregister(g->pinnedPtr);
    g->pinnedPtr = nullptr;
    g->pinnedBytes = 0;
}

// total = sum of the members' accumulators, on member 0's device (the workers must be idle: group_ahead_join)
int group_reduce(tinsel_hip_group* g)
{
    const size_t n = g->members.size();
    if (g->solo)
        return 0;                       // total IS member 0's accumulator
    if (!g->oneDevice)
        return group_run(g, GJ_REDUCE);
    SumSources src;
    src.n = (int)n;
    for (size_t k = 0; k < n; ++k)
        src.src[k] = g->members[k].r->accum;
    const size_t count = (size_t)g->width*g->height;
    HIP_TRY(hipSetDevice(g->members[0].device));
    hipLaunchKernelGGL(k_sum_accums, dim3((unsigned)((count + 255)/256)), dim3(256), 0, g->members[0].stream, src, g->total, count);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(g->members[0].stream));
    return 0;
}

} // namespace

extern "C" {

void tinsel_hip_group_destroy(tinsel_hip_group* g)
{
    if (!g)
        return;
    group_ahead_join(g);
    if (!g->members.empty() && g->members[0].r)
        group_unpin(g);
    bool threads = false;
    for (GroupMember& m : g->members)
        threads = threads || m.thread.joinable();
    if (threads)
    {
        (void)group_run(g, GJ_QUIT);
        for (GroupMember& m : g->members)
            if (m.thread.joinable())
                m.thread.join();
    }
    for (GroupMember& m : g->members)
    {
        (void)hipSetDevice(m.device);
        if (m.comm && g_rccl.CommDestroy)
            (void)g_rccl.CommDestroy(m.comm);
        if (m.stream)
            (void)hipStreamDestroy(m.stream);
    }
    if (!g->solo && !g->members.empty())
    {
        (void)hipSetDevice(g->members[0].device);
        if (g->total) (void)hipFree(g->total);
        if (g->totalNext) (void)hipFree(g->totalNext);
        if (g->copyStream) (void)hipStreamDestroy(g->copyStream);
    }
    for (GroupMember& m : g->members)
        tinsel_hip_destroy(m.r);
    delete g;
}

tinsel_hip_group* tinsel_hip_group_create(const tinsel_scene_desc* scene, int num_gpus, int tile)
{
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0)
    {
        fail("group_create: no HIP device visible -- this library has no CPU fallback");
        return nullptr;
    }
    const char* one = getenv("TINSEL_HIP_GROUP_ONE_DEVICE");
    const bool oneDevice = one && atoi(one) != 0;
    int n = num_gpus > 0 ? num_gpus : visible;
    if (n > 16)
    {
        fail("group_create: at most 16 members");
        return nullptr;
    }
    if (n > visible && !oneDevice)
    {
        fail("group_create: " + std::to_string(n) + " GPUs requested, " + std::to_string(visible) + " visible");
        return nullptr;
    }
    if (tile <= 0)
        tile = 64;

    // TINSEL_HIP_GROUP_FORCE_RCCL=1: a ONE-member group also takes the threaded path and a 1-rank ncclReduce -- the only
    // way to execute the RCCL calls (dlopen, communicator, reduce into `total` on the member's stream) on a single-GPU box
    const bool forceRccl = getenv("TINSEL_HIP_GROUP_FORCE_RCCL") && atoi(getenv("TINSEL_HIP_GROUP_FORCE_RCCL")) != 0 && !oneDevice;
    tinsel_hip_group* g = new tinsel_hip_group();
    g->oneDevice = oneDevice && n > 1;
    g->solo = n == 1 && !forceRccl;
    g->members.resize((size_t)n);
    for (int k = 0; k < n; ++k)
    {
        GroupMember& m = g->members[(size_t)k];
        m.device = g->oneDevice ? 0 : k;
        m.r = tinsel_hip_create(scene, m.device);
        if (!m.r || tinsel_hip_set_shard(m.r, k, n, tile) || hipSetDevice(m.device) != hipSuccess ||
            hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking) != hipSuccess)
        {
            if (m.r)
                fail("group_create: member " + std::to_string(k) + " could not be set up");
            tinsel_hip_group_destroy(g);
            return nullptr;
        }
    }
    if (!g->solo && !g->oneDevice)
    {
        std::vector<int> devs((size_t)n);
        std::vector<ncclComm_t> comms((size_t)n);
        for (int k = 0; k < n; ++k)
            devs[(size_t)k] = g->members[(size_t)k].device;
        if (!g_rccl.load())
        {
            fail("group_create: " + g_rccl.error);
            tinsel_hip_group_destroy(g);
            return nullptr;
        }
        const ncclResult_t e = g_rccl.CommInitAll(comms.data(), n, devs.data());
        if (e != ncclSuccess)
        {
            fail(std::string("group_create: ncclCommInitAll: ") + g_rccl.GetErrorString(e));
            tinsel_hip_group_destroy(g);
            return nullptr;
        }
        for (int k = 0; k < n; ++k)
            g->members[(size_t)k].comm = comms[(size_t)k];
    }
    if (!g->solo)
        for (int k = 0; k < n; ++k)
            g->members[(size_t)k].thread = std::thread(group_worker, g, k);
    return g;
}

int tinsel_hip_group_init(tinsel_hip_group* g, int width, int height)
{
    if (!g || width <= 0 || height <= 0)
        return fail("group_init: bad arguments");
    group_ahead_drop(g);
    group_unpin(g);                 // the reference's caller has freed its array by now (main.cpp:73-87)
    g->width = width;
    g->height = height;
    if (g->solo)
    {
        if (tinsel_hip_init(g->members[0].r, width, height))
            return -1;
        g->total = g->members[0].r->accum;
        return 0;
    }
    if (group_run(g, GJ_INIT))
        return -1;
    HIP_TRY(hipSetDevice(g->members[0].device));
    if (g->total)
        (void)hipFree(g->total);
    if (g->totalNext)
        (void)hipFree(g->totalNext);
    g->total = g->totalNext = nullptr;
    HIP_TRY(hipMalloc((void**)&g->total, sizeof(float4)*(size_t)width*height));
    HIP_TRY(hipMalloc((void**)&g->totalNext, sizeof(float4)*(size_t)width*height));
    HIP_TRY(hipMemset(g->total, 0, sizeof(float4)*(size_t)width*height));
    HIP_TRY(hipStreamSynchronize(nullptr));
    if (!g->copyStream)
        HIP_TRY(hipStreamCreateWithFlags(&g->copyStream, hipStreamNonBlocking));
    return 0;
}

int tinsel_hip_group_set_lookahead(tinsel_hip_group* g, int enable)
{
    if (!g)
        return fail("group_set_lookahead: null");
    if (g->solo)
        return tinsel_hip_set_lookahead(g->members[0].r, enable);
    if (!enable)
        group_ahead_drop(g);
    if (enable != TINSEL_LOOKAHEAD_PIN_OUTPUT)
        group_unpin(g);
    g->lookahead = enable == TINSEL_LOOKAHEAD_PIN_OUTPUT ? TINSEL_LOOKAHEAD_PIN_OUTPUT : (enable ? TINSEL_LOOKAHEAD_ON : TINSEL_LOOKAHEAD_OFF);
    return 0;
}

int tinsel_hip_group_render(tinsel_hip_group* g, const tinsel_camera* camera, const tinsel_options* options, float* out_rgba, int passes)
{
    if (!g || !camera || !options)
        return fail("group_render: null argument");
    if (!g->total)
        return fail("group_render: Init first");
    if (g->solo)
        return tinsel_hip_render(g->members[0].r, camera, options, out_rgba, passes);
    const size_t bytes = sizeof(float4)*(size_t)g->width*g->height;

    // 1. this call's passes: speculated by the previous call (every member holds their snapshot, totalNext their reduced sum)
    //    or traced and reduced now
    const bool wanted = g->lookahead && out_rgba && passes >= 1 && options->width == g->width && options->height == g->height &&
                        options->mode == TINSEL_MODE_PATHTRACE && options->max_depth >= 1;
    group_ahead_join(g);
    bool hit = wanted && g->aheadValid && passes == g->aheadPasses && memcmp(camera, &g->aheadCamera, sizeof(*camera)) == 0 &&
               memcmp(options, &g->aheadOptions, sizeof(*options)) == 0;
    for (const GroupMember& m : g->members)
        hit = hit && !m.r->specQueue.empty() && m.r->specPasses == passes;
    if (hit)
    {
        for (GroupMember& m : g->members)
        {
            HIP_TRY(hipSetDevice(m.device));
            if (lookahead_commit(m.r, passes))
                return -1;
        }
        if (g->oneDevice)
        {
            if (group_reduce(g))        // validation arm: the device-local sum of the snapshots just swapped in
                return -1;
        }
        else
            std::swap(g->total, g->totalNext);
    }
    else
    {
        group_ahead_drop(g);
        g->camera = *camera;
        g->options = *options;
        g->passes = passes;
        if (group_run(g, GJ_RENDER))
            return -1;
        if (!out_rgba)
            return 0;
        if (group_reduce(g))
            return -1;
    }
    HIP_TRY(hipSetDevice(g->members[0].device));
    if (!wanted)
    {
        HIP_TRY(hipMemcpy(out_rgba, g->total, bytes, hipMemcpyDeviceToHost));
        return 0;
    }

    // 2. the sum starts towards the host and the members go on with the next call meanwhile: its passes traced (a batch of
    //    `depth` calls at a time), its snapshots reduced into totalNext -- per call the caller waits for one reduce (already
    //    done, as a rule) and one copy.  Page-locking the caller's array is an explicit opt-in, as for one device.
    const bool pin = g->lookahead == TINSEL_LOOKAHEAD_PIN_OUTPUT;
    if (g->pinnedPtr && (!pin || g->pinnedPtr != (void*)out_rgba || g->pinnedBytes != bytes))
        group_unpin(g);
    if (pin && !g->pinnedPtr)
    {
        if (hipHostRegister(out_rgba, bytes, hipHostRegisterDefault) == hipSuccess)
        {
            g->pinnedPtr = out_rgba;
            g->pinnedBytes = bytes;
        }
        else
            (void)hipGetLastError();
    }
    auto post_ahead = [&] {
        g->aheadCamera = *camera;
        g->aheadOptions = *options;
        g->aheadPasses = passes;
        g->aheadValid = true;           // unless the job fails (group_ahead_join)
        g->aheadInFlight = true;
        group_post(g, GJ_AHEAD);
    };
    // the workers start first, then this thread copies (a blocking copy either way: the call cannot return before its image
    // is on the host; into a page-locked array it is one DMA, into a pageable one it is staged by the runtime)
    post_ahead();
    HIP_TRY(hipSetDevice(g->members[0].device));
    HIP_TRY(hipMemcpy(out_rgba, g->total, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int tinsel_hip_group_present(tinsel_hip_group* g, const tinsel_options* options, int nlm_width, float nlm_falloff, float* out_rgba)
{
    if (!g || !g->total || !options)
        return fail("group_present: bad arguments (Init and Render first)");
    tinsel_hip* r0 = g->members[0].r;
    if (g->solo)
        return tinsel_hip_present(r0, options, nlm_width, nlm_falloff, out_rgba);
    group_ahead_join(g);                // (what was speculated stays: the members' committed sums are not touched by it)
    if (group_reduce(g))
        return -1;
    // the display stage of member 0 on the reduced frame
    float4* own = r0->accum;
    r0->accum = g->total;
    const int rc = tinsel_hip_present(r0, options, nlm_width, nlm_falloff, out_rgba);
    r0->accum = own;
    if (r0->presented == g->total)
        r0->presented = nullptr;
    return rc;
}

int tinsel_hip_group_size(tinsel_hip_group* g) { return g ? (int)g->members.size() : 0; }

tinsel_hip* tinsel_hip_group_member(tinsel_hip_group* g, int rank)
{
    if (!g || rank < 0 || rank >= (int)g->members.size())
        return nullptr;
    group_ahead_drop(g);        // the caller may do anything to the member: nothing speculated may be in flight or survive
    return g->members[(size_t)rank].r;
}

} // extern "C"
