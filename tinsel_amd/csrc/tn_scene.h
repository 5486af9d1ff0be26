// tn_scene.h -- the scene as it lives in HBM (and, when it fits, in LDS).
//
// The reference keeps 32-B BVHNode records (bvh.h:9-20) and fetches THREE of them per
// internal visit (the node, then both children: intersection.h:766-776), 272-B Primitives
// (scene.h:138-159) and index+vertex gathers per triangle test (intersection.h:638-644).
// Here the same trees, boxes and visit order are re-laid for 64-B / 128-B aligned loads:
//
//   Node64   one record per INTERNAL node holding BOTH children's boxes and child refs:
//            one 64-B load (4 x dwordx4) per internal visit; node pairs are 128-B aligned.
//            A child ref is (leaf<<31 | index): leaves carry their item id in the ref,
//            so leaf nodes are never fetched at all.
//   Tri48    the three vertex positions pre-gathered per triangle (+ the three vertex
//            indices in the .w lanes for the normal fetch after the closest hit).
//   Prim64   what PrimitiveIntersect needs (pose, geometry, kind): 64 B.
//   Mat128   what shading needs, with every material-only sub-expression of
//            disney.h / scene.h pre-evaluated on the host in the reference's own
//            precision (see host_scene.cpp), 128 B.
#pragma once

#include "tn_math.h"

namespace tn {

constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kNoNode = 0xffffffffu;
struct alignas(64) Node64
{
    // children L and R of one internal node; boxes exactly as in the reference nodes
    float lminx, lminy, lminz, lmaxx;
    float lmaxy, lmaxz, rminx, rminy;
    float rminz, rmaxx, rmaxy, rmaxz;
    uint32_t left, right;       // child refs
    uint32_t pad0, pad1;
};
static_assert(sizeof(Node64) == 64, "Node64");

struct alignas(16) Tri48
{
    float ax, ay, az; int32_t i0;
    float bx, by, bz; int32_t i1;
    float cx, cy, cz; int32_t i2;
};
static_assert(sizeof(Tri48) == 48, "Tri48");

enum : uint32_t
{
    kPrimSphere = 0,
    kPrimPlane = 1,
    kPrimMesh = 2,
};

enum : uint32_t
{
    kPrimMoving = 1u,           // startTransform != endTransform: interpolate per ray
    kPrimWalked = 2u,           // mesh in HBM whose closest hit k_walk (tn_walk.h) computes ahead of the scan kernels
    kPrimQuadArena = 8u,        // a mesh that is ONE internal node over two triangles (a quad) and rides in the arena: g0 / g1 = where its node, triangles, normals and cdf are (quad_offsets)
    kPrimNoRot = 4u,            // static pose whose rotation is exactly the quaternion (+0, +0, +0, 1): Rotate() written down (tn_isect.h)
    kPrimWalkLaneShift = 8,     // bits 8..10: which of the (up to 7) walked primitives this is = its record lane
};

struct alignas(64) Prim64
{
    // pose at ray time for static primitives == InterpolateTransform(start, end, t) for any t
    float px, py, pz, s;
    float rx, ry, rz, rw;
    float g0, g1, g2, g3;       // sphere: radius,-,-,- ; plane: the four coefficients ; static mesh: -,-,-,1.0f/s (divided on the host); kPrimQuadArena: g0, g1 = QuadOffsets (bits)
    uint32_t type;
    uint32_t flags;
    uint32_t mesh;              // index into DevScene::meshes (kPrimMesh)
    uint32_t moving;            // index into DevScene::moving (kPrimMoving)
};
static_assert(sizeof(Prim64) == 64, "Prim64");

// kPrimQuadArena: the arena offsets of the quad's four arrays, in 128-B units (what ArenaBuilder aligns to), two per word
struct QuadOffsets { uint32_t nodes, tris, normals, cdf; };         // bytes
TN_HD QuadOffsets quad_offsets(uint32_t w0, uint32_t w1)
{
    QuadOffsets q = { (w0 & 0xffffu) << 7, (w0 >> 16) << 7, (w1 & 0xffffu) << 7, (w1 >> 16) << 7 };
    return q;
}

struct alignas(64) Moving64
{
    float spx, spy, spz, ss; float srx, sry, srz, srw;      // startTransform
    float epx, epy, epz, es; float erx, ery, erz, erw;      // endTransform
};
static_assert(sizeof(Moving64) == 64, "Moving64");

struct alignas(128) Mat128
{
    float emission[3]; float ior;           // Material::GetIndexOfRefraction() (scene.h:72-78)
    float color[3];    float metallic;
    float absorption[3]; float subsurface;
    float cspec0[3];   float roughness;     // Cspec0 of BSDFEval (disney.h:306-310)
    float sqrtColor[3]; float transmission; // sqrtf(color) of the sub-surface lobe (disney.h:352)
    float clearcoat;
    float clearcoatAlpha;                   // Lerp(.1,.001,clearcoatGloss) (disney.h:387)
    float area;                             // PrimitiveArea (intersection.h:833-853)
    int32_t lightSamples;
    float clearcoatA2;                      // a*a and logf(a*a) of GTR1 (disney.h:59-61), a = clearcoatAlpha
    float clearcoatLogA2;
    // per-light constants of the MIS weights, divided once on the host with the reference's own fp32 expressions instead of once per
    // light sample and bounce on the device (an IEEE division is a dozen instructions): 1.0f/PrimitiveArea (render.cpp:182, 292),
    // 1.0f/numSamples (:223), and with N = lightSamples + kBsdfSamples: kBsdfSamples/N, float(lightSamples)/N (:209-211, 296-298)
    float rcpArea, rcpLightSamples;
    float cbsdf, clight;
    float pad[2];
};
static_assert(sizeof(Mat128) == 128, "Mat128");

// World-space leaf box of one primitive, copied from the reference's scene BVH leaves
// (PrimitiveBounds, intersection.h:906-939).  Used by the flat scene-level scan (tn_isect.h).
struct alignas(32) PrimBox
{
    float minx, miny, minz, maxx, maxy, maxz;
    uint32_t alwaysHit;         // 1: an "infinite" box (planes: +-1e8) that every sane ray hits; 2: ... and the plane is in DevScene::planeEq (the flat scan tests those four at a time)
    uint32_t pad;
};
static_assert(sizeof(PrimBox) == 32, "PrimBox");

// One light of the scene as the light loops read it (LightCursor, nee_sum: tn_integrator.h): ONE 16-B record instead of the primitive index and,
// behind it, two words of that primitive's 128-B material record
struct alignas(16) LightRec
{
    int32_t prim;
    int32_t lightSamples;
    float rcpLightSamples;      // 1.0f/lightSamples, divided on the host (Mat128)
    int32_t pad;
};

struct DevMesh
{
    const Node64* nodes;
    const Tri48* tris;
    const float* normals;       // 3 floats per vertex
    const float* cdf;           // per-triangle area CDF (mesh.cpp:340-368)
    uint32_t root;              // child ref of the root (leaf ref when the mesh has one triangle)
    int32_t numTris;
    int32_t stackNeed;          // worst-case traversal stack entries for this tree
    int32_t inArena;            // 1: nodes/tris/normals/cdf live inside DevScene::arena (and follow it into LDS)
    uint32_t offNodes, offTris, offNormals, offCdf;     // byte offsets inside the arena (inArena only)
    int32_t topCount;           // nodes [0, topCount) are the top of the tree in breadth-first order (k_walk stages a prefix into LDS)
    int32_t twoLeaves;          // 1: the tree is one internal node over two one-triangle leaves (a quad): ray_mesh_two_leaves
    int32_t numInternal;        // Node64 records of this tree
    int32_t pad0;
};

struct DevProbe
{
    const float4* data;
    const float* pdfX;
    const float* cdfX;
    const float* pdfY;
    const float* cdfY;
    int32_t width, height;
    int32_t valid;
    int32_t pad;
    // opt-in (tinsel_hip_set_probe_sampling): alias table over the W*H texels, entry k = { probability of keeping k
    // (float bits), the texel to take otherwise }; null: the reference's two binary searches (probe.h:205-236)
    const uint2* alias;
};

struct DevScene
{
    const Node64* nodes;        // scene-level BVH
    const Prim64* prims;
    const Mat128* mats;         // one per primitive, same index
    const Moving64* moving;
    const DevMesh* meshes;
    const LightRec* lights;     // the primitives with lightSamples > 0, in primitive order
    uint32_t root;
    int32_t numPrims;
    int32_t numLights;
    int32_t totalLightSamples;  // sum of lightSamples over lights (+1 when a probe is valid = NEE rays per bounce)
    float horizon[3];
    float zenith[3];
    DevProbe probe;
    // The scene-level records (BVH, Prim64, Mat128, moving poses, light list, mesh table) and the arrays
    // of small meshes are ONE contiguous allocation, so a block can stage all of it into LDS with a
    // single cooperative copy when it is small enough (arenaLdsBytes != 0); see stage_scene_lds().
    const unsigned char* arena;
    uint32_t arenaBytes;
    uint32_t arenaLdsBytes;     // == arenaBytes when the kernels should stage it, else 0
    int32_t numMeshes;
    int32_t allInArena;         // every mesh rides in the arena: the LDS-only kernel variants may be used
    const PrimBox* primBoxes;   // [numPrims], in the arena
    int32_t flatScan;           // 1: few primitives -> scene level is a wave-uniform scan (trace_flat)
    int32_t hasMedia;           // 0: no material absorbs, rayAbsorption stays 0 -> its 16-B state record is skipped
    int32_t sortQueues;         // 1: the fused kernel sorts the next bounce's queue by ray_meets_bounded_prim (open scenes)
    int32_t deferMeshes;        // 1: trace_flat walks the meshes a ray enters after the scan, all lanes together (>= 2 mesh primitives)
    // the scene's always-hit planes once more, for the flat scan: equations four by four (padded with planes no ray meets) and
    // their primitive indices, in the arena (trace_flat, tn_isect.h)
    const float4* planeEq;
    const int32_t* planeIdx;
    int32_t numPlanes;              // (the table is padded to a multiple of four)
    // the primitives the flat scan's loop visits: bit i clear = primitive i is a plane of the table above, tested ahead of the loop (the loop
    // used to fetch such a primitive's leaf box only to read "pass by": five of cornell's eight scalar round trips per ray)
    unsigned long long scanMask;
};

// Compile-time view of where the scene lives.  SceneT<true>: the whole scene (arena incl. every mesh)
// has been staged into LDS and all accessors resolve to LDS addresses at compile time (ds_read);
// SceneT<false>: generic pointers (HBM, or an LDS copy reached through flat loads).
// WALKED_ONLY: every mesh primitive of the scene has its closest hits precomputed by k_walk (tn_walk.h), so the scan
// kernels are compiled without the inline mesh walk (no deep stack, half the registers, twice the waves).  2: or is a QUAD
// (DevMesh::twoLeaves: a lamp, a card) tested in the scan by ray_mesh_two_leaves, which needs neither stack nor loop.
// DEFER: trace_flat's deferred mesh walks (tn_isect.h) compiled out (0), in (1), or behind DevScene::deferMeshes (2).  The
// fused kernel is built both ways -- the second loop costs 2 % where there is nothing to defer (cornell).
typedef float ConstF4V __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) ConstF4V* ConstF4;

// MIXED (with LDS): the arena is staged whole and every scene record resolves to LDS at compile time, but some meshes live in
// HBM -- the split pipeline's scenes (glass, the 524k-triangle config).  Only the mesh accessors below then choose per mesh;
// without it those kernels reach everything through generic pointers (flat loads, which wait on both memory counters).
template <bool LDS, int WALKED_ONLY = 0, int DEFER = 2, bool MIXED = false>
struct SceneT : DevScene
{
    static constexpr bool kLds = LDS;
    static constexpr bool kMixed = MIXED;
    static constexpr bool kWalkedOnly = WALKED_ONLY != 0;
    static constexpr bool kQuadsInline = WALKED_ONLY == 2;
    static constexpr int kDefer = DEFER;
    const unsigned char* ldsBase;
    // The flat scan reads primitive records and leaf boxes at a wave-uniform index: through these pointers (the arena's copy
    // in HBM, constant address space) they are scalar loads into SGPRs instead of 64 lanes reading the same LDS words
    ConstF4 kPrims, kBoxes, kPlaneEq, kPlaneIdx;
    // closest-hit records of the walked primitives for the ray being traced (tn_walk.h): record lane kb of the ray
    // lives at walkRec[(walkItem + kb)*2 .. +1]; null = walk the mesh inline (ray_mesh)
    const float4* walkRec;
    uint32_t walkItem;
};

template <class SC> TN_D const Node64* mesh_nodes(const SC& sc, const DevMesh& m)
{
    if constexpr (SC::kLds && SC::kMixed) return m.inArena ? reinterpret_cast<const Node64*>(sc.ldsBase + m.offNodes) : m.nodes;
    else if constexpr (SC::kLds) return reinterpret_cast<const Node64*>(sc.ldsBase + m.offNodes); else return m.nodes;
}
template <class SC> TN_D const Tri48* mesh_tris(const SC& sc, const DevMesh& m)
{
    if constexpr (SC::kLds && SC::kMixed) return m.inArena ? reinterpret_cast<const Tri48*>(sc.ldsBase + m.offTris) : m.tris;
    else if constexpr (SC::kLds) return reinterpret_cast<const Tri48*>(sc.ldsBase + m.offTris); else return m.tris;
}
template <class SC> TN_D const float* mesh_normals(const SC& sc, const DevMesh& m)
{
    if constexpr (SC::kLds && SC::kMixed) return m.inArena ? reinterpret_cast<const float*>(sc.ldsBase + m.offNormals) : m.normals;
    else if constexpr (SC::kLds) return reinterpret_cast<const float*>(sc.ldsBase + m.offNormals); else return m.normals;
}
template <class SC> TN_D const float* mesh_cdf(const SC& sc, const DevMesh& m)
{
    if constexpr (SC::kLds && SC::kMixed) return m.inArena ? reinterpret_cast<const float*>(sc.ldsBase + m.offCdf) : m.cdf;
    else if constexpr (SC::kLds) return reinterpret_cast<const float*>(sc.ldsBase + m.offCdf); else return m.cdf;
}

} // namespace tn
