// ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// A thin extern "C" driver around the *unmodified* reference CPU path.  It is
// compiled together with the reference's own sources where they lie under
// /root/reference/src (scene.cpp mesh.cpp loader.cpp pfm.cpp platform.cpp nlm.cpp
// png.cpp; render.cpp is #included below, unmodified, so that this translation unit
// sees the definition of `struct CpuRenderer`) by oracle/Makefile into
// oracle/_ref/libtinsel_ref.so.  Nothing here re-implements the integrator or the
// framebuffer splat: radiance comes from the reference's
//     Vec3 PathTrace(const Scene&, const Vec3&, const Vec3&, float, int, Random&)
// (reference src/render.cpp:230), the framebuffer from the reference's own compiled
//     CpuRenderer::AddSample (render.cpp:401-445)
// and the faithful loop from CreateCpuRenderer (render.cpp:528).
//
// What this file adds (all of it cited):
//   * the per-path seed contract  Random(i + j*W + passSeed[s])   (render.cu:940,1050-1052,1099)
//   * the camera-sample draw order of the CPU oracle              (render.cpp:476-484)
//   * scene-pack (de)serialisation so scenes travel without the reference loader
//   * leaf-function tables (reference inline functions evaluated on caller arrays)
//   * a restatement of AddSample kept ONLY to be compared with the compiled one (ref_add_sample_agrees, tests/test_oracle.py)
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load this.

// The reference's render.cpp, compiled HERE, unmodified, from where it lies: PathTrace (:230; declared in none of its headers),
// struct CpuRenderer with its AddSample (:390-524), CreateCpuRenderer (:528).  (oracle/Makefile leaves render.cpp out of the list of
// separately compiled sources for that reason.  It comes first: disney.h, which it includes, has no include guard.)
#include "render.cpp"

#include "render.h"
#include "intersection.h"
#include "util.h"
#include "sampler.h"
#include "loader.h"
#include "mesh.h"
#include "scene.h"
#include "nlm.h"
#include "png.h"
#include "pfm.h"

#include "../include/tinsel_hip.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <vector>


static_assert(sizeof(Primitive) == sizeof(tinsel_primitive), "Primitive mirror");
static_assert(sizeof(BVHNode) == sizeof(tinsel_bvh_node), "BVHNode mirror");
static_assert(sizeof(Camera) == sizeof(tinsel_camera), "Camera mirror");
static_assert(sizeof(Transform) == sizeof(tinsel_transform), "Transform mirror");
static_assert(sizeof(Options) == sizeof(tinsel_options), "Options mirror");
static_assert(sizeof(Material) == sizeof(tinsel_material), "Material mirror");
static_assert(sizeof(MeshGeometry) == sizeof(tinsel_mesh_geometry), "MeshGeometry mirror");
static_assert(offsetof(Primitive, mesh) == offsetof(tinsel_primitive, geo), "Primitive.geo");
static_assert(offsetof(Primitive, material) == offsetof(tinsel_primitive, material), "Primitive.material");
static_assert(offsetof(Primitive, lightSamples) == offsetof(tinsel_primitive, light_samples), "Primitive.lightSamples");
static_assert(offsetof(Options, maxDepth) == offsetof(tinsel_options, max_depth), "Options.maxDepth");

namespace {

struct RefScene
{
    Scene scene;
    Camera camera;
    Options options;
    std::vector<unsigned char> blob;    // backing store when built from a pack
    std::vector<Mesh*> extraMeshes;     // meshes made by the harness (stand-ins)
};

inline Random SeededRandom(uint32_t seed)
{
    // Random(int seed) (maths.h:1040-1044) without the signed-overflow UB
    Random r;
    r.seed1 = 315645664u + seed;
    r.seed2 = r.seed1 ^ 0x13ab45feu;
    return r;
}

// passSeed[s] = (s+1)-th output of Random(1).Rand()   (render.cu:1050-1052 `seed = Random(frame)`, :1099 `seed.Rand()`)
inline uint32_t PassSeed(uint32_t passIndex)
{
    Random r = SeededRandom(1);
    uint32_t v = 0;
    for (uint32_t i = 0; i <= passIndex; ++i)
        v = r.Rand();
    return v;
}

// A restatement of CpuRenderer::AddSample (render.cpp:401-445).  NOT what the oracle's framebuffer is made with (that is the
// reference's own compiled member, ref_render_seeded below): kept to be compared with it (ref_add_sample_agrees).
void AddSampleRestated(Color* output, int width, int height, float rasterX, float rasterY, float clamp, const Filter& filter, const Vec3& sample)
{
    int startX = Max(0, int(rasterX - filter.width));
    int startY = Max(0, int(rasterY - filter.width));
    int endX = Min(int(rasterX + filter.width), width - 1);
    int endY = Min(int(rasterY + filter.width), height - 1);

    Vec3 c = ClampLength(sample, clamp);

    for (int x = startX; x <= endX; ++x)
    {
        for (int y = startY; y <= endY; ++y)
        {
            if (filter.type == eFilterBox)
            {
                output[y*width + x] += Color(c, 1.0f);
            }
            else
            {
                float w = filter.Eval(x - rasterX, y - rasterY);
                output[y*width + x] += Color(c*w, w);
            }
        }
    }
}

void SetDefaults(RefScene* rs)
{
    // main.cpp:181-193
    rs->options.width = 512;
    rs->options.height = 256;
    rs->options.filter = Filter(eFilterGaussian, 0.75f, 1.0f);
    rs->options.mode = ePathTrace;
    rs->options.exposure = 1.0f;
    rs->options.limit = 1.5f;
    rs->options.clamp = FLT_MAX;
    rs->options.maxDepth = 4;
    rs->options.maxSamples = INT_MAX;

    rs->camera.position = Vec3(0.0f, 1.0f, 5.0f);
    rs->camera.rotation = Quat();
    rs->camera.fov = DegToRad(35.0f);
}

size_t Align16(size_t x) { return (x + 15) & ~size_t(15); }

// BVHBuilder leaves `rightIndex` of leaf nodes uninitialised (bvh.h:236-240): zero it in the serialised
// copy so packs are byte-reproducible (no consumer reads it for leaves)
std::vector<BVHNode> CleanNodes(const BVHNode* nodes, int n)
{
    std::vector<BVHNode> out(nodes, nodes + n);
    for (int i = 0; i < n; ++i)
        if (out[i].leaf)
            out[i].rightIndex = 0;
    return out;
}

} // namespace

extern "C" {

// ---------------------------------------------------------------------------
// scene lifetime

void* ref_scene_load_tin(const char* path)
{
    RefScene* rs = new RefScene();
    SetDefaults(rs);
    if (!LoadTin(path, &rs->scene, &rs->camera, &rs->options))
    {
        delete rs;
        return NULL;
    }
    rs->scene.Build();      // main.cpp:199
    fflush(stdout);
    return rs;
}

// Deterministic stand-in for the missing data/meshes/ajax.obj:
// CreateSphere(slices, segments) (mesh.cpp:1056-1100) radially displaced by a
// closed-form ripple, then the ImportMesh post-processing (mesh.cpp:120-129).
// Inserted at primitive index `insertAt` with the given material copied from
// primitive `materialFrom`... the caller passes the material explicitly.
int ref_scene_add_standin_mesh(void* h, int slices, int segments, float scale, const tinsel_material* material, int insertAt)
{
    RefScene* rs = (RefScene*)h;
    Mesh* mesh = CreateSphere(slices, segments, 1.0f);
    for (size_t i = 0; i < mesh->positions.size(); ++i)
    {
        Vec3 p = mesh->positions[i];
        float d = 1.0f + 0.08f*sinf(9.0f*p.x)*sinf(7.0f*p.y + 1.0f)*sinf(11.0f*p.z + 2.0f)
                       + 0.03f*sinf(31.0f*p.x + 0.5f)*sinf(29.0f*p.y)*sinf(37.0f*p.z)
                       + 0.25f*p.y*p.y;
        mesh->positions[i] = p*d;
    }
    mesh->Normalize();
    mesh->CalculateNormals();
    mesh->RebuildBVH();

    rs->extraMeshes.push_back(mesh);
    rs->scene.meshes.push_back(mesh);

    Primitive prim;
    prim.type = eMesh;
    prim.mesh = GeometryFromMesh(mesh);
    prim.startTransform = Transform(Vec3(0.0f), Quat(), scale);
    prim.endTransform = prim.startTransform;
    memcpy(&prim.material, material, sizeof(Material));
    prim.lightSamples = 0;

    if (insertAt < 0 || insertAt > (int)rs->scene.primitives.size())
        insertAt = (int)rs->scene.primitives.size();
    rs->scene.primitives.insert(rs->scene.primitives.begin() + insertAt, prim);

    delete[] rs->scene.bvh.nodes;
    rs->scene.bvh.nodes = NULL;
    rs->scene.Build();
    return 0;
}

// A REAL scanned mesh for data/ajax.tin (SURVEY.md 0.1's other option: ajax.obj is a missing blob, the largest shipped mesh is
// data/meshes/Aphrodite_from_jotero_com.obj, 106,846 triangles): the reference's own ImportMesh (mesh.cpp:105-132: importer, Normalize,
// CalculateNormals, RebuildBVH), then `subdivisions` rounds of 1 -> 4 midpoint subdivision (every edge's midpoint shared by the two
// triangles on it) and the same post-processing again -- Normalize, CalculateNormals, the reference's SAH BVHBuilder, RebuildCDF.  An
// IRREGULAR tree (a scan: triangle sizes over two orders of magnitude), unlike the stand-in's regular tessellation.
// `axisShift`: the file's axes cyclically shifted by that many places first (new[(i + shift) % 3] = old[i]: a proper rotation of the mesh data,
// so that the primitive keeps ajax.tin's identity rotation) -- the Aphrodite scan's up axis is z.
int ref_scene_add_obj_mesh(void* h, const char* path, int subdivisions, int axisShift, float scale, const tinsel_material* material, int insertAt)
{
    RefScene* rs = (RefScene*)h;
    Mesh* mesh = ImportMesh(path);
    if (!mesh)
        return -1;
    axisShift = ((axisShift % 3) + 3) % 3;
    if (axisShift)
        for (size_t i = 0; i < mesh->positions.size(); ++i)
        {
            const float old[3] = { mesh->positions[i].x, mesh->positions[i].y, mesh->positions[i].z };
            float nw[3];
            for (int c = 0; c < 3; ++c)
                nw[(c + axisShift) % 3] = old[c];
            mesh->positions[i] = Vec3(nw[0], nw[1], nw[2]);
        }
    for (int round = 0; round < subdivisions; ++round)
    {
        std::map<std::pair<int, int>, int> mid;
        std::vector<int> out;
        out.reserve(mesh->indices.size()*4);
        auto midpoint = [&](int a, int b) -> int {
            const std::pair<int, int> key(a < b ? a : b, a < b ? b : a);
            std::map<std::pair<int, int>, int>::iterator it = mid.find(key);
            if (it != mid.end())
                return it->second;
            const int idx = (int)mesh->positions.size();
            mesh->positions.push_back((mesh->positions[a] + mesh->positions[b])*0.5f);
            mid[key] = idx;
            return idx;
        };
        const size_t numTris = mesh->indices.size()/3;
        for (size_t t = 0; t < numTris; ++t)
        {
            const int a = mesh->indices[t*3 + 0], b = mesh->indices[t*3 + 1], c = mesh->indices[t*3 + 2];
            const int ab = midpoint(a, b), bc = midpoint(b, c), ca = midpoint(c, a);
            const int tri[12] = { a, ab, ca, ab, b, bc, ca, bc, c, ab, bc, ca };
            out.insert(out.end(), tri, tri + 12);
        }
        mesh->indices.swap(out);
    }
    if (subdivisions > 0 || axisShift)
    {
        mesh->normals.resize(mesh->positions.size());
        mesh->Normalize();
        mesh->CalculateNormals();
        mesh->RebuildBVH();
    }

    rs->extraMeshes.push_back(mesh);
    rs->scene.meshes.push_back(mesh);

    Primitive prim;
    prim.type = eMesh;
    prim.mesh = GeometryFromMesh(mesh);
    prim.startTransform = Transform(Vec3(0.0f), Quat(), scale);
    prim.endTransform = prim.startTransform;
    memcpy(&prim.material, material, sizeof(Material));
    prim.lightSamples = 0;

    if (insertAt < 0 || insertAt > (int)rs->scene.primitives.size())
        insertAt = (int)rs->scene.primitives.size();
    rs->scene.primitives.insert(rs->scene.primitives.begin() + insertAt, prim);

    delete[] rs->scene.bvh.nodes;
    rs->scene.bvh.nodes = NULL;
    rs->scene.Build();
    return (int)(mesh->indices.size()/3);
}

// A primitive moves: new start / end transforms, then the reference's own Scene::Build (scene.cpp:4-16: BVHBuilder over
// PrimitiveBounds, intersection.h:906-939) -- what main.cpp:318-327 gets by re-loading an animated scene file per batch frame.
int ref_scene_set_transform(void* h, int index, const tinsel_transform* start, const tinsel_transform* end)
{
    RefScene* rs = (RefScene*)h;
    if (index < 0 || index >= (int)rs->scene.primitives.size())
        return -1;
    memcpy(&rs->scene.primitives[index].startTransform, start, sizeof(Transform));
    memcpy(&rs->scene.primitives[index].endTransform, end, sizeof(Transform));
    delete[] rs->scene.bvh.nodes;
    rs->scene.bvh.nodes = NULL;
    rs->scene.Build();
    return 0;
}

// the scene BVH as Scene::Build left it (2P-1 nodes; returns the count, copies at most `max_nodes`)
int ref_scene_get_bvh(void* h, tinsel_bvh_node* out, int max_nodes)
{
    RefScene* rs = (RefScene*)h;
    const int n = rs->scene.bvh.numNodes;
    if (out)
        memcpy(out, rs->scene.bvh.nodes, sizeof(BVHNode)*(size_t)(n < max_nodes ? n : max_nodes));
    return n;
}

// Replace the sky probe by a procedural lat-long HDR (width x height) so probe
// sampling can be exercised without shipping a 20 MB .hdr.
// Probe::BuildCDF is the reference's own (probe.h:31-79).
int ref_scene_set_procedural_probe(void* h, int width, int height)
{
    RefScene* rs = (RefScene*)h;
    Probe& p = rs->scene.sky.probe;
    p.width = width;
    p.height = height;
    p.data = new Color[width*height];
    for (int j = 0; j < height; ++j)
    {
        for (int i = 0; i < width; ++i)
        {
            float u = (i + 0.5f)/width, v = (j + 0.5f)/height;
            // dim blue-grey dome + a warm "sun" blob + a cool "window"
            float sun = expf(-((u - 0.3f)*(u - 0.3f)*60.0f + (v - 0.25f)*(v - 0.25f)*120.0f));
            float win = (u > 0.6f && u < 0.75f && v > 0.35f && v < 0.5f) ? 1.0f : 0.0f;
            float base = 0.15f + 0.35f*(1.0f - v);
            p.data[j*width + i] = Color(base*0.8f + 40.0f*sun + 4.0f*win, base*0.9f + 32.0f*sun + 5.0f*win, base + 20.0f*sun + 6.0f*win, 1.0f);
        }
    }
    p.BuildCDF();
    return 0;
}

void ref_scene_free(void* h)
{
    RefScene* rs = (RefScene*)h;
    if (!rs)
        return;
    if (!rs->blob.empty())
    {
        // primitives point into the blob; nothing else to free
        delete[] rs->scene.bvh.nodes;
        rs->scene.bvh.nodes = NULL;
        rs->scene.meshes.clear();
    }
    delete rs;
}

void ref_scene_get(void* h, tinsel_camera* cam, tinsel_options* opt)
{
    RefScene* rs = (RefScene*)h;
    memcpy(cam, &rs->camera, sizeof(Camera));
    memcpy(opt, &rs->options, sizeof(Options));
}

int ref_scene_num_primitives(void* h) { return (int)((RefScene*)h)->scene.primitives.size(); }

void ref_scene_get_primitive(void* h, int i, tinsel_primitive* out)
{
    memcpy(out, &((RefScene*)h)->scene.primitives[i], sizeof(Primitive));
}

// Filter(type,width,falloff) constructor (render.h:15-19) -- gives tests the
// reference's own `offset` for explicitly constructed filters.
void ref_make_filter(int type, float width, float falloff, tinsel_filter* out)
{
    Filter f((FilterType)type, width, falloff);
    memcpy(out, &f, sizeof(Filter));
}

// ---------------------------------------------------------------------------
// scene packs

// Serialises the scene into one relocatable blob (layout: include/tinsel_hip.h).
// Returns bytes written, 0 on failure.
size_t ref_scene_write_pack(void* h, const char* path)
{
    RefScene* rs = (RefScene*)h;
    const Scene& s = rs->scene;

    std::vector<unsigned char> blob(sizeof(tinsel_pack_header), 0);

    auto append = [&](const void* data, size_t bytes) -> uint64_t {
        size_t off = Align16(blob.size());
        blob.resize(off + bytes, 0);
        if (bytes)
            memcpy(&blob[off], data, bytes);
        return (uint64_t)off;
    };

    tinsel_pack_header hdr;
    memset(&hdr, 0, sizeof(hdr));
    memcpy(hdr.magic, TINSEL_PACK_MAGIC, 8);
    hdr.version = 1;
    hdr.num_primitives = (uint32_t)s.primitives.size();
    hdr.num_bvh_nodes = (uint32_t)s.bvh.numNodes;

    std::vector<Primitive> prims(s.primitives.begin(), s.primitives.end());

    // dedupe meshes by MeshGeometry::id (util.h:20)
    std::vector<unsigned long> ids;
    std::vector<MeshGeometry> packed;
    for (size_t i = 0; i < prims.size(); ++i)
    {
        Primitive& p = prims[i];
        // bump maps are dead in the reference; never serialise host pointers
        p.material.bumpMap = Texture();
        // struct padding is uninitialised in the reference's constructors: zero it so packs are byte-reproducible
        {
            unsigned char* raw = (unsigned char*)&p;
            memset(raw + offsetof(Primitive, type) + sizeof(GeometryType), 0, offsetof(Primitive, mesh) - offsetof(Primitive, type) - sizeof(GeometryType));
            memset(raw + offsetof(Primitive, material) + offsetof(Material, transmission) + sizeof(float), 0,
                   offsetof(Material, bumpMap) - offsetof(Material, transmission) - sizeof(float));
            memset(raw + offsetof(Primitive, material) + offsetof(Material, bumpMap) + offsetof(Texture, depth) + sizeof(int), 0,
                   sizeof(Texture) - offsetof(Texture, depth) - sizeof(int));
            memset(raw + offsetof(Primitive, lightSamples) + sizeof(int), 0, sizeof(Primitive) - offsetof(Primitive, lightSamples) - sizeof(int));
        }
        if (p.type != eMesh)
        {
            // zero the unused tail of the geometry union so packs are byte-reproducible
            unsigned char* g = (unsigned char*)&p.mesh;
            size_t used = (p.type == eSphere) ? sizeof(SphereGeometry) : sizeof(PlaneGeometry);
            memset(g + used, 0, sizeof(MeshGeometry) - used);
            continue;
        }

        size_t k = 0;
        for (; k < ids.size(); ++k)
            if (ids[k] == p.mesh.id)
                break;

        if (k == ids.size())
        {
            MeshGeometry g = p.mesh;
            MeshGeometry o = g;
            o.positions = (const Vec3*)append(g.positions, sizeof(Vec3)*g.numVertices);
            o.normals = (const Vec3*)append(g.normals, sizeof(Vec3)*g.numVertices);
            o.indices = (const int*)append(g.indices, sizeof(int)*g.numIndices);
            std::vector<BVHNode> cleanMeshNodes = CleanNodes(g.nodes, g.numNodes);
            o.nodes = (const BVHNode*)append(cleanMeshNodes.data(), sizeof(BVHNode)*g.numNodes);
            o.cdf = (const float*)append(g.cdf, sizeof(float)*(g.numIndices/3));
            o.id = (unsigned long)(k + 1);
            ids.push_back(p.mesh.id);
            packed.push_back(o);
        }
        p.mesh = packed[k];
    }
    hdr.num_meshes = (uint32_t)packed.size();

    hdr.off_primitives = append(prims.empty() ? NULL : &prims[0], sizeof(Primitive)*prims.size());
    std::vector<BVHNode> cleanSceneNodes = CleanNodes(s.bvh.nodes, s.bvh.numNodes);
    hdr.off_bvh_nodes = append(cleanSceneNodes.data(), sizeof(BVHNode)*s.bvh.numNodes);

    const Probe& pr = s.sky.probe;
    if (pr.valid)
    {
        hdr.probe_width = pr.width;
        hdr.probe_height = pr.height;
        hdr.off_probe_data = append(pr.data, sizeof(Color)*pr.width*pr.height);
        hdr.off_probe_pdf_x = append(pr.pdfValuesX, sizeof(float)*pr.width*pr.height);
        hdr.off_probe_cdf_x = append(pr.cdfValuesX, sizeof(float)*pr.width*pr.height);
        hdr.off_probe_pdf_y = append(pr.pdfValuesY, sizeof(float)*pr.height);
        hdr.off_probe_cdf_y = append(pr.cdfValuesY, sizeof(float)*pr.height);
    }
    memcpy(&hdr.sky_horizon, &s.sky.horizon, sizeof(Vec3));
    memcpy(&hdr.sky_zenith, &s.sky.zenith, sizeof(Vec3));
    memcpy(&hdr.camera, &rs->camera, sizeof(Camera));
    memcpy(&hdr.options, &rs->options, sizeof(Options));

    hdr.total_bytes = Align16(blob.size());
    blob.resize(hdr.total_bytes, 0);
    memcpy(&blob[0], &hdr, sizeof(hdr));

    FILE* f = fopen(path, "wb");
    if (!f)
        return 0;
    size_t n = fwrite(&blob[0], 1, blob.size(), f);
    fclose(f);
    return n;
}

void* ref_scene_load_pack(const void* data, size_t size)
{
    if (size < sizeof(tinsel_pack_header))
        return NULL;

    RefScene* rs = new RefScene();
    rs->blob.assign((const unsigned char*)data, (const unsigned char*)data + size);
    unsigned char* base = &rs->blob[0];

    tinsel_pack_header hdr;
    memcpy(&hdr, base, sizeof(hdr));
    if (memcmp(hdr.magic, TINSEL_PACK_MAGIC, 8) != 0 || hdr.version != 1 || hdr.total_bytes > size)
    {
        delete rs;
        return NULL;
    }

    const Primitive* prims = (const Primitive*)(base + hdr.off_primitives);
    for (uint32_t i = 0; i < hdr.num_primitives; ++i)
    {
        Primitive p = prims[i];
        if (p.type == eMesh)
        {
            p.mesh.positions = (const Vec3*)(base + (size_t)p.mesh.positions);
            p.mesh.normals = (const Vec3*)(base + (size_t)p.mesh.normals);
            p.mesh.indices = (const int*)(base + (size_t)p.mesh.indices);
            p.mesh.nodes = (const BVHNode*)(base + (size_t)p.mesh.nodes);
            p.mesh.cdf = (const float*)(base + (size_t)p.mesh.cdf);
        }
        rs->scene.primitives.push_back(p);
    }

    rs->scene.bvh.numNodes = (int)hdr.num_bvh_nodes;
    rs->scene.bvh.nodes = new BVHNode[hdr.num_bvh_nodes];
    memcpy(rs->scene.bvh.nodes, base + hdr.off_bvh_nodes, sizeof(BVHNode)*hdr.num_bvh_nodes);

    memcpy(&rs->scene.sky.horizon, &hdr.sky_horizon, sizeof(Vec3));
    memcpy(&rs->scene.sky.zenith, &hdr.sky_zenith, sizeof(Vec3));
    if (hdr.off_probe_data)
    {
        Probe& pr = rs->scene.sky.probe;
        pr.width = hdr.probe_width;
        pr.height = hdr.probe_height;
        pr.data = (Color*)(base + hdr.off_probe_data);
        pr.pdfValuesX = (float*)(base + hdr.off_probe_pdf_x);
        pr.cdfValuesX = (float*)(base + hdr.off_probe_cdf_x);
        pr.pdfValuesY = (float*)(base + hdr.off_probe_pdf_y);
        pr.cdfValuesY = (float*)(base + hdr.off_probe_cdf_y);
        pr.valid = true;
    }
    memcpy(&rs->camera, &hdr.camera, sizeof(Camera));
    memcpy(&rs->options, &hdr.options, sizeof(Options));
    return rs;
}

// ---------------------------------------------------------------------------
// the per-path-seeded oracle

// For every pass s in [passBegin, passBegin+numPasses) and every pixel (i,j) of
// the window [x0,x1) x [y0,y1):
//     Random rand(i + j*W + passSeed[s]);  Sample2D(x,y); Sample1D(t);   (render.cpp:476-477)
//     time = Lerp(shutterStart, shutterEnd, t);  x += i; y += j;          (render.cpp:479-482)
//     GenerateRay; sample = PathTrace(...);                               (render.cpp:484-486)
// then AddSample in raster order, pass by pass (render.cpp:490).
// `accum` (W*H*4, +=) and `radiance` (numPasses*winW*winH*3, =) may each be NULL.
// Returns wall seconds spent in the trace phase.
double ref_render_seeded(void* h, const tinsel_camera* cam_, const tinsel_options* opt_,
                         uint32_t passBegin, uint32_t numPasses,
                         int x0, int y0, int x1, int y1,
                         float* accum, float* radiance, int numThreads)
{
    RefScene* rs = (RefScene*)h;
    Camera camera;
    Options options;
    memcpy(&camera, cam_, sizeof(Camera));
    memcpy(&options, opt_, sizeof(Options));

    const int W = options.width, H = options.height;
    if (x1 <= x0 || y1 <= y0) { x0 = 0; y0 = 0; x1 = W; y1 = H; }
    const int winW = x1 - x0, winH = y1 - y0;

    CameraSampler sampler(Transform(camera.position, camera.rotation), camera.fov, 0.001f, 1.0f, W, H);   // render.cpp:450-456

    if (numThreads < 1)
        numThreads = 1;

    std::vector<Vec3> samples((size_t)winW*winH);
    std::vector<Vec2> rasters((size_t)winW*winH);

    double traceSeconds = 0.0;

    for (uint32_t s = 0; s < numPasses; ++s)
    {
        const uint32_t passSeed = PassSeed(passBegin + s);

        auto worker = [&](int tid) {
            CameraSampler local = sampler;      // GenerateRay is non-const
            for (int j = y0 + tid; j < y1; j += numThreads)
            {
                for (int i = x0; i < x1; ++i)
                {
                    Random rand = SeededRandom((uint32_t)i + (uint32_t)j*(uint32_t)W + passSeed);

                    float x, y, t;
                    Sample2D(rand, x, y);
                    Sample1D(rand, t);

                    float time = Lerp(camera.shutterStart, camera.shutterEnd, t);
                    x += i;
                    y += j;

                    Vec3 origin, dir;
                    local.GenerateRay(x, y, origin, dir);

                    Vec3 sample = PathTrace(rs->scene, origin, dir, time, options.maxDepth, rand);

                    size_t k = (size_t)(j - y0)*winW + (i - x0);
                    samples[k] = sample;
                    rasters[k] = Vec2(x, y);
                }
            }
        };

        auto t0 = std::chrono::steady_clock::now();
        if (numThreads == 1)
        {
            worker(0);
        }
        else
        {
            std::vector<std::thread> threads;
            for (int t = 0; t < numThreads; ++t)
                threads.emplace_back(worker, t);
            for (auto& th : threads)
                th.join();
        }
        auto t1 = std::chrono::steady_clock::now();
        traceSeconds += std::chrono::duration<double>(t1 - t0).count();

        if (radiance)
            memcpy(radiance + (size_t)s*winW*winH*3, &samples[0], sizeof(Vec3)*samples.size());

        if (accum)
        {
            // the reference's own compiled CpuRenderer::AddSample (render.cpp:401-445), in raster order like render.cpp:462-490
            CpuRenderer splat(&rs->scene);
            Color* out = (Color*)accum;
            for (int j = y0; j < y1; ++j)
                for (int i = x0; i < x1; ++i)
                {
                    size_t k = (size_t)(j - y0)*winW + (i - x0);
                    splat.AddSample(out, W, H, rasters[k].x, rasters[k].y, options.clamp, options.filter, samples[k]);
                }
        }
    }
    return traceSeconds;
}

// The faithful single-threaded reference loop: CreateCpuRenderer + `passes` x Render()
// exactly as main.cpp:246-250 drives it (serial RNG stream, render.cpp:399).
// `out` must hold W*H*4 floats and is accumulated into (+=), like render.cpp:418,439.
double ref_render_faithful(void* h, const tinsel_camera* cam_, const tinsel_options* opt_, int passes, float* out)
{
    RefScene* rs = (RefScene*)h;
    Camera camera;
    Options options;
    memcpy(&camera, cam_, sizeof(Camera));
    memcpy(&options, opt_, sizeof(Options));

    Renderer* r = CreateCpuRenderer(&rs->scene);
    r->Init(options.width, options.height);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < passes; ++i)
        r->Render(camera, options, (Color*)out);
    auto t1 = std::chrono::steady_clock::now();
    delete r;
    return std::chrono::duration<double>(t1 - t0).count();
}

// ---------------------------------------------------------------------------
// leaf-function tables: the reference's own inline functions on caller arrays

// Random(seed): n successive Rand() and Randf() outputs (maths.h:1036-1091)
void ref_leaf_random(uint32_t seed, int n, uint32_t* outRand, float* outRandf)
{
    Random a = SeededRandom(seed);
    Random b = SeededRandom(seed);
    for (int i = 0; i < n; ++i)
    {
        outRand[i] = a.Rand();
        outRandf[i] = b.Randf();
    }
}

uint32_t ref_pass_seed(uint32_t passIndex) { return PassSeed(passIndex); }

// CameraSampler + GenerateRay (util.h:45-79)
void ref_leaf_camera_rays(const tinsel_camera* cam_, int W, int H, int n, const float* rasterXY, float* outOriginDir)
{
    Camera camera;
    memcpy(&camera, cam_, sizeof(Camera));
    CameraSampler sampler(Transform(camera.position, camera.rotation), camera.fov, 0.001f, 1.0f, W, H);
    for (int i = 0; i < n; ++i)
    {
        Vec3 o, d;
        sampler.GenerateRay(rasterXY[i*2 + 0], rasterXY[i*2 + 1], o, d);
        memcpy(outOriginDir + i*6, &o, 12);
        memcpy(outOriginDir + i*6 + 3, &d, 12);
    }
}

// BSDFEval / BSDFPdf (disney.h:125-166, 296-405); inputs per row: n(3) V(3) L(3) etaI etaO
void ref_leaf_bsdf_eval(const tinsel_material* mat_, int n, const float* in, float* outF, float* outPdf)
{
    Material mat;
    memcpy(&mat, mat_, sizeof(Material));
    for (int i = 0; i < n; ++i)
    {
        const float* r = in + i*11;
        Vec3 N(r[0], r[1], r[2]), V(r[3], r[4], r[5]), L(r[6], r[7], r[8]);
        Vec3 f = BSDFEval(mat, r[9], r[10], Vec3(0.0f), N, V, L);
        float pdf = BSDFPdf(mat, r[9], r[10], Vec3(0.0f), N, V, L);
        memcpy(outF + i*3, &f, 12);
        outPdf[i] = pdf;
    }
}

// BSDFSample (disney.h:170-293) after BasisFromVector (maths.h:1261-1275);
// inputs per row: n(3) V(3) etaI etaO seed ; outputs: L(3) pdf type randsConsumedState(2 as float bits)
void ref_leaf_bsdf_sample(const tinsel_material* mat_, int n, const float* in, const uint32_t* seeds, float* outL, float* outPdf, int32_t* outType, uint32_t* outRngState)
{
    Material mat;
    memcpy(&mat, mat_, sizeof(Material));
    for (int i = 0; i < n; ++i)
    {
        const float* r = in + i*8;
        Vec3 N(r[0], r[1], r[2]), V(r[3], r[4], r[5]);
        Vec3 u, v;
        BasisFromVector(N, &u, &v);
        Random rand = SeededRandom(seeds[i]);
        Vec3 L(0.0f);
        float pdf = 0.0f;
        BSDFType type = eReflected;
        BSDFSample(mat, r[6], r[7], Vec3(0.0f), u, v, N, V, L, pdf, type, rand);
        memcpy(outL + i*3, &L, 12);
        outPdf[i] = pdf;
        outType[i] = (int)type;
        outRngState[i*2 + 0] = rand.seed1;
        outRngState[i*2 + 1] = rand.seed2;
    }
}

// PrimitiveIntersect (intersection.h:951-1020) for primitive `prim` of the scene;
// inputs per row: origin(3) dir(3) time ; outputs: hit t n(3)
void ref_leaf_primitive_intersect(void* h, int prim, int n, const float* in, int32_t* outHit, float* outT, float* outN)
{
    RefScene* rs = (RefScene*)h;
    const Primitive& p = rs->scene.primitives[prim];
    for (int i = 0; i < n; ++i)
    {
        const float* r = in + i*7;
        Ray ray(Vec3(r[0], r[1], r[2]), Vec3(r[3], r[4], r[5]), r[6]);
        float t = 0.0f;
        Vec3 nrm(0.0f);
        bool hit = PrimitiveIntersect(p, ray, t, &nrm);
        outHit[i] = hit ? 1 : 0;
        outT[i] = hit ? t : 0.0f;
        if (!hit) nrm = Vec3(0.0f);
        memcpy(outN + i*3, &nrm, 12);
    }
}

// PrimitiveSample (intersection.h:855-904); per row: time seed ; outputs pos(3) normal(3) rng state
void ref_leaf_primitive_sample(void* h, int prim, int n, const float* times, const uint32_t* seeds, float* outPos, float* outN, uint32_t* outRngState)
{
    RefScene* rs = (RefScene*)h;
    const Primitive& p = rs->scene.primitives[prim];
    for (int i = 0; i < n; ++i)
    {
        Random rand = SeededRandom(seeds[i]);
        Vec3 pos, nrm;
        PrimitiveSample(p, times[i], pos, nrm, rand);
        memcpy(outPos + i*3, &pos, 12);
        memcpy(outN + i*3, &nrm, 12);
        outRngState[i*2 + 0] = rand.seed1;
        outRngState[i*2 + 1] = rand.seed2;
    }
}

// ProbeSample / ProbePdf / Sky::Eval (probe.h:128-236, scene.h:168-178)
void ref_leaf_probe(void* h, int n, const uint32_t* seeds, float* outDir, float* outColor, float* outPdf, float* outPdfOfDir, float* outEvalOfDir)
{
    RefScene* rs = (RefScene*)h;
    const Sky& sky = rs->scene.sky;
    for (int i = 0; i < n; ++i)
    {
        Random rand = SeededRandom(seeds[i]);
        Vec3 dir(0.0f), color(0.0f);
        float pdf = 0.0f;
        if (sky.probe.valid)
            ProbeSample(sky.probe, dir, color, pdf, rand);
        else
            dir = UniformSampleSphere(rand.Randf(), rand.Randf());
        memcpy(outDir + i*3, &dir, 12);
        memcpy(outColor + i*3, &color, 12);
        outPdf[i] = pdf;
        outPdfOfDir[i] = sky.probe.valid ? ProbePdf(sky.probe, dir) : 0.0f;
        Vec3 e = sky.Eval(dir);
        memcpy(outEvalOfDir + i*3, &e, 12);
    }
}

// ---------------------------------------------------------------------------
// display stage: the per-frame post-processing of main.cpp:258-282 and the image writers

// g_filtered[i] = LinearToSrgb(ToneMap(g_pixels[i]*(exposure/w), limit))  -- main.cpp:262-271
void ref_present(const float* pixels, int numPixels, float exposure, float limit, float* filtered)
{
    const Color* in = (const Color*)pixels;
    Color* out = (Color*)filtered;
    for (int i = 0; i < numPixels; ++i)
    {
        float s = exposure/in[i].w;
        out[i] = LinearToSrgb(ToneMap(in[i]*s, limit));
    }
}

// NonLocalMeansFilter (nlm.cpp:35-77), as main.cpp:273-277 calls it
void ref_nlm(const float* in, float* out, int width, int height, float falloff, int radius)
{
    NonLocalMeansFilter((const Color*)in, (Color*)out, width, height, falloff, radius);
}

// WritePng (png.cpp:323-371) / PfmSave (pfm.cpp:70-85) to `path`
void ref_write_png(const float* pixels, int width, int height, const char* path)
{
    WritePng((const Color*)pixels, width, height, path);
}

void ref_pfm_save(const float* rgb, int width, int height, const char* path)
{
    PfmImage image;
    memset(&image, 0, sizeof(image));
    image.width = width;
    image.height = height;
    image.depth = 1;
    image.data = const_cast<float*>(rgb);
    PfmSave(path, image);
    fflush(NULL);       // PfmSave never closes its FILE (pfm.cpp:70-85)
}

// the reference's own PrimitiveBounds (intersection.h:906-939): the leaf box Scene::Build gives the scene BVH builder
void ref_primitive_bounds(void* h, int prim, float* out6)
{
    const Bounds b = PrimitiveBounds(((RefScene*)h)->scene.primitives[(size_t)prim]);
    out6[0] = b.lower.x; out6[1] = b.lower.y; out6[2] = b.lower.z;
    out6[3] = b.upper.x; out6[4] = b.upper.y; out6[5] = b.upper.z;
}

int ref_hardware_threads() { return (int)std::thread::hardware_concurrency(); }

// The restated AddSample against the reference's compiled CpuRenderer::AddSample (render.cpp:401-445): `n` seeded random samples
// (raster positions over and around a W x H frame, radiance up to 8 so that `clamp` bites) splatted with both into two buffers; returns
// the number of floats that differ (0 expected).  filterType: eFilterBox / eFilterGaussian.
int ref_add_sample_agrees(int W, int H, int filterType, float filterWidth, float filterFalloff, float clamp, uint32_t seed, int n)
{
    std::vector<Color> a((size_t)W*H), b((size_t)W*H);
    Filter filter((FilterType)filterType, filterWidth, filterFalloff);
    CpuRenderer compiled(nullptr);
    Random rand = SeededRandom(seed);
    for (int k = 0; k < n; ++k)
    {
        const float x = rand.Randf(-2.0f, (float)W + 2.0f), y = rand.Randf(-2.0f, (float)H + 2.0f);
        const Vec3 c(rand.Randf(0.0f, 8.0f), rand.Randf(0.0f, 8.0f), rand.Randf(0.0f, 8.0f));
        compiled.AddSample(&a[0], W, H, x, y, clamp, filter, c);
        AddSampleRestated(&b[0], W, H, x, y, clamp, filter, c);
    }
    int bad = 0;
    for (size_t i = 0; i < a.size(); ++i)
        bad += memcmp(&a[i], &b[i], sizeof(Color)) != 0;
    return bad;
}

} // extern "C"
