// hip_renderer.cpp -- the reference-side binding: what a Tinsel maintainer adds to make
//     Renderer* CreateGpuRenderer(const Scene* s)            (reference src/render.h:79)
// return the MI355X back-end instead of the CUDA one (reference src/render.cu:978-1110).
//
// Compiled by g++ against the REFERENCE headers (-I$(REF)/src) and linked with
// tinsel_amd/libtinsel_hip.so; it is the only translation unit that sees both worlds.
// `Scene` holds std::vectors (scene.h:186-190), so it is walked HERE and handed down as the
// flat tinsel_scene_desc; the POD records cross the C-ABI unchanged (same layouts, asserted below).

#include "render.h"
#include "scene.h"

#include "../include/tinsel_hip.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

static_assert(sizeof(Primitive) == sizeof(tinsel_primitive), "Primitive layout");
static_assert(sizeof(BVHNode) == sizeof(tinsel_bvh_node), "BVHNode layout");
static_assert(sizeof(Camera) == sizeof(tinsel_camera), "Camera layout");
static_assert(sizeof(Options) == sizeof(tinsel_options), "Options layout");
static_assert(sizeof(Color) == 4*sizeof(float), "Color layout");

struct HipRenderer : public Renderer
{
    // Every visible GPU of the node behind the one Renderer the caller asked for (TINSEL_HIP_NUM_GPUS limits it): pixel
    // tiles sharded over the devices, one RCCL reduce of the accumulation buffer per Render (include/tinsel_hip.h,
    // tinsel_hip_group_*).  With one device this is exactly one tinsel_hip.
    tinsel_hip_group* group;

    HipRenderer(const Scene* s) : group(NULL)
    {
        tinsel_scene_desc d = {};
        d.primitives = (const tinsel_primitive*)&s->primitives[0];
        d.num_primitives = (int32_t)s->primitives.size();
        d.bvh_nodes = (const tinsel_bvh_node*)s->bvh.nodes;
        d.num_bvh_nodes = s->bvh.numNodes;
        d.sky_horizon = { s->sky.horizon.x, s->sky.horizon.y, s->sky.horizon.z };
        d.sky_zenith = { s->sky.zenith.x, s->sky.zenith.y, s->sky.zenith.z };
        const Probe& p = s->sky.probe;
        if (p.valid)
        {
            d.probe_valid = 1;
            d.probe_width = p.width;
            d.probe_height = p.height;
            d.probe_data = (const tinsel_vec4*)p.data;
            d.probe_pdf_x = p.pdfValuesX;
            d.probe_cdf_x = p.cdfValuesX;
            d.probe_pdf_y = p.pdfValuesY;
            d.probe_cdf_y = p.cdfValuesY;
        }
        const char* n = getenv("TINSEL_HIP_NUM_GPUS");
        group = tinsel_hip_group_create(&d, n ? atoi(n) : 0, 64);
        if (!group && !(n && atoi(n) == 1))
        {
            // several GPUs visible but no group (RCCL missing, communicator refused, a device busy): one GPU still renders
            fprintf(stderr, "CreateGpuRenderer: %s -- falling back to one GPU\n", tinsel_hip_last_error());
            group = tinsel_hip_group_create(&d, 1, 64);
        }
        if (!group)
            fprintf(stderr, "CreateGpuRenderer: %s\n", tinsel_hip_last_error());
        else
        {
            if (tinsel_hip_group_size(group) > 1)
                fprintf(stderr, "CreateGpuRenderer: %d GPUs\n", tinsel_hip_group_size(group));
            // main.cpp:246-250 calls Render() once per pass with a full-frame read-back: the next calls are traced (and, with
            // several GPUs, reduced) while this one's image is copied out.  The caller's array is NOT page-locked by default:
            // main.cpp:73-87 frees it before Init.  TINSEL_HIP_PIN_OUTPUT=1 for callers whose array outlives the renderer.
            if (!getenv("TINSEL_HIP_NO_LOOKAHEAD"))
                tinsel_hip_group_set_lookahead(group, getenv("TINSEL_HIP_PIN_OUTPUT") ? TINSEL_LOOKAHEAD_PIN_OUTPUT : TINSEL_LOOKAHEAD_ON);
        }
    }

    virtual ~HipRenderer() { tinsel_hip_group_destroy(group); }

    // Renderer::Init (render.h:70): allocate + zero the accumulators (render.cu:1070-1075)
    virtual void Init(int width, int height)
    {
        if (group && tinsel_hip_group_init(group, width, height))
            fprintf(stderr, "HipRenderer::Init: %s\n", tinsel_hip_last_error());
    }

    // Renderer::Render (render.h:71): one more sample per pixel; `output` receives the running
    // sum (rgb*w, w) like render.cu:1102.  On failure `output` is left untouched (the reference
    // reports no errors either; the message is on stderr and in tinsel_hip_last_error()).
    virtual void Render(const Camera& camera, const Options& options, Color* output)
    {
        if (group && tinsel_hip_group_render(group, (const tinsel_camera*)&camera, (const tinsel_options*)&options, (float*)output, 1))
            fprintf(stderr, "HipRenderer::Render: %s\n", tinsel_hip_last_error());
    }
};

Renderer* CreateGpuRenderer(const Scene* s)
{
    return new HipRenderer(s);
}

// beyond the reference interface: N passes per call without the per-pass full-frame D2H
// (SURVEY.md 8b "API-compat mitigation"); used by the headless driver.
extern "C" int HipRendererRenderPasses(Renderer* r, const Camera& camera, const Options& options, Color* output, int passes)
{
    HipRenderer* h = static_cast<HipRenderer*>(r);
    return h->group ? tinsel_hip_group_render(h->group, (const tinsel_camera*)&camera, (const tinsel_options*)&options, (float*)output, passes) : -1;
}

// The display stage of main.cpp:258-282 on the device accumulator: `filtered` receives what main.cpp calls
// g_filtered (or g_exposed when nlmWidth != 0) -- the array it hands to glDrawPixels and WritePng.
extern "C" int HipRendererPresent(Renderer* r, const Options& options, Color* filtered, int nlmWidth, float nlmFalloff)
{
    HipRenderer* h = static_cast<HipRenderer*>(r);
    return h->group ? tinsel_hip_group_present(h->group, (const tinsel_options*)&options, nlmWidth, nlmFalloff, (float*)filtered) : -1;
}

// The NEXT FRAME of an animation on the renderer the caller already has.  The reference's batch mode (main.cpp:314-327) deletes its renderer
// and re-runs Init per frame: loader, Scene::Build, a new GpuRenderer -- every mesh converted and uploaded again.  When `next` is `prev` (the
// scene this renderer was created from, or last updated to) with other primitive transforms -- a rigid animation -- only the moved primitives'
// records are rewritten (tinsel_hip_set_primitive_transform) and the scene level is rebuilt from next->bvh, the reference's own Scene::Build of
// that frame (tinsel_hip_rebuild_scene): bit-identical to a renderer created from `next`.  The pass index goes back to 0, where a fresh renderer
// starts.  Returns 0 when done; 1 when the frames differ in more than transforms (nothing changed: delete + CreateGpuRenderer, as the reference
// does); -1 on error.
// two trees node by node.  Not a memcmp: a leaf's rightIndex bits are whatever the builder's heap held (bvh.h:9-20 stores the item in
// leftIndex and sets `leaf`; nothing ever reads the other 31 bits), and they differ from one load of the same file to the next.
static bool same_nodes(const BVHNode* a, const BVHNode* b, int n)
{
    for (int i = 0; i < n; ++i)
    {
        if (memcmp(&a[i].bounds, &b[i].bounds, sizeof(Bounds)) != 0 || a[i].leftIndex != b[i].leftIndex || a[i].leaf != b[i].leaf)
            return false;
        if (!a[i].leaf && a[i].rightIndex != b[i].rightIndex)
            return false;
    }
    return true;
}

static const char* mesh_difference(const MeshGeometry& a, const MeshGeometry& b)
{
    if (a.numVertices != b.numVertices || a.numIndices != b.numIndices || a.numNodes != b.numNodes)
        return "mesh topology";
    if (a.area != b.area)
        return "mesh area";
    if (memcmp(a.positions, b.positions, sizeof(Vec3)*a.numVertices) != 0) return "mesh vertex positions";
    if (memcmp(a.normals, b.normals, sizeof(Vec3)*a.numVertices) != 0) return "mesh vertex normals";
    if (memcmp(a.indices, b.indices, sizeof(int)*a.numIndices) != 0) return "mesh indices";
    if (!same_nodes(a.nodes, b.nodes, a.numNodes)) return "mesh BVH";
    if (memcmp(a.cdf, b.cdf, sizeof(float)*(a.numIndices/3)) != 0) return "mesh area CDF";
    return NULL;
}

// what keeps `next` from being `prev` with other primitive transforms (NULL: nothing); *prim: the primitive, or -1
static const char* scene_difference(const Scene* prev, const Scene* next, int* prim)
{
    *prim = -1;
    const size_t P = prev->primitives.size();
    if (next->primitives.size() != P)
        return "number of primitives";
    if (memcmp(&prev->sky.horizon, &next->sky.horizon, sizeof(Vec3)) != 0 || memcmp(&prev->sky.zenith, &next->sky.zenith, sizeof(Vec3)) != 0)
        return "sky";
    const Probe &pa = prev->sky.probe, &pb = next->sky.probe;
    if (pa.valid != pb.valid || (pa.valid && (pa.width != pb.width || pa.height != pb.height || memcmp(pa.data, pb.data, sizeof(Color)*pa.width*pa.height) != 0)))
        return "probe";
    for (size_t i = 0; i < P; ++i)
    {
        const Primitive &a = prev->primitives[i], &b = next->primitives[i];
        *prim = (int)i;
        if (a.type != b.type)
            return "primitive type";
        if (a.lightSamples != b.lightSamples)
            return "lightSamples";
        // (the material's parameters: bytes [0, 84) and [112, 128) -- between them 4 bytes of padding, whatever the loader's stack held, and
        // Material::bumpMap, a host pointer + sizes of a texture the integrator never reads: bump mapping is dead code in the reference)
        const char *ma = (const char*)&a.material, *mb = (const char*)&b.material;
        if (memcmp(ma, mb, 84) != 0 || memcmp(ma + 112, mb + 112, sizeof(Material) - 112) != 0)
            return "material";
        if (a.type == eSphere && a.sphere.radius != b.sphere.radius)
            return "sphere radius";
        if (a.type == ePlane && memcmp(a.plane.plane, b.plane.plane, sizeof(a.plane.plane)) != 0)
            return "plane equation";
        if (a.type == eMesh)
            if (const char* why = mesh_difference(a.mesh, b.mesh))
                return why;
    }
    *prim = -1;
    return NULL;
}

extern "C" int HipRendererUpdateScene(Renderer* r, const Scene* prev, const Scene* next)
{
    HipRenderer* h = static_cast<HipRenderer*>(r);
    if (!h->group || !prev || !next)
        return -1;
    int where = -1;
    if (const char* why = scene_difference(prev, next, &where))
    {
        if (where >= 0)
            fprintf(stderr, "HipRendererUpdateScene: the frames differ in more than transforms (primitive %d: %s)\n", where, why);
        else
            fprintf(stderr, "HipRendererUpdateScene: the frames differ in more than transforms (%s)\n", why);
        return 1;
    }
    const size_t P = prev->primitives.size();
    const int n = tinsel_hip_group_size(h->group);
    for (int k = 0; k < n; ++k)
    {
        tinsel_hip* m = tinsel_hip_group_member(h->group, k);
        for (size_t i = 0; i < P; ++i)
        {
            const Primitive &a = prev->primitives[i], &b = next->primitives[i];
            if (memcmp(&a.startTransform, &b.startTransform, sizeof(Transform)) == 0 && memcmp(&a.endTransform, &b.endTransform, sizeof(Transform)) == 0)
                continue;
            if (tinsel_hip_set_primitive_transform(m, (int)i, (const tinsel_transform*)&b.startTransform, (const tinsel_transform*)&b.endTransform))
                return -1;
        }
        if (tinsel_hip_rebuild_scene(m, TINSEL_SCENE_BVH_NODES, (const tinsel_bvh_node*)next->bvh.nodes, next->bvh.numNodes, NULL) ||
            tinsel_hip_set_pass_index(m, 0))
            return -1;
    }
    return 0;
}

extern "C" int HipRendererNumGpus(Renderer* r)
{
    HipRenderer* h = static_cast<HipRenderer*>(r);
    return h->group ? tinsel_hip_group_size(h->group) : 0;
}
