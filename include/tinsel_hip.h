/*
 * tinsel_hip.h -- C-ABI of the MI355X (gfx950) path-tracing back-end for Tinsel.
 *
 * This is the drop-in boundary. It replaces what the reference does behind
 *     Renderer* CreateGpuRenderer(const Scene* s);            (reference src/render.h:79)
 *     struct Renderer { Init(w,h); Render(cam, opts, out); }  (reference src/render.h:66-73)
 * which the reference implements with CUDA in src/render.cu:978-1110.
 *
 * Everything that crosses this ABI is plain pointers + sizes.  The POD records
 * below are *layout mirrors* of the reference structs (sizes/offsets asserted at
 * the bottom of this file), so a Tinsel maintainer passes `&scene->primitives[0]`,
 * `scene->bvh.nodes`, `&camera`, `&options` unchanged -- see INTEGRATION.md for
 * the ~40-line `HipRenderer : Renderer` shim (shim/hip_renderer.cpp).
 *
 * No torch types, no C++ types, no exceptions cross this boundary.  All entry
 * points return 0 on success / non-zero on failure (or NULL for constructors) and
 * leave a message retrievable with tinsel_hip_last_error().
 */
#ifndef TINSEL_HIP_H
#define TINSEL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------- */
/* POD mirrors of the reference structs                                       */

typedef struct tinsel_vec3 { float x, y, z; } tinsel_vec3;          /* maths.h:214-234  (12 B) */
typedef struct tinsel_vec4 { float x, y, z, w; } tinsel_vec4;       /* maths.h:290-312  (16 B), Color/Quat alike */

typedef struct tinsel_transform {                                   /* maths.h:575-589  (32 B) */
    tinsel_vec3 p;
    tinsel_vec4 r;      /* quaternion x,y,z,w */
    float s;
} tinsel_transform;

typedef struct tinsel_bvh_node {                                    /* bvh.h:9-20       (32 B) */
    tinsel_vec3 lower;
    tinsel_vec3 upper;
    uint32_t left_index;            /* leaf: item index */
    uint32_t right_index_leaf;      /* bits 0..30 = rightIndex, bit 31 = leaf */
} tinsel_bvh_node;

typedef struct tinsel_camera {                                      /* scene.h:11-30    (40 B) */
    tinsel_vec3 position;
    tinsel_vec4 rotation;
    float fov;
    float shutter_start;
    float shutter_end;
} tinsel_camera;

typedef struct tinsel_texture {                                     /* scene.h:33-42    (24 B) */
    float* data;
    int32_t width, height, depth;
    int32_t _pad;
} tinsel_texture;

typedef struct tinsel_material {                                    /* scene.h:45-100   (128 B) */
    tinsel_vec3 emission;
    tinsel_vec3 color;
    tinsel_vec3 absorption;
    float eta;
    float metallic;
    float subsurface;
    float specular;
    float roughness;
    float specular_tint;
    float anisotropic;
    float sheen;
    float sheen_tint;
    float clearcoat;
    float clearcoat_gloss;
    float transmission;
    int32_t _pad0;
    tinsel_texture bump_map;
    float bump;
    tinsel_vec3 bump_tile;
} tinsel_material;

enum { TINSEL_GEOM_SPHERE = 0, TINSEL_GEOM_PLANE = 1, TINSEL_GEOM_MESH = 2 };   /* scene.h:102-107 */

typedef struct tinsel_mesh_geometry {                               /* scene.h:119-135  (64 B) */
    const tinsel_vec3* positions;
    const tinsel_vec3* normals;
    const int32_t* indices;
    const tinsel_bvh_node* nodes;
    const float* cdf;
    int32_t num_vertices;
    int32_t num_indices;
    int32_t num_nodes;
    float area;
    uint64_t id;
} tinsel_mesh_geometry;

typedef struct tinsel_primitive {                                   /* scene.h:138-159  (272 B) */
    tinsel_transform start_transform;
    tinsel_transform end_transform;
    int32_t type;
    int32_t _pad0;
    union {
        struct { float radius; } sphere;
        struct { float plane[4]; } plane;
        tinsel_mesh_geometry mesh;
    } geo;
    tinsel_material material;
    int32_t light_samples;
    int32_t _pad1;
} tinsel_primitive;

enum { TINSEL_FILTER_BOX = 0, TINSEL_FILTER_GAUSSIAN = 1 };          /* render.h:7-11 */

typedef struct tinsel_filter {                                      /* render.h:13-39   (16 B) */
    int32_t type;
    float width;
    float falloff;
    float offset;       /* taken verbatim from the caller, never recomputed (loader quirk, loader.cpp:74) */
} tinsel_filter;

enum { TINSEL_MODE_NORMALS = 0, TINSEL_MODE_COMPLEXITY = 1, TINSEL_MODE_PATHTRACE = 2 }; /* render.h:42-47 */

typedef struct tinsel_options {                                     /* render.h:50-63   (48 B) */
    int32_t mode;
    int32_t width;
    int32_t height;
    tinsel_filter filter;
    float exposure;
    float limit;
    float clamp;
    int32_t max_depth;
    int32_t max_samples;
} tinsel_options;

/* Flat view of a reference `Scene` (scene.h:183-215).  `Scene` itself holds
 * std::vectors, so the C++ side of the shim walks it and fills this struct;
 * nothing behind this ABI touches libstdc++ containers. */
typedef struct tinsel_scene_desc {
    const tinsel_primitive* primitives;     /* &scene->primitives[0]; mesh pointers are HOST pointers */
    int32_t num_primitives;
    int32_t num_bvh_nodes;                  /* scene->bvh.numNodes */
    const tinsel_bvh_node* bvh_nodes;       /* scene->bvh.nodes (built by Scene::Build, scene.cpp:4-16) */
    tinsel_vec3 sky_horizon;                /* scene->sky.horizon  (scene.h:163) */
    tinsel_vec3 sky_zenith;                 /* scene->sky.zenith */
    int32_t probe_valid;                    /* scene->sky.probe.valid (probe.h:81) */
    int32_t probe_width;
    int32_t probe_height;
    int32_t _pad;
    const tinsel_vec4* probe_data;          /* probe.data,  width*height RGBA32F */
    const float* probe_pdf_x;               /* probe.pdfValuesX  [w*h] */
    const float* probe_cdf_x;               /* probe.cdfValuesX  [w*h] */
    const float* probe_pdf_y;               /* probe.pdfValuesY  [h]   */
    const float* probe_cdf_y;               /* probe.cdfValuesY  [h]   */
} tinsel_scene_desc;

/* ------------------------------------------------------------------------- */
/* The renderer object                                                        */

typedef struct tinsel_hip tinsel_hip;       /* opaque */

/* Pipeline selection (tinsel_hip_set_pipeline).  WAVEFRONT is the product path: one streaming
 * kernel per bounce over HBM ray queues with wave64 compaction.  MEGAKERNEL (one lane per whole
 * path) and WAVEFRONT_SPLIT (extend / shade / shadow kernels per bounce) are A/B arms with
 * identical per-path arithmetic. */
enum { TINSEL_PIPELINE_WAVEFRONT = 0, TINSEL_PIPELINE_MEGAKERNEL = 1, TINSEL_PIPELINE_WAVEFRONT_SPLIT = 2,
       /* PAIRED: the split pipeline re-cut for scenes with meshes in HBM -- a bounce's shadow rays and the next bounce's extension rays are
        * walked in ONE k_walk launch, ONE streaming kernel per bounce does the rest (tn_paired.h); same arithmetic, same bits */
       TINSEL_PIPELINE_WAVEFRONT_PAIRED = 4,
       /* default: WAVEFRONT (fused bounce kernel) when the whole scene is LDS-resident; else WAVEFRONT_PAIRED where every mesh is walked by
        * k_walk, nothing moves and no light is a large mesh (measured: DESIGN.md section 5), else WAVEFRONT_SPLIT */
       TINSEL_PIPELINE_AUTO = 3 };

/* Replaces GpuRenderer::GpuRenderer (render.cu:989-1053): deep-copies the scene
 * to device `device_index` (dedupes meshes by MeshGeometry::id, re-lays BVHs out
 * for 64-B two-child records, pre-gathers triangles).  The desc and everything it
 * points to may be freed once this returns. */
tinsel_hip* tinsel_hip_create(const tinsel_scene_desc* scene, int device_index);

/* Tuning: every choice between two code paths of the library that a scene or a batch size normally decides, as ONE plain
 * struct (no reference counterpart: the reference has one kernel and no choices).  Nothing here changes a result -- every
 * setting renders the same image bit for bit (tests/test_gpu_switches.py drives all of them) -- and nothing is read from the
 * caller's ENVIRONMENT: a production caller gets the defaults (tinsel_hip_create), a test or an A/B run fills the struct.
 * Convention: -1 (or 0 where stated) = "the library decides"; struct_bytes = sizeof(tinsel_hip_tuning) as the CALLER compiled
 * it (a library built against a longer struct takes the defaults for the fields the caller does not have). */
typedef struct tinsel_hip_tuning {
    uint32_t struct_bytes;
    /* ---- where the scene's data lives and how its scene level is scanned: read by tinsel_hip_create_tuned only ---- */
    int32_t flat_scan;          /* -1 auto | 0: scene level by BVH walk even where the flat scan could take it (<= 64 primitives) */
    int32_t lds_scene;          /* -1 auto | 0: the scene arena stays in HBM (never staged into LDS) */
    int32_t walk;               /* -1 auto | 0: meshes in HBM are walked inline by k_extend / k_shadow, no k_walk */
    int32_t inline_max_tris;    /* -1 default | triangles up to which a mesh rides in the arena beside a mesh in HBM */
    int32_t walk_min_tris;      /* -1 default | triangle count from which k_walk takes a mesh in HBM */
    int64_t small_mesh_bytes;   /* -1 default | size up to which a mesh rides in the arena (0: every mesh lives in HBM) */
    int64_t arena_lds_limit;    /* -1 default | arena size up to which it is staged into LDS */
    /* ---- per render: tinsel_hip_set_tuning changes them at any time between two renders ---- */
    int64_t batch_paths;        /* 0 default | path slots resident per batch (>= 65536; tinsel_hip_set_batch_paths sets the same field) */
    int32_t grid_mult;          /* 0 default (32) | upper bound of the streaming grid in workgroups per CU */
    int32_t bounce_share;       /* -1 auto | 0 / 1: k_bounce deals its workgroup's four regions to its waves as one stream: never / always */
    int32_t repack;             /* -1 auto | 0 / 1: k_bounce's per-wave shading pools: never / always (where the LDS allows) */
    int32_t tail_split;         /* -1 auto | 0 off | 1: the last tail_share of a batch in regions 1/tail_divide as long */
    float tail_share;
    int32_t tail_divide;
    int32_t shade_sorted;       /* -1 auto | 0 / 1: k_shade / k_shade_sorted */
    int32_t overlap;            /* -1 auto | 0 / 1: a batch's passes as two overlapped chunks on two streams: never / wherever it has two passes */
    int32_t scene_walk;         /* -1 auto | 0: scenes beyond the flat scan through k_extend / k_shadow instead of k_swalk */
    int32_t swalk_lds;          /* -1 auto | 0: k_swalk's 256-thread generic-pointer variant */
    int32_t accumulate;         /* 0 auto | TINSEL_ACCUMULATE_TILED / _WIDE / _PIPED (filter widths up to 1) */
    int32_t walk_block;         /* 0 auto | 256 / 1024: k_walk's workgroup size */
    int32_t walk_single;        /* -1 auto | 0: per-lane tree pointers (k_walk_rays) also for ONE walked primitive */
    int32_t walk_lds_stack;     /* -1 default (8) | stack entries per lane kept in LDS (0: all of them, one workgroup per CU) */
    int32_t walk_refill_min;    /* 0 default (24) | idle lanes of a wave that trigger k_walk's refill */
    int32_t walk_leaf_min;      /* 0 default (8) | lanes waiting at a leaf that trigger k_walk's triangle phase */
    int32_t walk_grid_mult;     /* 0 default (1) | k_walk's grid in resident sets of workgroups: each workgroup a contiguous 1/grid of the work list */
    int32_t quads_in_scan;      /* -1 auto | 0: a quad (two-triangle mesh: a lamp) beside meshes walked by k_walk sends the scene to the general scan kernels (inline mesh walk compiled in) */
} tinsel_hip_tuning;
enum { TINSEL_ACCUMULATE_AUTO = 0, TINSEL_ACCUMULATE_TILED = 1, TINSEL_ACCUMULATE_WIDE = 2, TINSEL_ACCUMULATE_PIPED = 3 };

/* Fills `t` with the defaults ("the library decides" everywhere). */
void tinsel_hip_tuning_init(tinsel_hip_tuning* t);
/* tinsel_hip_create with a tuning (NULL: the defaults, i.e. exactly tinsel_hip_create). */
tinsel_hip* tinsel_hip_create_tuned(const tinsel_scene_desc* scene, int device_index, const tinsel_hip_tuning* tuning);
/* The per-render fields of `tuning` from the next render on (the create-time fields are ignored: they are baked into the
 * uploaded scene).  Frees the path buffers (they are sized by grid_mult), drops any look-ahead. */
int tinsel_hip_set_tuning(tinsel_hip* r, const tinsel_hip_tuning* tuning);
/* The tuning in force (create-time fields as given to create, per-render fields as last set). */
int tinsel_hip_get_tuning(tinsel_hip* r, tinsel_hip_tuning* out);

/* Replaces GpuRenderer::~GpuRenderer (render.cu:1055-1068). */
void tinsel_hip_destroy(tinsel_hip* r);

/* Replaces GpuRenderer::Init (render.cu:1070-1075): (re)allocates and zeroes the
 * W*H float4 accumulation buffer. */
int tinsel_hip_init(tinsel_hip* r, int width, int height);

/* Same, but the accumulator is caller-owned DEVICE memory of W*H*4 floats (e.g. a
 * torch tensor, so the host language can hand it to RCCL); it is zeroed here and
 * must outlive the renderer or the next init. */
int tinsel_hip_init_external(tinsel_hip* r, int width, int height, float* device_accum);

/* Replaces GpuRenderer::Render (render.cu:1077-1103) with `passes` == 1: adds one
 * sample per pixel per pass to the device accumulator, then copies the running sum
 * (rgb*w, w) to `out_rgba` (W*H*4 floats, host memory).  `out_rgba` may be NULL
 * to skip the D2H copy (the accumulator stays resident in HBM). */
int tinsel_hip_render(tinsel_hip* r, const tinsel_camera* camera, const tinsel_options* options,
                      float* out_rgba, int passes);

/* Look-ahead for callers that use tinsel_hip_render the way the reference's main loop does (main.cpp:246-250: one pass and
 * the full-frame running sum to the host per call, 16 calls per displayed frame).  With it on, a tinsel_hip_render call
 * returns as soon as ITS running sum has been copied out, and meanwhile the passes the next call will most probably ask
 * for (same camera, options and pass count) are traced into a second accumulator; a matching next call only swaps
 * buffers and copies, anything else drops the speculation.  Images are bit-identical to the plain path; the statistics
 * counters and tinsel_hip_read_batch_radiance see one call ahead (every other entry point drops the speculation first).
 * Off by default; the C++ shim turns it on (TINSEL_LOOKAHEAD_ON).
 * TINSEL_LOOKAHEAD_PIN_OUTPUT additionally page-locks the caller's output array IN PLACE (hipHostRegister) so that the
 * read-back is an asynchronous DMA: only for callers who guarantee that the array outlives the renderer, the next
 * tinsel_hip_init or the next tinsel_hip_set_lookahead -- the registration is dropped there, and registered memory must
 * not be freed before (the reference's own caller frees its array before Init, main.cpp:73-87: it gets plain ON). */
enum { TINSEL_LOOKAHEAD_OFF = 0, TINSEL_LOOKAHEAD_ON = 1, TINSEL_LOOKAHEAD_PIN_OUTPUT = 2 };
int tinsel_hip_set_lookahead(tinsel_hip* r, int enable);

/* Same, but never touches host memory: enqueues `passes` passes on `stream`
 * (a hipStream_t, may be NULL for the default stream) and returns without
 * synchronising.  Used by bench.py / multi-GPU hosts that own the stream. */
int tinsel_hip_render_async(tinsel_hip* r, const tinsel_camera* camera, const tinsel_options* options,
                            int passes, void* stream);

/* Device pointer to the W*H*4 float accumulation buffer (for an RCCL reduce by the
 * host language, e.g. torch.distributed).  Valid until the next init/destroy. */
float* tinsel_hip_accum_device_ptr(tinsel_hip* r);

/* Blocking copy of the accumulator to host (W*H*4 floats). */
int tinsel_hip_read_accum(tinsel_hip* r, float* out_rgba);

/* Mesh BVHs.  TINSEL_BVH_REFERENCE (default): the trees the reference's host builder made (bvh.h:30-263), as
 * passed in tinsel_mesh_geometry::nodes -- the parity path, since a tree's visit order resolves exact-t ties.
 * TINSEL_BVH_LBVH: rebuild every mesh that does not live in LDS on the DEVICE (Morton-code linear BVH, one
 * triangle per leaf like the reference's trees; tn_lbvh.h) -- for meshes rebuilt per frame (main.cpp:318-327
 * re-inits per batch frame) or too large to wait for the host SAH sweep.  Same hits except exact ties; slower
 * to traverse than the SAH tree, far faster to build.  TINSEL_BVH_PLOC: the same, the hierarchy by parallel locally-ordered
 * clustering over the Morton order (agglomerative, surface-area driven: close to the SAH tree in render speed, a few ms to build).
 * *build_ms (may be NULL) receives the device build time. */
enum { TINSEL_BVH_REFERENCE = 0, TINSEL_BVH_LBVH = 1, TINSEL_BVH_PLOC = 2 };
int tinsel_hip_set_mesh_bvh(tinsel_hip* r, int mode, double* build_ms);

/* Refit for deforming meshes (per-frame vertex animation with unchanged topology; the reference rebuilds with its host SAH
 * sweep, mesh.cpp:314-338 -- 1.1 s for 524k triangles): `primitive`'s mesh gets new vertex positions (num_vertices x 3
 * floats, host; must be the vertex count the mesh was created with) and optionally new vertex normals; the triangle
 * records are re-gathered and every box of the mesh's CURRENT tree (the reference's, or a device-built one) is recomputed
 * bottom-up on the device -- exactly the boxes the reference's builder would store for that tree shape; the area CDF and
 * PrimitiveArea of every instance follow (Mesh::RebuildCDF's serial order).  All instances of the mesh change.  Meshes
 * small enough for the LDS-staged arena are refused (create a new renderer: it costs less than the refit). */
int tinsel_hip_refit_mesh(tinsel_hip* r, int primitive, const float* positions_xyz, int num_vertices, const float* normals_xyz);

/* Primitives that MOVE between renders, and the scene-level BVH that follows them.  The reference mutates Scene::primitives and
 * re-runs Scene::Build (scene.cpp:4-16: its host BVHBuilder over PrimitiveBounds(p), intersection.h:906-939), or re-loads an animated
 * scene file and re-creates the renderer per batch frame (main.cpp:318-327) -- re-uploading every mesh.  Here:
 *   tinsel_hip_set_primitive_transform   new start / end transforms of primitive `index` (the pose used by every intersection, the
 *       motion-blur interpolation when they differ, PrimitiveArea of a mesh light).  Takes effect with the next rebuild_scene: until
 *       then the renderer must not be asked to render.
 *   tinsel_hip_rebuild_scene             the scene BVH, the primitives' leaf boxes and the traversal stack depth, from PrimitiveBounds
 *       of the primitives as they are NOW.  TINSEL_SCENE_BVH_NODES: `nodes` (2P-1 reference-format nodes, e.g. the reference's own
 *       Scene::Build run on the host) is validated like at create and used as it is -- bit-identical to a renderer created from the moved
 *       scene.  TINSEL_SCENE_BVH_DEVICE: built on the device (the mesh builders' kernels, tn_lbvh.h: Morton order over the boxes'
 *       centroids, agglomerative clustering by surface area); nodes / num_nodes are ignored.  At the scene level the reference's
 *       QueryBVH has no closest-t cull (intersection.h:751-799), so the SET of primitives a ray tests does not depend on the tree:
 *       another tree changes the visit order only, which decides nothing but exact-t ties.  *build_ms (may be NULL): device time.
 * Meshes keep their own trees (tinsel_hip_refit_mesh / tinsel_hip_set_mesh_bvh for those). */
enum { TINSEL_SCENE_BVH_NODES = 0, TINSEL_SCENE_BVH_DEVICE = 1 };
int tinsel_hip_set_primitive_transform(tinsel_hip* r, int index, const tinsel_transform* start, const tinsel_transform* end);
int tinsel_hip_rebuild_scene(tinsel_hip* r, int mode, const tinsel_bvh_node* nodes, int num_nodes, double* build_ms);

/* Importance sampling of the environment probe.  TINSEL_PROBE_CDF (default): the reference's two binary searches over
 * the row and column CDFs (ProbeSample, probe.h:205-236; ~21 dependent loads per sample on the 1600x800 loft.hdr) --
 * sample-identical to the reference.  TINSEL_PROBE_ALIAS (opt-in): an alias table over the W*H texels built once on the
 * host from the same pdf tables; consumes the same two random numbers, draws texels with the same probabilities and
 * returns the same pdf for a texel, so the estimator is as unbiased -- but a given seed picks another texel, so images
 * agree with the reference's statistically, not sample for sample. */
enum { TINSEL_PROBE_CDF = 0, TINSEL_PROBE_ALIAS = 1 };
int tinsel_hip_set_probe_sampling(tinsel_hip* r, int mode);

/* Russian roulette, OPT-IN (start_bounce = 0, the default, is the reference's behaviour: render.cpp:250 runs every
 * path to maxDepth and so does the parity path).  With start_bounce = b > 0, after every bounce i >= b - 1 that has a
 * successor the path survives with probability q = min(1, max(throughput.rgb)) and its throughput is divided by q: the
 * estimate stays unbiased, deep low-throughput paths (glass at maxDepth 12) stop early.  Draws one extra number from
 * the path's stream when q < 1, so images differ from the reference's sample by sample; the same rule is restated in
 * the C oracle (port_set_russian_roulette) and the two are compared bit for bit. */
int tinsel_hip_set_russian_roulette(tinsel_hip* r, int start_bounce);

/* Resume a progressive render: replaces the accumulator with a saved one (W*H*4 floats, as read_accum returned
 * it) and sets the index of the next pass, so that `read_accum after k passes` + `write_accum(.., k)` + more passes
 * gives the same image bit for bit as one uninterrupted render (pass seeds depend on the pass index only). */
int tinsel_hip_write_accum(tinsel_hip* r, const float* rgba, uint32_t next_pass_index);

/* ---- display stage: what main.cpp does with the accumulator every frame (main.cpp:258-282) ----
 *
 * g_filtered[i] = LinearToSrgb(ToneMap(g_pixels[i]*(exposure/g_pixels[i].w), limit))   util.h:25-42, maths.h:1545-1555
 * and, when nlm_width != 0, NonLocalMeansFilter(g_filtered, g_exposed, W, H, nlm_falloff, nlm_width)  (nlm.cpp:35-77;
 * main.cpp:59-60 defaults: width 0 = off, falloff 200).  Runs on the device accumulator (the sum of all passes since
 * Init); options->mode != ePathTrace presents the raw accumulator, as the reference does.  The float image is
 * bit-identical to the host code's (same operation order; powf/expf as glibc 2.35 evaluates them).
 * out_rgba (W*H*4 floats, the array main.cpp hands to glDrawPixels / WritePng) may be NULL. */
int tinsel_hip_present(tinsel_hip* r, const tinsel_options* options, int nlm_width, float nlm_falloff, float* out_rgba);
int tinsel_hip_present_async(tinsel_hip* r, const tinsel_options* options, int nlm_width, float nlm_falloff, void* stream);
/* Device pointer to the most recently presented W*H*4 float image. */
const float* tinsel_hip_present_device_ptr(tinsel_hip* r);

/* WritePng's float -> 8-bit RGB conversion (png.cpp:323-343): x*255 + Randf + Randf - 0.5 from ONE default-seeded
 * serial Random stream over all pixels and channels, clamped and truncated.  Host code (the generator has no
 * skip-ahead); rgb receives W*H*3 bytes, top row first -- the payload of the PNG the reference writes. */
int tinsel_image_quantize_rgb8(const float* rgba, int width, int height, unsigned char* rgb);

/* Pixel-tile sharding for multi-GPU: this renderer traces only paths whose
 * generating pixel lies in a tile t with (t % world) == rank, tiles of
 * `tile`x`tile` pixels in raster order.  Seeds depend on (pixel, pass) only, so
 * the sum over ranks of the accumulators equals the world==1 image up to float
 * summation order.  Default: rank 0 of 1. */
int tinsel_hip_set_shard(tinsel_hip* r, int rank, int world, int tile);

int tinsel_hip_set_pipeline(tinsel_hip* r, int pipeline);

/* Arithmetic contract of the path kernels.
 * TINSEL_ARITH_EXACT (default, the parity path): no FMA contraction, IEEE divide / sqrt, glibc 2.35's sinf / cosf / expf
 *   restated -- every path bit-identical to the reference's CPU PathTrace on the same seeds.
 * TINSEL_ARITH_FAST (opt-in): the same kernels built the way the reference builds itself (`-O3 -ffast-math`, makefile:4;
 *   `-use_fast_math -prec-div=false -prec-sqrt=false`, tinsel.vcxproj:134): FMA contraction, v_rcp / v_rsq / v_sqrt,
 *   the hardware's sin / cos / exp.  Same seeds, same RNG streams; paths agree to rounding until a branch flips, and the
 *   image stays within the stated 1e-3 per-pixel L2 of the CPU reference at the spp the BASELINE configs use (measured by
 *   tests/test_gpu_fast.py and reported by bench.py as fast_l2).  Not bit-reproducible against the reference. */
enum { TINSEL_ARITH_EXACT = 0, TINSEL_ARITH_FAST = 1 };
int tinsel_hip_set_arithmetic(tinsel_hip* r, int mode);
int tinsel_hip_get_arithmetic(tinsel_hip* r);

/* The per-pass seed is the (pass_index+1)-th output of Random(1).Rand()
 * (mirrors `seed = Random(frame)` / `seed.Rand()`, render.cu:1050-1052,1099).
 * A new renderer starts at pass 0; Init does not reset it (nor does the reference). */
int tinsel_hip_set_pass_index(tinsel_hip* r, uint32_t pass_index);
uint32_t tinsel_hip_get_pass_index(tinsel_hip* r);

/* Counters since creation (or the last reset): rays = Trace()-equivalent casts
 * (extension + shadow; reference render.cpp:17 call sites :253,:122,:175),
 * samples = camera paths started, gpu_seconds = sum of HIP-event-timed spans. */
void tinsel_hip_stats(tinsel_hip* r, unsigned long long* rays, unsigned long long* samples,
                      double* gpu_seconds);
void tinsel_hip_reset_stats(tinsel_hip* r);

/* Per-kernel timing (HIP events on the launch stream) of the most recent
 * tinsel_hip_render* call, for bench.py's roofline block.  Returns the number
 * of kernel classes written (<= max_entries).  Blocks until the events resolve. */
typedef struct tinsel_kernel_time {
    char name[32];
    uint32_t launches;
    float total_ms;     /* sum of the launches' durations */
    float busy_ms;      /* union of their intervals: less than total_ms where a call's chunks overlap on two streams */
} tinsel_kernel_time;
int tinsel_hip_kernel_times(tinsel_hip* r, tinsel_kernel_time* out, int max_entries);
/* sizeof(tinsel_kernel_time) in THIS build of the library (40; 36 before busy_ms was added in round 4): a caller that may be handed an
 * older library (TINSEL_HIP_LIB) asks before it reads the records -- the symbol's absence says "36-byte records, no busy_ms". */
int tinsel_hip_kernel_time_bytes(void);
int tinsel_hip_enable_kernel_timing(tinsel_hip* r, int enable);

/* Extended counters since the last reset: [0]=rays [1]=samples [2]=internal BVH node visits
 * [3]=triangle tests [4]=primitive tests [5]=shadow rays [6..7] reserved.  [2..4] only advance
 * while detail counting is on (it costs a few percent); they feed B_ray of DESIGN.md. */
int tinsel_hip_stats_detail(tinsel_hip* r, unsigned long long* out8);
int tinsel_hip_set_detail_counters(tinsel_hip* r, int enable);

/* Allocates the per-batch path buffers a later render of `passes` passes at `max_depth` will need, so that the first
 * such call does not pay for hipMalloc (tinsel_hip_render* allocate on demand otherwise). */
int tinsel_hip_reserve(tinsel_hip* r, int passes, int max_depth);

/* Upper bound on path slots resident per batch (default 64 Mi for the wavefront pipelines, 8 Mi for the megakernel arm); the
 * same field as tinsel_hip_tuning::batch_paths. */
int tinsel_hip_set_batch_paths(tinsel_hip* r, unsigned long long max_paths);

/* Per-path radiance of the most recent batch (test hook): copies min(max_paths, paths in batch)
 * float4 records {r,g,b,-} in slot order (slot = pass_in_batch*W*H + j*W + i) and returns the count. */
long long tinsel_hip_read_batch_radiance(tinsel_hip* r, float* out_rgbx, unsigned long long max_paths);

/* Test hook: evaluates one device leaf function on caller arrays (host pointers; rows of `in_stride` /
 * `out_stride` floats, one thread per row) so tests can table the HIP restatements against the reference's
 * inline functions.  op: 0 Random, 1 CameraSampler::GenerateRay, 2 BSDFEval+BSDFPdf, 3 BSDFSample,
 * 4 PrimitiveIntersect, 5 PrimitiveSample, 6 ProbeSample/ProbePdf/Sky::Eval, 7 libm restatements,
 * 8 display-stage functions (row layouts: tn_kernels.h LeafOp).
 * `index` = primitive (its material for ops 2,3). */
int tinsel_hip_leaf(tinsel_hip* r, int op, int index, int n, const float* in, int in_stride, const uint32_t* seeds,
                    float* out, int out_stride, const tinsel_camera* camera, int width, int height);

/* Introspection: LDS traversal-stack entries per lane chosen for this scene, NEE rays per bounce. */
int tinsel_hip_stack_entries(tinsel_hip* r);
int tinsel_hip_nee_per_path(tinsel_hip* r);
/* Primitives whose mesh BVH is walked by the dedicated k_walk kernel ahead of the scan kernels (large meshes in HBM,
 * split pipeline; tn_walk.h).  0: every mesh is walked inline, as IntersectRayMesh is called in the reference. */
int tinsel_hip_walked_prims(tinsel_hip* r);
/* Queue lengths of the LAST batch of the wavefront pipelines: out[b] = paths alive at the start of bounce b (entries of the
 * extension queue; bounce 0: the paths generated, tile padding of a shard included), out[max_bounces + b] = paths with
 * shadow rays at bounce b (split pipeline; 0 otherwise).  Synchronises.
 * Returns the number of bounces written (<= max_bounces), or -1. */
int tinsel_hip_queue_counts(tinsel_hip* r, uint32_t* out, int max_bounces);

const char* tinsel_hip_last_error(void);

/* ------------------------------------------------------------------------- */
/* All GPUs of one node behind ONE Renderer (tinsel_hip_create(scene, num_gpus); north_star: pixel   */
/* tiles sharded over the 8 GPUs with an RCCL reduce of the float4 accumulation buffer).                            */
/*                                                                                                                   */
/* The reference's caller is one single-threaded C++ loop (main.cpp:207 creates the renderer, :246-250 calls         */
/* Render): a group gives that caller N devices with no second process and no Python.  One host thread per device    */
/* drives a full tinsel_hip renderer of its own (scene uploaded to every device, rank-local path slots: set_shard);  */
/* every member traces the paths of ITS pixel tiles for every pass of a call, seeds depend on (pixel, pass) only, so  */
/* the sum over members of the accumulators is the single-GPU image up to float summation order.  That sum is ONE    */
/* ncclReduce(SUM, float, 4*W*H) over xGMI into a buffer on member 0 per read-back (the members' own accumulators    */
/* are never modified by it, so repeated calls cannot double-count), followed by the D2H copy.  RCCL is loaded with  */
/* dlopen on first use (librccl.so.1): a single-GPU user of this library needs no RCCL at all.                       */

typedef struct tinsel_hip_group tinsel_hip_group;      /* opaque */

/* Replaces `new GpuRenderer(scene)` for `num_gpus` devices (0 = every visible device), devices 0..num_gpus-1, pixel
 * tiles of `tile` x `tile` (0 = 64) dealt round-robin.  num_gpus == 1 is exactly one tinsel_hip (no RCCL, no threads'
 * worth of difference in the image).  Fails (NULL + tinsel_hip_last_error) when fewer devices are visible -- except
 * under TINSEL_HIP_GROUP_ONE_DEVICE=1, a VALIDATION switch for single-GPU boxes: all members share device 0 and the
 * reduce is a device-local sum in rank order instead of the RCCL call (threads, shards, slots and read-back as real). */
tinsel_hip_group* tinsel_hip_group_create(const tinsel_scene_desc* scene, int num_gpus, int tile);
/* The same with a tuning for every member (NULL: the defaults). */
tinsel_hip_group* tinsel_hip_group_create_tuned(const tinsel_scene_desc* scene, int num_gpus, int tile, const tinsel_hip_tuning* tuning);
void tinsel_hip_group_destroy(tinsel_hip_group* g);
/* Renderer::Init on every member (+ the reduce target on member 0). */
int tinsel_hip_group_init(tinsel_hip_group* g, int width, int height);
/* Renderer::Render: `passes` more samples per pixel of the WHOLE frame (each member its tiles), then -- when out_rgba
 * is not NULL -- reduce + copy the running sum of everything since Init to out_rgba (W*H*4 floats, host). */
int tinsel_hip_group_render(tinsel_hip_group* g, const tinsel_camera* camera, const tinsel_options* options, float* out_rgba, int passes);
/* Look-ahead for the reference's call pattern (tinsel_hip_set_lookahead) with N members: after a read-back every member
 * goes on with the calls that will most probably follow (same camera, options, pass count) -- a batch of up to 16 calls of
 * ITS shard traced at once, one snapshot accumulator per call -- and the NEXT call's snapshots are reduced (one
 * ncclReduce, each member from its own thread and stream) while this call's sum crosses PCIe.  A matching call waits
 * for that reduce (done, as a rule), swaps two buffers and copies; anything else drops the speculation.  Images are
 * those of the plain path bit for bit.  `enable` as for tinsel_hip_set_lookahead (a group of one forwards to it). */
int tinsel_hip_group_set_lookahead(tinsel_hip_group* g, int enable);
/* The display stage (tinsel_hip_present) on the reduced accumulator, run on member 0. */
int tinsel_hip_group_present(tinsel_hip_group* g, const tinsel_options* options, int nlm_width, float nlm_falloff, float* out_rgba);
int tinsel_hip_group_size(tinsel_hip_group* g);
/* Member `rank`'s renderer, for the introspection / statistics entry points (do not render or init through it).
 * Whatever the group had speculated is dropped first (look-ahead starts over with the next read-back). */
tinsel_hip* tinsel_hip_group_member(tinsel_hip_group* g, int rank);

/* One process per GPU (an MPI-style host: bench.py --gpus N under torch.distributed.run): the SAME collective as the group's -- one
 * ncclReduce(SUM, float, 4*W*H) of the accumulators to `root` -- with the communicator made by ncclCommInitRank.  The host language only
 * carries the 128-byte id from rank 0 to the other ranks (a torch.distributed broadcast, MPI_Bcast, a file); set_shard(rank, world, tile)
 * is the caller's, as before.  comm_init is collective (every rank calls it); comm_size = ncclCommCount of the communicator (0: none);
 * comm_reduce_accum enqueues on `stream`, WAITS for it, and leaves the sum in `out_device` (W*H*4 floats of device memory) on `root`
 * only -- this rank's own accumulator is never written, so a later render + reduce cannot count a sample twice. */
#define TINSEL_HIP_COMM_ID_BYTES 128
int tinsel_hip_comm_unique_id(unsigned char* id_bytes, int capacity);
int tinsel_hip_comm_init(tinsel_hip* r, const unsigned char* id_bytes, int rank, int world);
int tinsel_hip_comm_size(tinsel_hip* r);
int tinsel_hip_comm_reduce_accum(tinsel_hip* r, float* out_device, int root, void* stream);

/* How a batch of `slots` path slots is cut into regions on a device of `num_cus` CUs (the dense path state of the wavefront pipelines:
 * DESIGN.md section 4; `fused` != 0: the fused pipeline, which ends a batch with shorter regions).  Pure host arithmetic -- no device is
 * touched: out[6] = number of regions, positions per region, regions of that length (the rest are short), positions per short region,
 * workgroups of a launch, capacity of the region arrays.  Introspection for the tests; no reference counterpart. */
int tinsel_hip_plan_regions(unsigned long long slots, int num_cus, int nee_per_path, int fused, unsigned int* out);

/* Exhaustive self-test of the parity arm's short reciprocal / square-root sequences (tn_math.h rcp_candidate / sqrt_candidate):
 * compares the candidate with the compiler's correctly rounded `1.0f/x` (op 0), `sqrtf(x)` (op 1) or `1.0f/sqrtf(x)` (op 2) on ALL 2^32 fp32 bit patterns
 * on the device.  variant < 0: the variant this library's kernels are built with (0 = the compiler's own expansion).
 * out_counts[260]: [0..3] mismatches in total / with a denormal operand / with |x| >= 2^126 (op 0) or x < 0 (ops 1, 2) / any
 * other; [4 + e] mismatches by the operand's exponent field e;
 * out_first_bad: the smallest mismatching bit pattern (0xffffffff when none).  No reference counterpart (test infrastructure
 * of this library: the reference divides with the host FPU / nvcc's IEEE division, render.cpp / maths.h throughout). */
int tinsel_hip_selftest_arith(int device_index, int op, int variant, unsigned long long* out_counts, unsigned int* out_first_bad);
/* The library's own stable radix sort and exclusive scan (the device BVH builder's, tinsel_amd/csrc/tn_sort.h) on caller data, for tests:
 * keys[0, n) sorted in place by their bits [begin_bit, end_bit) (multiples of 8; keys with equal bits keep their order);
 * out[i] = in[0] + .. + in[i - 1]. */
int tinsel_hip_selftest_sort(int device_index, unsigned long long* keys, unsigned long long n, int begin_bit, int end_bit);
int tinsel_hip_selftest_scan(int device_index, const int* in, int* out, unsigned long long n);

/* Yard-sticks measured on the GPU itself, for bench.py's roofline (not part of the render path): kind 0 = a float4 stream
 * copy of `bytes` bytes (*out_units = bytes read + written); kinds 1..3 = dependent chases through a table of 64-B records
 * of `bytes` bytes (rounded down to a power of two), `steps` visits per lane, 16 waves per CU (*out_units = records
 * visited) -- the access pattern of a BVH walk; the kind only names the kernel for the profiler: 1 a table beyond the
 * Infinity Cache (calibrates FETCH_SIZE for random 64-B gathers), 2 one the size of a walked tree, 3 one inside an L2.
 * *out_ms = the timed launch (HIP events). */
int tinsel_hip_ubench(int device_index, int kind, unsigned long long bytes, int steps, double* out_ms, double* out_units);

/* ------------------------------------------------------------------------- */
/* Scene packs: a relocatable single-blob serialisation of tinsel_scene_desc   */
/* (+ the scene's camera/options) so that scenes travel to machines without    */
/* the reference's loader.  See DESIGN.md "Scene pack".                         */

#define TINSEL_PACK_MAGIC "TINPACK1"

typedef struct tinsel_pack_header {
    char magic[8];
    uint32_t version;               /* 1 */
    uint32_t num_primitives;
    uint32_t num_bvh_nodes;
    uint32_t num_meshes;
    uint64_t total_bytes;
    uint64_t off_primitives;        /* tinsel_primitive[]; mesh pointer fields hold BYTE OFFSETS into the blob */
    uint64_t off_bvh_nodes;
    uint64_t off_probe_data;        /* 0 when there is no probe */
    uint64_t off_probe_pdf_x;
    uint64_t off_probe_cdf_x;
    uint64_t off_probe_pdf_y;
    uint64_t off_probe_cdf_y;
    int32_t probe_width;
    int32_t probe_height;
    tinsel_vec3 sky_horizon;
    tinsel_vec3 sky_zenith;
    tinsel_camera camera;
    tinsel_options options;
    uint8_t _reserved[48];
} tinsel_pack_header;

/* Resolves the offsets of a pack blob (in place: `blob` must stay alive and
 * writable) into a tinsel_scene_desc whose pointers point into the blob. */
int tinsel_pack_open(void* blob, size_t size, tinsel_scene_desc* out_scene,
                     tinsel_camera* out_camera, tinsel_options* out_options);

#ifdef __cplusplus
}

static_assert(sizeof(tinsel_vec3) == 12, "Vec3");
static_assert(sizeof(tinsel_transform) == 32, "Transform");
static_assert(sizeof(tinsel_bvh_node) == 32, "BVHNode");
static_assert(sizeof(tinsel_camera) == 40, "Camera");
static_assert(sizeof(tinsel_texture) == 24, "Texture");
static_assert(sizeof(tinsel_material) == 128, "Material");
static_assert(offsetof(tinsel_material, eta) == 36, "Material.eta");
static_assert(offsetof(tinsel_material, transmission) == 80, "Material.transmission");
static_assert(offsetof(tinsel_material, bump_map) == 88, "Material.bumpMap");
static_assert(offsetof(tinsel_material, bump) == 112, "Material.bump");
static_assert(sizeof(tinsel_mesh_geometry) == 64, "MeshGeometry");
static_assert(offsetof(tinsel_mesh_geometry, num_vertices) == 40, "MeshGeometry.numVertices");
static_assert(offsetof(tinsel_mesh_geometry, area) == 52, "MeshGeometry.area");
static_assert(offsetof(tinsel_mesh_geometry, id) == 56, "MeshGeometry.id");
static_assert(sizeof(tinsel_primitive) == 272, "Primitive");
static_assert(offsetof(tinsel_primitive, type) == 64, "Primitive.type");
static_assert(offsetof(tinsel_primitive, geo) == 72, "Primitive.geo");
static_assert(offsetof(tinsel_primitive, material) == 136, "Primitive.material");
static_assert(offsetof(tinsel_primitive, light_samples) == 264, "Primitive.lightSamples");
static_assert(sizeof(tinsel_filter) == 16, "Filter");
static_assert(sizeof(tinsel_options) == 48, "Options");
static_assert(offsetof(tinsel_options, max_depth) == 40, "Options.maxDepth");
static_assert(sizeof(tinsel_pack_header) == 256, "pack header");
static_assert(sizeof(tinsel_hip_tuning) == 120, "tuning");
#endif

#endif /* TINSEL_HIP_H */
