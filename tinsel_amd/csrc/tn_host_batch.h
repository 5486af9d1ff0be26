// tn_host_batch.h -- one batch of passes: timers, kernel launches, grids and regions, render_batch / render_impl
// (part of the library's one host translation unit: included by tinsel_hip.hip, in this order, never on its own)
#pragma once

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
constexpr int kGridMultDefault = 32;
int grid_mult(const tinsel_hip* r) { return r->tune.grid_mult > 0 ? r->tune.grid_mult : kGridMultDefault; }

// the paired pipeline (tn_paired.h) takes flat-scan scenes without detail counting; anything else asked of it runs the split pipeline
bool paired_can(const tinsel_hip* r) { return r->scene.flatScan != 0 && !r->countDetail; }

// Which scan variants a scene gets (LaunchArgs::walkedOnly): 1 = every mesh primitive is walked by k_walk -- the lean kernels, compiled without
// the inline mesh walk; 2 = every mesh primitive is walked OR is a quad (one internal node over two triangles: a lamp, a card), which the scan
// tests with ray_mesh_two_leaves -- no stack, no loop -- where the arena is staged (glass.tin: sphere and cube walked, the lamp a quad);
// 0 = some mesh is walked inline: the general kernels.
int walked_only_level(const tinsel_hip* r)
{
    if (!r->walkEnabled || r->walkPrims.count == 0 || r->countDetail || r->scene.allInArena)
        return 0;
    int meshPrims = 0, quads = 0;
    for (size_t i = 0; i < r->primMesh.size(); ++i)
    {
        const int m = r->primMesh[i];
        if (m < 0)
            continue;
        ++meshPrims;
        if (!(r->primsHost[i].flags & kPrimWalked) && r->meshesNow[(size_t)m].twoLeaves)
            ++quads;
    }
    if (meshPrims == r->walkPrims.count)
        return 1;
    const bool mixedArena = r->scene.arenaLdsBytes != 0 && r->scene.arenaLdsBytes == r->scene.arenaBytes;
    if (r->tune.quads_in_scan != 0 && mixedArena && meshPrims == r->walkPrims.count + quads)
        return 2;
    return 0;
}

// AUTO's choice between the split pipeline and its paired re-cut, by what was measured (profiles/r06_h_ab_paired.md, Msamples/s split -> paired):
// k_step runs a path's shadow resolve, its closest hit and its shading in ONE kernel at four waves per SIMD, so it wins where all three are lean --
// every mesh walked by k_walk (the scan kernels' WONLY variants: no inline mesh walk, no deep stack), nothing moving (no pose interpolated per ray),
// lights that are sampled without a search (spheres, a probe, small meshes): the 524k-triangle config 2232 -> 2536, the Aphrodite scan 2865 -> 3290,
// transmission.tin 1211 -> 1338 -- and is a wash or loses elsewhere: glass 1613 -> 1608 and table.tin 1323 -> 1307 (a mesh walked inline),
// motionblur 1488 -> 1410 (a moving mesh), meshlight.tin 1471 -> 1380 (a 36,752-triangle light: a CDF search per sample).
bool paired_preferred(const tinsel_hip* r)
{
    // (level 2 -- a quad tested in the scan beside the walked meshes, glass.tin -- stays with the split pipeline: 1662 against 1601 Msamples/s,
    // profiles/r06_2d_ab_glass.md; its k_step is not short of registers any more, it has 2x the instructions per path-step of the 524k-triangle config's)
    if (!paired_can(r) || r->walkPrims.count == 0 || !r->walkEnabled || walked_only_level(r) != 1)
        return false;
    for (const Prim64& p : r->primsHost)
        if (p.flags & kPrimMoving)
            return false;
    for (int32_t i : r->lightPrims)
        if (r->primMesh[(size_t)i] >= 0 && r->meshesNow[(size_t)r->primMesh[(size_t)i]].numTris > 64)
            return false;
    return true;
}

int resolve_pipeline(const tinsel_hip* r)
{
    if (r->pipeline == TINSEL_PIPELINE_WAVEFRONT_PAIRED)
        return paired_can(r) ? TINSEL_PIPELINE_WAVEFRONT_PAIRED : TINSEL_PIPELINE_WAVEFRONT_SPLIT;
    if (r->pipeline == TINSEL_PIPELINE_AUTO && !r->scene.allInArena && paired_preferred(r))
        return TINSEL_PIPELINE_WAVEFRONT_PAIRED;
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
int alloc_dense(tinsel_hip* r, size_t slots, int maxDepth, bool split, bool paired = false)
{
    const size_t K = (split || paired) ? (size_t)r->neePerPath : 0;
    // (half as many again as the widest grid: the short regions at the end of a batch, split_tail_regions)
    const size_t maxRegions = (size_t)r->numCUs*(size_t)grid_mult(r)*(kBlock/kWave)*3/2;
    const size_t cap = (slots + maxRegions*kWave + kWave - 1)/kWave*kWave;       // a region is a whole number of waves long; whole 64-position blocks
    SplitState& ss = r->ss;
    memset(&ss, 0, sizeof(ss));
#if TN_STATE_BLOCKS
    if (cap*kStateFields >= 0xffffffffull)      // (sidx() is 32-bit arithmetic; 858 M positions are 250 GB of path state anyway)
        return fail("render: batch too large for the path-state blocks");
#endif
    for (int b = 0; b < 2; ++b)
    {
#if TN_STATE_BLOCKS
        // ONE allocation per buffer: blocks of 64 positions x 5 fields (tn_layout.h); a field's pointer is its first block entry
        float4* base = nullptr;
        if (batch_alloc(r, &base, cap*kStateFields))
            return -1;
        ss.rayO[b] = base; ss.rayD[b] = base + 64; ss.thr[b] = base + 128; ss.rad[b] = base + 192; ss.rngId[b] = base + 256;
#else
        if (batch_alloc(r, &ss.rayO[b], cap) || batch_alloc(r, &ss.rayD[b], cap) || batch_alloc(r, &ss.thr[b], cap) ||
            batch_alloc(r, &ss.rad[b], cap) || batch_alloc(r, &ss.rngId[b], cap))
            return -1;
#endif
    }
    if (batch_alloc(r, &ss.segFront, maxRegions*((size_t)maxDepth + 2)) || batch_alloc(r, &ss.segBack, maxRegions*((size_t)maxDepth + 2)) ||
        batch_alloc(r, &r->regionOrder, maxRegions/(kBlock/kWave)) || batch_alloc(r, &r->regionOrderNee, maxRegions/(kBlock/kWave)))
        return -1;
    ss.radOut = r->ps.rad;
    ss.capacity = (uint32_t)cap;
    r->splitCap = cap;
    r->splitMaxRegions = (uint32_t)maxRegions;
    r->walkRec = nullptr;
    r->walkList = nullptr;
    r->segPrefix = nullptr;
    if (paired)
    {
        // the K pending light samples of every path, by position, double-buffered like the state (tn_paired.h); k_walk's records: K + 1 rays per position
        ss.neePerPath = (int32_t)K;
        for (int b = 0; b < 2; ++b)
            if (batch_alloc(r, &ss.pairThr[b], K ? cap : 1) || batch_alloc(r, &ss.pairRay[b], cap*K*2) || batch_alloc(r, &ss.pairPend[b], K > 1 ? cap*K*2 : cap))     // (sample k's two records at (2k, 2k + 1)*cap; sample 0 has no second one)
                return -1;
        if (r->walkPrims.count > 0 && r->walkEnabled && (double)cap*(double)(K + 1)*r->walkPrims.count < 2147483648.0)
            if (batch_alloc(r, &r->walkRec, cap*(K + 1)*(size_t)r->walkPrims.count*2) || batch_alloc(r, &r->walkList, cap) ||
                batch_alloc(r, &r->segPrefix, maxRegions + 1))
                return -1;
        return 0;
    }
    if (!split)
        return 0;

#if TN_STATE_BLOCKS
    {
        // the hand-over records by position, one block of 64: hit (1 KB) | hit primitive (256 B) | NEE position (256 B)
        float4* hbase = nullptr;
        if (batch_alloc(r, &hbase, cap/kWave*kHitBlockF4))
            return -1;
        ss.hit = hbase;
        ss.hitPrim = reinterpret_cast<int32_t*>(hbase) + 256;
        ss.pathNee = reinterpret_cast<uint32_t*>(hbase) + 320;
    }
    if (
#else
    if (batch_alloc(r, &ss.hit, cap) || batch_alloc(r, &ss.hitPrim, cap) || batch_alloc(r, &ss.pathNee, K ? cap : 1) ||
#endif
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
            if (alloc_dense(r, stateSlots, maxDepth, pipeline == TINSEL_PIPELINE_WAVEFRONT_SPLIT, pipeline == TINSEL_PIPELINE_WAVEFRONT_PAIRED))
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
// tinsel_hip_create, which falls back to planning within 64 KB per workgroup when the runtime refuses a kernel (r->prepRefused: "kernel name (arm)").
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
// (mixed: the paired pipeline's walk -- K shadow rays + the extension ray of every queued position, tn_paired.h)
int launch_walk(tinsel_hip* r, hipStream_t st, LaunchArgs a, const uint32_t* regionCounts, bool shadowRays, bool mixed = false)
{
    // (measured and settled, profiles/EXPERIMENTS.md: one resident set of workgroups; a refill once 24 lanes idle; a triangle phase once 8 wait)
    const int gridMult = r->tune.walk_grid_mult > 0 ? r->tune.walk_grid_mult : 1, refillMin = r->tune.walk_refill_min > 0 ? r->tune.walk_refill_min : 24, leafMin = r->tune.walk_leaf_min > 0 ? r->tune.walk_leaf_min : 8;
    const int forceBlock = r->tune.walk_block;
    prepare_kernels_once(r);
    // the work list: the front entries of every region (paths / shadow-ray bundles whose ray enters a walked mesh's box)
    const SplitState& ss = a.ss;
    // the list visits the regions a golden-section step apart
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
            return fail("k_seg_prefix: " + std::to_string(ss.numRegions) + " regions do not fit its LDS (tinsel_hip_tuning::grid_mult too large for this device)");
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
    job.mixed = 0;
    if (r->lastPipeline == TINSEL_PIPELINE_WAVEFRONT_PAIRED)
    {
        job.nee = ss.pairRay[a.bounce & 1];
        job.mixed = mixed ? 1 : 0;
        if (shadowRays && !mixed)
            job.mixed = 2;          // shadow rays only, their time in the state (tn_walk.h reads rayO[slot].w whenever `mixed` is set)
    }
    job.numPrims = r->walkPrims.count;
    int entries = 1;
    for (int k = 0; k < kWalkMaxPrims; ++k)
    {
        job.prim[k] = k < r->walkPrims.count ? r->walkPrims.prim[k] : 0;
        job.topCount[k] = 0;
        if (k < r->walkPrims.count)
            entries = std::max(entries, r->meshesNow[(size_t)r->walkPrimMesh[k]].stackNeed);
    }
    job.stackEntries = entries;
    job.prof = r->walkProf;
    job.refillMin = std::min(64, std::max(1, refillMin));
    job.leafMin = std::min(64, std::max(1, leafMin));
    // ONE walked primitive: its tree as kernel-argument scalars (tinsel_hip_tuning::walk_single = 0: per-lane pointers as for several; tests)
    a.walkSingle = (r->walkPrims.count == 1 && r->tune.walk_single != 0) ? 1 : 0;

    // (k_walk_rays keeps one LDS entry per walked primitive behind its control words: tn_walk.h kWalkPrimWords)
    const size_t ctl = (kWalkCtlWords + (a.walkSingle ? 0 : kWalkMaxPrims*kWalkPrimWords))*sizeof(uint32_t);
    // n stack entries per lane in LDS (tinsel_hip_tuning::walk_lds_stack, default 8; 0: the deepest tree's need, one workgroup per CU), the
    // rest of the deepest tree's need in HBM, and TWO 1024-thread workgroups per CU (8 waves per SIMD at 64 VGPRs) sharing the CU's
    // LDS: the 524k-triangle config's k_walk 19.0 -> 16.6 ms per 32 passes (2042 -> 2199 Msamples/s; 6 entries 17.2, 12 entries 16.7),
    // glass 10.4 -> 9.9; results unchanged (a stack entry is a stack entry wherever it lives)
    const int ldsStackEnv = r->tune.walk_lds_stack >= 0 ? r->tune.walk_lds_stack : 8;
    const bool twoPerCU = ldsStackEnv > 0 && !forceBlock;
    const int ldsEntries = twoPerCU ? std::min(entries, std::max(1, ldsStackEnv)) : entries;
    job.stackEntries = ldsEntries;
    job.overflow = nullptr;
    job.overflowEntries = 0;
    const size_t stackBig = (size_t)(ldsEntries + (a.walkSingle ? kWalkLaneRows : kWalkRayRows))*1024*sizeof(uint32_t);     // (+ the per-lane rows, tn_walk.h)
    const size_t ldsBudget = twoPerCU ? (size_t)r->sharedMemLimit/2 : (size_t)r->sharedMemLimit;
    const bool big = forceBlock ? forceBlock == 1024 : stackBig + ctl + 16384 <= ldsBudget;
    const int block = big ? 1024 : 256;
    size_t lds = (size_t)(ldsEntries + (a.walkSingle ? kWalkLaneRows : kWalkRayRows))*block*sizeof(uint32_t) + ctl;
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
    }
    // (a work item is a ray; with several walked primitives a lane keeps its ray as position | k << 27 while primitives are left: tn_walk.h)
    // (k_walk_rays, whatever the number of walked primitives: ADVICE r05)
    if (!a.walkSingle && (ss.capacity >= (1u << 27) || r->neePerPath >= 32))
        return fail("k_walk_rays: batch too large (>= 2^27 positions) or too many shadow rays per path (>= 32): lower tinsel_hip_tuning::batch_paths");
    const size_t items = r->lastBatchSlots*(size_t)((shadowRays && r->neePerPath > 1 ? r->neePerPath : 1) + (mixed ? 1 : 0));
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
    const bool noLds = r->tune.swalk_lds == 0;
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
    std::vector<std::pair<long long, int>> weighted;
    for (int ty = 0; ty < tilesY; ++ty)
    {
        for (int tx = 0; tx < tilesX; ++tx)
        {
            // candidate paths of this tile are generated at pixels [x0, x1] x [y0, y1]
            const int x0 = std::max(0, tx*kAccTile - reachLo), x1 = std::min(fp.width - 1, tx*kAccTile + kAccTile - 1 + reachHi);
            const int y0 = std::max(0, ty*kAccTile - reachLo), y1 = std::min(fp.height - 1, ty*kAccTile + kAccTile - 1 + reachHi);
            // ... of which this shard's: the rectangle cut with every shard tile it touches
            long long owned = 0;
            for (int sy = y0/fp.shardTile; sy <= y1/fp.shardTile; ++sy)
                for (int sx = x0/fp.shardTile; sx <= x1/fp.shardTile; ++sx)
                    if (((sy*shardX + sx) % fp.shardWorld) == fp.shardRank)
                        owned += (long long)(std::min(x1, sx*fp.shardTile + fp.shardTile - 1) - std::max(x0, sx*fp.shardTile) + 1)*
                                 (std::min(y1, sy*fp.shardTile + fp.shardTile - 1) - std::max(y0, sy*fp.shardTile) + 1);
            if (owned > 0)
                weighted.push_back({ -owned, ty*tilesX + tx });
        }
    }
    // the tiles with the most candidates first: a halo tile (a strip of a neighbouring shard tile's pixels) is a fraction of an inner
    // tile's work per pass (k_accumulate_tiled), and the launch ends in whatever was handed out last
    std::stable_sort(weighted.begin(), weighted.end(), [](const std::pair<long long, int>& a, const std::pair<long long, int>& b) { return a.first < b.first; });
    std::vector<int> list;
    for (const auto& w : weighted)
        list.push_back(w.second);
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
            // Which kernel (the same adds in the same order, tests/test_gpu_switches.py):
            //   a block per CU or less: staging of pass s + 1 overlapped with the gather of pass s (k_accumulate_piped: the launch is as long as
            //     one tile's pass loop; cornell 256^2 x 16 passes 0.053 -> 0.037 ms, profiles/r05_q_ab_acc_piped.md; at 1024 tiles 0.100 -> 0.107,
            //     one shard of 8 of cornell 1024^2 0.565 -> 0.525 at 64-pixel tiles but 0.563 -> 0.613 at 32: not used there);
            //   up to a wave per SIMD: 512-thread workgroups, the second half only stages (profiles/r03_y_ab_acc_wide.md);
            //   else 256-thread workgroups.
            bool piped = tiles <= r->numCUs;
            bool wide = tiles <= r->numCUs*4;
            if (r->tune.accumulate != TINSEL_ACCUMULATE_AUTO)
            {
                piped = r->tune.accumulate == TINSEL_ACCUMULATE_PIPED;
                wide = r->tune.accumulate == TINSEL_ACCUMULATE_WIDE;
            }
            if (span == 3 && piped)
                hipLaunchKernelGGL((k_accumulate_piped<3>), dim3(tiles), dim3(kAccPipeThreads), 0, st, r->ps, fp, target, r->passSeeds, tileList);
            else if (span == 4 && piped)
                hipLaunchKernelGGL((k_accumulate_piped<4>), dim3(tiles), dim3(kAccPipeThreads), 0, st, r->ps, fp, target, r->passSeeds, tileList);
            else if (span == 3 && wide)
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
    // (at least as many workgroups as the chip holds at once -- k_bounce's resident workgroups per CU, three for the split pipeline's kernels --
    // where the batch has that many 256-path pieces: a 1 M-path batch would otherwise leave the last wave slot empty)
    const int gridMin = pipeline == TINSEL_PIPELINE_WAVEFRONT ? r->bounceWaves : 3;
    const size_t lo = std::min<size_t>((size_t)r->numCUs*(size_t)gridMin, (slots + kBlock - 1)/kBlock), hi = (size_t)r->numCUs*(size_t)grid_mult(r);
    size_t grid = std::max<size_t>(1, std::min(hi, std::max(lo, blocks)));
    // The workgroups that HAVE work (regions are a whole number of waves long, so fewer than the grid may) as close to a whole number
    // of resident sets (gridMin per CU) as the region length allows within +-25 %: the last set of a launch is then full instead
    // of, say, two thirds empty.  Glass at 20 passes per batch 1262 -> 1291 Msamples/s, the 524k-triangle config 2024 -> 2034, the
    // fused configs +-0 (profiles/r03_s_ab_grid_round.txt)
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
    if (W < 2)
        return false;
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
// 2414 -> 2610 (profiles/r03_z5_ab_tail_split.md).  tinsel_hip_tuning::tail_split / tail_share / tail_divide (A/B).  On return *grid is the number
// of region groups = the workgroups of a launch that gives every group its own.
int cut_regions(tinsel_hip* r, LaunchArgs& a, size_t slots, int* grid, size_t maxRegions)
{
    if (set_regions(r, a, slots, *grid))
        return -1;
    double share = -0.5;
    int divide = 4;
    if (r->tune.tail_split == 0)
        share = 0.0;
    else if (r->tune.tail_split > 0)
    {
        share = (double)r->tune.tail_share;
        divide = r->tune.tail_divide;
    }
    // (r->bounceWaves workgroups per CU are resident: plan_bounce)
    if (share < 0.0 && maxRegions > 0 && (size_t)*grid <= (size_t)r->numCUs*r->bounceWaves && (size_t)*grid > (size_t)r->numCUs*(r->bounceWaves - 1) && split_one_set(r, a, slots))
    {
        *grid = (int)(a.ss.numRegions/(kBlock/kWave));
        return 0;
    }
    if (share < 0.0)
    {
        // a negative share: that multiple of ONE resident set's part of the batch (r->bounceWaves workgroups per CU).  The default, half a
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

// k_bounce's LDS plan for the scene: does it close ranks through the waves' shading pools (returned)?  Pools (25 KB of LDS per workgroup)
// where rays can LEAVE the scene -- veach 1515 -> 1866 Msamples/s, features 755 -> 865, env_loft 3598 -> 3793, gloss 7584 -> 7934 when they
// were introduced; between two facing planes every ray hits something and the pools only cost (cornell 2919 -> 2894) -- and where they do not
// cost the third resident workgroup (features' 32-KB arena + pools would leave two).  The kernel runs four waves per SIMD = four workgroups
// per CU by its registers (kBounceWaves); r->bounceWaves, which the grid and the region cut are sized by, is what the LDS lets be resident.
bool plan_bounce(tinsel_hip* r)
{
    const size_t perCU = 160u*1024u;
    const size_t lds = stack_bytes(r), withPool = lds + kPoolWords*sizeof(uint32_t);
    const bool want = r->tune.repack >= 0 ? r->tune.repack != 0 : !r->sceneEnclosed;      // tinsel_hip_tuning::repack 0 / 1: never / always; default: open scenes
    const bool pools = want && withPool*3 <= perCU && withPool <= (size_t)r->sharedMemLimit;
    // the workgroups per CU the grid and the region cut are sized by = the ones that ARE resident: four by the registers, fewer where a
    // workgroup's LDS (stacks + staged arena + pools) says so (with pools, three fit where four times their LDS do not)
    const size_t ldsPerGroup = std::max<size_t>(1, pools ? withPool : lds);
    r->bounceWaves = (int)std::max<size_t>(1, std::min<size_t>((size_t)kBounceWaves, perCU/ldsPerGroup));
    return pools;
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
        // bounces > 0: a workgroup's four regions as ONE stream dealt to its waves -- where a round is long (three or more shadow rays)
        // and where the regions are short (a small batch: the ragged last round of every region and bounce weighs more)
        {
            const int shareLen = 512;       // cornell 256^2 x 16 passes (regions of 384): 2258 -> 2311 Msamples/s; 1024^2 x 20 (regions of 2048) +0.3 %
            a.fp.share = r->tune.bounce_share >= 0 ? (r->tune.bounce_share != 0) : (r->neePerPath >= 3 || (int)a.ss.regionLen <= shareLen);
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
    else if (pipeline == TINSEL_PIPELINE_WAVEFRONT_PAIRED)
    {
        // ONE k_walk and ONE streaming kernel per bounce (tn_paired.h): walk { shadow rays of bounce b - 1, extension rays of bounce b }, then k_step(b)
        const bool walk = walk_records(r) != nullptr;
        const int walkedLevel = walk ? walked_only_level(r) : 0;
        const bool walkedOnly = walkedLevel != 0;
        const int stackScan = walkedOnly ? std::max(1, pick_stack(r->sceneStackNeed)) : r->stackNeed;
        const uint32_t ldsTrace = walkedOnly ? (uint32_t)(((size_t)stackScan*kBlock + kScanWords)*sizeof(uint32_t) + r->scene.arenaLdsBytes) : a.ldsBytes;
        a.walkedOnly = walkedLevel;
        if (r->walkList != nullptr)
            gridPersist = std::max(1, std::min(gridPersist, (int)(seg_prefix_max_regions(r)/(kBlock/kWave))));
        if (cut_regions(r, a, slots, &gridPersist, (size_t)0))
            return -1;
        const size_t W = a.ss.numRegions;
        {
            ScopedTimer t(r, KN_GENERATE, st);
            a.grid = gridPersist;
            launch_path(r, PK_GENERATE, a, st);
        }
        const int K = r->neePerPath;
        // (the step after the last bounce only resolves the last bounce's light samples: nothing to do without lights)
        const int steps = fp.maxDepth + (K > 0 ? 1 : 0);
        for (int bounce = 0; bounce < steps; ++bounce)
        {
            a.bounce = bounce;
            a.order = nullptr;
            if (walk)
            {
                a.grid = gridPersist;
                const bool first = bounce == 0, last = bounce >= fp.maxDepth;
                if (launch_walk(r, st, a, r->ss.segFront + (size_t)bounce*W, /*shadowRays*/ !first && K > 0, /*mixed*/ !first && !last && K > 0))
                    return -1;
            }
            // k_step's region groups longest first (k_region_order) where paths are deep: by bounce 6 most regions are nearly empty and the few
            // full ones should not start last -- transmission.tin (depth 16) k_step 13.45 -> 11.81 ms, 1329 -> 1396 Msamples/s; at depth 4 the sort's
            // launch costs what it saves (the 524k-triangle config 2562 / 2555: profiles/r06_r_ab_step_order.md)
            if (bounce > 0 && fp.maxDepth >= 6 && gridPersist > r->numCUs*2)
            {
                ScopedTimer t(r, KN_SEG, st);
                hipLaunchKernelGGL(k_region_order, dim3(1), dim3(kOrderBlock), 0, st, (const uint32_t*)(r->ss.segFront + (size_t)bounce*W), (const uint32_t*)(r->ss.segBack + (size_t)bounce*W),
                                   a.ss.numRegions, r->regionOrder);
                a.order = r->regionOrder;
            }
            ScopedTimer t(r, KN_STEP, st);
            a.grid = gridPersist;
            a.ldsBytes = ldsTrace;
            a.stackEntries = stackScan;
            launch_path(r, PK_STEP, a, st);
        }
    }
    else
    {
        const bool walk = walk_records(r) != nullptr;
        // every mesh primitive walked by k_walk: the scan kernels run their lean variants with the scene-level stack only
        const int walkedLevel = walk ? walked_only_level(r) : 0;
        const bool walkedOnly = walkedLevel != 0;
        const int stackScan = walkedOnly ? std::max(1, pick_stack(r->sceneStackNeed)) : r->stackNeed;
        const uint32_t ldsTrace = walkedOnly ? (uint32_t)(((size_t)stackScan*kBlock + kScanWords)*sizeof(uint32_t) + r->scene.arenaLdsBytes) : a.ldsBytes;
        // k_shade has no traversal stacks in LDS and reads a material per path: an arena too large to sit beside the stacks of the
        // trace kernels (32 KB) is still staged by it up to 60 KB (many_spheres, 39 KB of primitive and material records: k_shade
        // 7.8 -> 6.7 ms; staged in the trace kernels too it costs them their fourth wave per SIMD, 1380 -> 1280 Msamples/s, and
        // k_lights reads too little of it to repay the copy, 2.7 -> 3.1 ms)
        const uint32_t arenaLdsTrace = r->scene.arenaLdsBytes;
        const uint32_t arenaLdsShade = (arenaLdsTrace == 0 && r->scene.arenaBytes <= 61440u && r->tune.lds_scene != 0) ? r->scene.arenaBytes : arenaLdsTrace;
        const uint32_t ldsShade = r->scene.allInArena ? r->scene.arenaBytes : arenaLdsShade;
        a.walkedOnly = walkedLevel;
        const bool noSceneWalkEarly = r->tune.scene_walk == 0;
        // the lean k_extend draws the light samples itself (tn_launch.h launches it when walkedOnly and not counting)
        // ... and so does the variant for a staged arena with meshes in HBM (glass): without the SLP vectoriser it fits 128 VGPRs and
        // saves k_lights' pass over the path state (k_extend 5.7 + k_lights 7.1 -> 11.2 ms per 32 passes, glass 1366 -> 1425 Msamples/s;
        // round 2, 170 VGPRs: 29.1 apart, 31.5 together; profiles/r03_k_ab_lights_in_extend.txt)
        const bool mixedArena = !r->scene.allInArena && r->scene.arenaLdsBytes != 0 && r->scene.arenaLdsBytes == r->scene.arenaBytes;
        const bool lightsInMixed = !walkedOnly && !r->countDetail && mixedArena && !(!noSceneWalkEarly && !r->scene.flatScan);
        a.lightsInExtend = lightsInMixed ? 1 : 0;
        const bool lightsInExtend = (walkedOnly && !r->countDetail) || lightsInMixed;
        // No short regions at the end here: the launches are many and short, k_walk cuts its
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
        const bool noSceneWalk = r->tune.scene_walk == 0;
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
                // tinsel_hip_tuning::shade_sorted = 0 / 1 forces either arm (A/B, tests).
                const bool shadeSorted = (r->tune.shade_sorted >= 0 ? r->tune.shade_sorted != 0 : (!r->sceneEnclosed && !walk)) &&
                                         (size_t)ldsShade + kShadeListWords*sizeof(uint32_t) <= (size_t)r->sharedMemLimit;
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
    // or more.  tinsel_hip_tuning::overlap = 0 / 1: never / wherever a batch has two passes (A/B, tests).
    int chunkPasses = perBatch;
    int lanes = 1;
    {
        const size_t minPaths = (size_t)8u << 20;
        const int pipeline = resolve_pipeline(r);
        const bool can = !traceOnly && perBatch >= 2 && pipeline != TINSEL_PIPELINE_MEGAKERNEL;
        const bool noSceneWalk = r->tune.scene_walk == 0;
        const bool sceneWalked = pipeline == TINSEL_PIPELINE_WAVEFRONT_SPLIT && !r->scene.flatScan && !noSceneWalk && !r->countDetail &&
                                 !(r->walkPrims.count > 0 && r->walkEnabled);
        const bool want = r->tune.overlap >= 0 ? r->tune.overlap != 0 : (sceneWalked && perPass*(size_t)perBatch >= minPaths);
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


} // namespace
