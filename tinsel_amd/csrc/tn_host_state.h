// tn_host_state.h -- struct tinsel_hip: everything a renderer owns
// (part of the library's one host translation unit: included by tinsel_hip.hip, in this order, never on its own)
#pragma once

// the defaults of include/tinsel_hip.h's tinsel_hip_tuning: "the library decides" everywhere
inline tinsel_hip_tuning tuning_defaults()
{
    tinsel_hip_tuning t;
    memset(&t, 0, sizeof(t));
    t.struct_bytes = (uint32_t)sizeof(t);
    t.flat_scan = t.lds_scene = t.walk = t.inline_max_tris = t.walk_min_tris = -1;
    t.small_mesh_bytes = t.arena_lds_limit = -1;
    t.batch_paths = 0;
    t.grid_mult = 0;
    t.bounce_share = t.repack = t.tail_split = t.shade_sorted = t.overlap = t.scene_walk = t.swalk_lds = -1;
    t.tail_share = 0.0f;
    t.tail_divide = 4;
    t.accumulate = TINSEL_ACCUMULATE_AUTO;
    t.walk_block = 0;
    t.walk_single = t.walk_lds_stack = t.quads_in_scan = -1;
    t.walk_refill_min = t.walk_leaf_min = t.walk_grid_mult = 0;
    return t;
}

// a caller's struct, possibly shorter than this build's: the fields it has over the defaults
inline tinsel_hip_tuning tuning_from_caller(const tinsel_hip_tuning* in)
{
    tinsel_hip_tuning t = tuning_defaults();
    if (in && in->struct_bytes >= sizeof(uint32_t))
        memcpy(&t, in, std::min<size_t>(in->struct_bytes, sizeof(t)));
    t.struct_bytes = (uint32_t)sizeof(t);
    return t;
}

struct tinsel_hip
{
    int device = 0;
    int numCUs = 256;
    tinsel_hip_tuning tune = tuning_defaults();     // (tinsel_hip_create_tuned / tinsel_hip_set_tuning; nothing is read from the environment)

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
    std::vector<int32_t> lightPrims;    // primitives with lightSamples > 0
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
    int bounceWaves = kBounceWaves;     // k_bounce's waves per SIMD = the workgroups per CU its grids and region cuts are sized by
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
    uint32_t* walkOverflow = nullptr;                   // k_walk's stack entries beyond the LDS ones (tinsel_hip_tuning::walk_lds_stack)
    size_t walkOverflowCap = 0;
    bool walkEnabled = true;                            // tinsel_hip_tuning::walk == 0: walk meshes inline in k_extend / k_shadow (A/B)
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
    bool batchSlotsExplicit = false;     // set by tinsel_hip_tuning::batch_paths / tinsel_hip_set_batch_paths
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

    // process-per-GPU arm of the reduce (tinsel_hip_comm_*, tn_host_group.h): this rank's RCCL communicator
    void* comm = nullptr;               // ncclComm_t
    int commRank = 0, commWorld = 0;

    bool timing = false;
    std::vector<TimedSpan> spans;
    std::vector<hipEvent_t> eventPool;
    double gpuSeconds = 0.0;
};
