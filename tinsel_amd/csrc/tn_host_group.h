// tn_host_group.h -- tinsel_hip_group: N devices of one node behind one Renderer
// (part of the library's one host translation unit: included by tinsel_hip.hip, in this order, never on its own)
#pragma once

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
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
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
        GetUniqueId = (decltype(GetUniqueId))dlsym(lib, "ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))dlsym(lib, "ncclCommInitRank");
        CommCount = (decltype(CommCount))dlsym(lib, "ncclCommCount");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        Reduce = (decltype(Reduce))dlsym(lib, "ncclReduce");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!CommInitAll || !GetUniqueId || !CommInitRank || !CommCount || !CommDestroy || !Reduce || !GetErrorString)
        {
            error = "RCCL library lacks ncclCommInitAll / ncclCommInitRank / ncclReduce";
            lib = nullptr;
            return false;
        }
        return true;
    }
};

RcclApi g_rccl;

// THE collective of the path -- the only place the library calls ncclReduce: sum of every rank's W*H float4 accumulator (or look-ahead snapshot)
// into `dst` on `root` (dst is read on the root only), on the caller's stream, then a wait for that stream.  Its callers: a group's worker
// threads (one communicator per member, ncclCommInitAll) and the process-per-GPU arm below (tinsel_hip_comm_*: ncclCommInitRank).
void comm_release(tinsel_hip* r)
{
    if (r && r->comm && g_rccl.CommDestroy)
        (void)g_rccl.CommDestroy((ncclComm_t)r->comm);
    if (r)
    {
        r->comm = nullptr;
        r->commWorld = 0;
    }
}

int rccl_reduce_accum(ncclComm_t comm, const float4* src, float4* dst, size_t pixels, int root, hipStream_t st, const char* who)
{
    const ncclResult_t e = g_rccl.Reduce(src, dst, pixels*4, ncclFloat, ncclSum, root, comm, st);
    if (e != ncclSuccess)
        return fail(std::string(who) + ": ncclReduce: " + g_rccl.GetErrorString(e));
    if (hipStreamSynchronize(st) != hipSuccess)
        return fail(std::string(who) + ": the reduce failed on the device");
    return 0;
}

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
            rc = rccl_reduce_accum(m.comm, m.r->accum, g->total, (size_t)g->width*g->height, 0, m.stream, "group");
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
                    rc = rccl_reduce_accum(m.comm, shot.buf, g->totalNext, (size_t)g->width*g->height, 0, m.stream, "group look-ahead");
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

tinsel_hip_group* tinsel_hip_group_create(const tinsel_scene_desc* scene, int num_gpus, int tile) { return tinsel_hip_group_create_tuned(scene, num_gpus, tile, nullptr); }

tinsel_hip_group* tinsel_hip_group_create_tuned(const tinsel_scene_desc* scene, int num_gpus, int tile, const tinsel_hip_tuning* tuning)
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
        m.r = tinsel_hip_create_tuned(scene, m.device, tuning);
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


// ---------------------------------------------------------------------------
// One process per GPU (bench.py --gpus N under torch.distributed.run; any MPI-style host): the same collective, the communicator made
// with ncclCommInitRank from a unique id that the HOST LANGUAGE carries from rank 0 to the others (a torch.distributed broadcast, an
// MPI_Bcast, a file) -- the library opens no sockets of its own beyond RCCL's.

int tinsel_hip_comm_unique_id(unsigned char* id_bytes, int capacity)
{
    if (!id_bytes || capacity < (int)sizeof(ncclUniqueId))
        return fail("comm_unique_id: the buffer must hold TINSEL_HIP_COMM_ID_BYTES bytes");
    if (!g_rccl.load())
        return fail("comm_unique_id: " + g_rccl.error);
    ncclUniqueId id;
    const ncclResult_t e = g_rccl.GetUniqueId(&id);
    if (e != ncclSuccess)
        return fail(std::string("comm_unique_id: ncclGetUniqueId: ") + g_rccl.GetErrorString(e));
    memset(id_bytes, 0, (size_t)capacity);
    memcpy(id_bytes, &id, sizeof(id));
    return 0;
}

int tinsel_hip_comm_init(tinsel_hip* r, const unsigned char* id_bytes, int rank, int world)
{
    if (!r || !id_bytes || world < 1 || rank < 0 || rank >= world)
        return fail("comm_init: bad arguments");
    if (!g_rccl.load())
        return fail("comm_init: " + g_rccl.error);
    comm_release(r);
    HIP_TRY(hipSetDevice(r->device));
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    ncclComm_t comm = nullptr;
    const ncclResult_t e = g_rccl.CommInitRank(&comm, world, id, rank);
    if (e != ncclSuccess)
        return fail(std::string("comm_init: ncclCommInitRank: ") + g_rccl.GetErrorString(e));
    r->comm = comm;
    r->commRank = rank;
    r->commWorld = world;
    return 0;
}

int tinsel_hip_comm_size(tinsel_hip* r)
{
    if (!r || !r->comm)
        return 0;
    int n = 0;
    if (g_rccl.CommCount((ncclComm_t)r->comm, &n) != ncclSuccess)
        return -1;
    return n;
}

int tinsel_hip_comm_reduce_accum(tinsel_hip* r, float* out_device, int root, void* stream)
{
    if (!r || !r->comm || !r->accum)
        return fail("comm_reduce_accum: no communicator (tinsel_hip_comm_init) or no accumulator (tinsel_hip_init)");
    if (root < 0 || root >= r->commWorld || (r->commRank == root && !out_device))
        return fail("comm_reduce_accum: bad root / the root needs an output buffer");
    HIP_TRY(hipSetDevice(r->device));
    return rccl_reduce_accum((ncclComm_t)r->comm, r->accum, (float4*)out_device, (size_t)r->width*r->height, root, (hipStream_t)stream, "comm_reduce_accum");
}

} // extern "C"
