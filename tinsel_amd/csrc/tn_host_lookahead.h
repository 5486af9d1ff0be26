// tn_host_lookahead.h -- look-ahead for the reference's one-pass-per-call pattern
// (part of the library's one host translation unit: included by tinsel_hip.hip, in this order, never on its own)
#pragma once

namespace {

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
