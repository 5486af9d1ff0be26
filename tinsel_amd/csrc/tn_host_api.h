// tn_host_api.h -- C-ABI: init / render / present / refit / rebuild / settings / statistics / test hooks / scene packs
// (part of the library's one host translation unit: included by tinsel_hip.hip, in this order, never on its own)
#pragma once

extern "C" {

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
    r->scene.scanMask = 0;
    for (int k = 0; k < P && k < 64; ++k)
        if (boxes[(size_t)k].alwaysHit != 2u)
            r->scene.scanMask |= 1ull << k;
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
    if (!r || pipeline < TINSEL_PIPELINE_WAVEFRONT || pipeline > TINSEL_PIPELINE_WAVEFRONT_PAIRED)
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
    r->tune.batch_paths = (int64_t)max_paths;
    return 0;
}

// The per-render fields of the tuning (include/tinsel_hip.h): the create-time ones are baked into the uploaded scene and stay as created.
int tinsel_hip_set_tuning(tinsel_hip* r, const tinsel_hip_tuning* tuning)
{
    if (!r || !tuning)
        return fail("set_tuning: null argument");
    lookahead_cancel(r);
    tinsel_hip_tuning t = tuning_from_caller(tuning);
    if (t.grid_mult < 0 || t.grid_mult > 256 || (t.walk_block != 0 && t.walk_block != 256 && t.walk_block != 1024) ||
        t.accumulate < TINSEL_ACCUMULATE_AUTO || t.accumulate > TINSEL_ACCUMULATE_PIPED || t.walk_refill_min > 64 || t.walk_leaf_min > 64 || t.walk_grid_mult < 0 || t.walk_grid_mult > 64 ||
        (t.tail_split > 0 && (!(t.tail_share >= 0.0f) || t.tail_divide < 1)) || (t.batch_paths != 0 && t.batch_paths < 1024))
        return fail("set_tuning: a field is out of range");
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipDeviceSynchronize());        // (launches in flight use the path buffers free_batch gives back)
    const tinsel_hip_tuning was = r->tune;
    t.flat_scan = was.flat_scan; t.lds_scene = was.lds_scene; t.walk = was.walk; t.inline_max_tris = was.inline_max_tris;
    t.walk_min_tris = was.walk_min_tris; t.small_mesh_bytes = was.small_mesh_bytes; t.arena_lds_limit = was.arena_lds_limit;
    r->tune = t;
    if (t.batch_paths > 0)
    {
        r->maxBatchSlots = (size_t)t.batch_paths;
        r->batchSlotsExplicit = true;
    }
    else
    {
        r->maxBatchSlots = 8u << 20;
        r->batchSlotsExplicit = false;
    }
    free_batch(r);          // (the region arrays are sized by grid_mult)
    return 0;
}

int tinsel_hip_get_tuning(tinsel_hip* r, tinsel_hip_tuning* out)
{
    if (!r || !out)
        return fail("get_tuning: null argument");
    *out = r->tune;
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
    const size_t maxRegions = (size_t)num_cus*(size_t)grid_mult(r)*(kBlock/kWave)*3/2;
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

// The library's own sort and scan (tn_sort.h: the device BVH builder's) on caller data: keys[0, n) sorted in place by their bits
// [begin_bit, end_bit) (multiples of 8), keys with equal bits keeping their order; out[i] = in[0] + .. + in[i - 1].
int tinsel_hip_selftest_sort(int device_index, unsigned long long* keys, unsigned long long n, int begin_bit, int end_bit)
{
    if (!keys || n == 0 || n >= (1ull << 31) || begin_bit < 0 || end_bit > 64 || begin_bit >= end_bit || (begin_bit & 7) || (end_bit & 7))
        return fail("selftest_sort: bad arguments");
    HIP_TRY(hipSetDevice(device_index));
    unsigned long long *a = nullptr, *b = nullptr;
    int* scratch = nullptr;
    int rc = 0;
    if (hipMalloc((void**)&a, n*8) != hipSuccess || hipMalloc((void**)&b, n*8) != hipSuccess ||
        hipMalloc((void**)&scratch, sort_scratch_ints((size_t)n)*sizeof(int)) != hipSuccess)
        rc = fail("selftest_sort: allocation failed");
    else if (hipMemcpy(a, keys, n*8, hipMemcpyHostToDevice) != hipSuccess)
        rc = fail("selftest_sort: upload failed");
    else
    {
        radix_sort_keys(a, b, (size_t)n, begin_bit, end_bit, scratch, nullptr);
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess || hipMemcpy(keys, b, n*8, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail("selftest_sort: kernels failed");
    }
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(scratch);
    return rc;
}

int tinsel_hip_selftest_scan(int device_index, const int* in, int* out, unsigned long long n)
{
    if (!in || !out || n == 0 || n >= (1ull << 31))
        return fail("selftest_scan: bad arguments");
    HIP_TRY(hipSetDevice(device_index));
    int *a = nullptr, *scratch = nullptr;
    int rc = 0;
    if (hipMalloc((void**)&a, n*sizeof(int)) != hipSuccess || hipMalloc((void**)&scratch, scan_scratch_ints((size_t)n)*sizeof(int)) != hipSuccess)
        rc = fail("selftest_scan: allocation failed");
    else if (hipMemcpy(a, in, n*sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
        rc = fail("selftest_scan: upload failed");
    else
    {
        exclusive_scan(a, a, (size_t)n, scratch, nullptr);         // (in place, as the builder's radix passes use it)
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess || hipMemcpy(out, a, n*sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail("selftest_scan: kernels failed");
    }
    (void)hipFree(a); (void)hipFree(scratch);
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
