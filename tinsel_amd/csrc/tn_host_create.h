// tn_host_create.h -- C-ABI: last_error, create (scene flattening + upload), destroy
// (part of the library's one host translation unit: included by tinsel_hip.hip, in this order, never on its own)
#pragma once

namespace { void comm_release(tinsel_hip* r); }      // (tn_host_group.h: the RCCL communicator of the process-per-GPU arm)

extern "C" {


const char* tinsel_hip_last_error(void) { return g_error.c_str(); }

void tinsel_hip_tuning_init(tinsel_hip_tuning* t)
{
    if (t)
        *t = tuning_defaults();
}

tinsel_hip* tinsel_hip_create(const tinsel_scene_desc* desc, int device_index) { return tinsel_hip_create_tuned(desc, device_index, nullptr); }

tinsel_hip* tinsel_hip_create_tuned(const tinsel_scene_desc* desc, int device_index, const tinsel_hip_tuning* tuning)
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
    r->tune = tuning_from_caller(tuning);
    const tinsel_hip_tuning& tune = r->tune;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_index) == hipSuccess)
    {
        r->numCUs = prop.multiProcessorCount;
        r->sharedMemLimit = (int)prop.sharedMemPerBlock;
    }
    // every kernel's dynamic-LDS limit is raised here, once, and the results are checked.  A runtime that refuses to raise one (it grants
    // less than the device reports) is not refused in turn: the renderer then plans every launch inside the 64 KB no attribute is needed
    // for -- 256-thread k_walk / k_swalk workgroups, no shading pools beside a large arena, k_seg_prefix's grids clamped -- and says so
    // once, by the kernel's name (a device that can render small scenes should)
    prepare_kernels_once(r);
    if (!r->prepRefused.empty() && r->sharedMemLimit > 65536)
    {
        fprintf(stderr, "tinsel_hip: the runtime refused %d B of dynamic LDS for %s: planning within 65536 B per workgroup\n", r->sharedMemLimit, r->prepRefused.c_str());
        r->sharedMemLimit = 65536;
        r->segPrefixLds = std::min(r->segPrefixLds, 65536 - 1024);
    }

    if (tune.batch_paths >= 65536)
    {
        r->maxBatchSlots = (size_t)tune.batch_paths;
        r->batchSlotsExplicit = true;
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
    const int inlineMaxTris = tune.inline_max_tris >= 0 ? tune.inline_max_tris : kInlineMaxTris;
    auto lives_in_arena = [&](size_t meshBytes, int numTris) {
        // tinsel_hip_tuning::small_mesh_bytes: test / A-B knob (0 = every mesh lives in HBM, so the queue sort and k_walk see them all)
        if (tune.small_mesh_bytes >= 0)
            return meshBytes <= (size_t)tune.small_mesh_bytes;
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
    r->lightPrims = lights;
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
        const bool flatScan = everyPrimHasALeaf && P <= 64 && tune.flat_scan != 0;

        // primitives whose mesh lives in HBM (flat-scan scenes, the first 7): their leaf-box test sorts the ray queues
        // (k_generate, k_shade), and they are walked by k_walk ahead of the scan kernels (tn_walk.h).  That includes trees
        // that stay in L1/L2 (glass.tin's 1280-triangle sphere: 80 KB of nodes; its 12-triangle cube): what the lean kernel
        // buys there is ray replacement for incoherent bounces (glass, maxDepth 12: 924 -> 1001 Msamples/s with the sphere,
        // 1050 with the cube too; with k_walk's work list in image order the sphere had lost, 732 inline vs 657-690).
        const int walkMinTris = tune.walk_min_tris >= 0 ? tune.walk_min_tris : kInlineMaxTris + 1;
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
        // quads that ride in the arena: where their node, triangles and normals are goes into the primitive record, so that the scan reaches them
        // without the mesh table's record in between (one dependent fetch less per ray that enters the quad's box: every shadow ray of cornell)
        for (int k = 0; k < P; ++k)
            if (prims[(size_t)k].type == kPrimMesh && meshes[prims[(size_t)k].mesh].twoLeaves && meshes[prims[(size_t)k].mesh].inArena)
            {
                const DevMesh& qm = meshes[prims[(size_t)k].mesh];
                if (((qm.offNodes | qm.offTris | qm.offNormals | qm.offCdf) & 127u) != 0u || ((qm.offNodes | qm.offTris | qm.offNormals | qm.offCdf) >> 7) > 0xffffu)
                    continue;
                Prim64& qp = prims[(size_t)k];
                qp.flags |= kPrimQuadArena;
                const uint32_t w0 = (qm.offNodes >> 7) | ((qm.offTris >> 7) << 16), w1 = (qm.offNormals >> 7) | ((qm.offCdf >> 7) << 16);
                memcpy(&qp.g0, &w0, 4); memcpy(&qp.g1, &w1, 4);
            }
        r->walkEnabled = tune.walk != 0;
#ifdef TN_WALK_PROF
        if (hipMalloc((void**)&r->walkProf, 16*sizeof(unsigned long long)) == hipSuccess)
            (void)hipMemset(r->walkProf, 0, 16*sizeof(unsigned long long));
#endif

        // one contiguous arena: scene BVH, Prim64, Mat128, moving poses, lights, mesh table (+ small meshes, added above)
        const size_t offNodes = arena.add(sceneBvh.nodes.data(), sceneBvh.nodes.size());
        const size_t offPrims = arena.add(prims.data(), prims.size());
        const size_t offMats = arena.add(mats.data(), mats.size());
        const size_t offMoving = arena.add(moving.data(), moving.size());
        std::vector<LightRec> lightRecs(lights.size());
        for (size_t l = 0; l < lights.size(); ++l)
            lightRecs[l] = { lights[l], mats[(size_t)lights[l]].lightSamples, mats[(size_t)lights[l]].rcpLightSamples, 0 };
        const size_t offLights = arena.add(lightRecs.data(), lightRecs.size());
        const size_t offMeshes = arena.add(meshes.data(), meshes.size());

        // The always-hit planes once more, four by four, for the flat scan (trace_flat): four IntersectRayPlane calls in a block are four
        // independent IEEE-division chains.  Where: scenes with a mesh in HBM (the split pipeline's kernels: glass k_extend -6 %, k_shadow
        // -5 %, round 4) and -- since k_bounce has the registers for it, round 5 -- fused scenes with at least one full block of four
        // (cornell 5283 -> 5450 Msamples/s, cfg1 +2.5 %); with fewer planes the table's remainder code only costs (gloss, env_loft -1 %:
        // profiles/r05_d_ab_plane_table_fused.md), so those keep the loop.
        std::vector<float> planeEq;
        std::vector<int32_t> planeIdx;
        int alwaysHitPlanes = 0;
        bool anyMeshInHbm = false;
        for (int k = 0; k < P; ++k)
        {
            alwaysHitPlanes += (prims[(size_t)k].type == kPrimPlane && boxes[(size_t)k].alwaysHit) ? 1 : 0;
            anyMeshInHbm = anyMeshInHbm || (prims[(size_t)k].type == kPrimMesh && !meshes[prims[(size_t)k].mesh].inArena);
        }
        if (flatScan && (alwaysHitPlanes >= 4 || anyMeshInHbm))
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
            const size_t ldsLimit = tune.arena_lds_limit >= 0 ? (size_t)tune.arena_lds_limit : kArenaLdsLimit;
            sc.arenaLdsBytes = (arena.bytes.size() <= ldsLimit && tune.lds_scene != 0) ? sc.arenaBytes : 0u;
            sc.nodes = reinterpret_cast<const Node64*>(arenaDev + offNodes);
            sc.prims = reinterpret_cast<const Prim64*>(arenaDev + offPrims);
            sc.mats = reinterpret_cast<const Mat128*>(arenaDev + offMats);
            sc.moving = reinterpret_cast<const Moving64*>(arenaDev + offMoving);
            sc.lights = reinterpret_cast<const LightRec*>(arenaDev + offLights);
            sc.meshes = reinterpret_cast<const DevMesh*>(arenaDev + offMeshes);
            sc.numMeshes = (int)meshes.size();
            r->sceneStackNeed = sceneBvh.maxLeafDepth + 1;
            sc.primBoxes = reinterpret_cast<const PrimBox*>(arenaDev + offBoxes);
            sc.planeEq = reinterpret_cast<const float4*>(arenaDev + offPlaneEq);
            sc.planeIdx = reinterpret_cast<const int32_t*>(arenaDev + offPlaneIdx);
            sc.numPlanes = (int32_t)r->planeTablePrims.size();
            sc.scanMask = 0;
            for (int k = 0; k < P && k < 64; ++k)
                if (boxes[(size_t)k].alwaysHit != 2u)
                    sc.scanMask |= 1ull << k;
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
    comm_release(r);
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

} // extern "C"
