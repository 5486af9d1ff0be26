// tn_host_bvh_build.h -- mesh BVHs built on the device (tn_lbvh.h, tn_sort.h)
// (part of the library's one host translation unit: included by tinsel_hip.hip, in this order, never on its own)
#pragma once

// ---------------------------------------------------------------------------
// device-side mesh BVH build (tn_lbvh.h); the reference trees stay the default and the parity path

namespace {

// one device allocation carved into aligned pieces (hipMalloc / hipFree dominate a small build otherwise)
struct ScratchPool
{
    unsigned char* base = nullptr;
    size_t size = 0, used = 0;
    ~ScratchPool() { if (base) (void)hipFree(base); }
    static size_t padded(size_t bytes) { return (bytes + 255) & ~size_t(255); }
    bool reserve(size_t bytes) { size = bytes; return hipMalloc((void**)&base, bytes ? bytes : 1) == hipSuccess; }
    template <class T> T* get(size_t count)
    {
        T* p = (T*)(base + used);
        used += padded(sizeof(T)*count);
        return used <= size ? p : nullptr;
    }
};

// Builds a BVH over mesh `dm`'s triangles on the device -- TINSEL_BVH_LBVH: Karras' hierarchy over the Morton order + box fitting level by
// level; TINSEL_BVH_PLOC: agglomerative clustering over the same order (tn_lbvh.h) -- and emits it with its top numbered breadth-first
// (k_walk stages a prefix of the node array into LDS).  On success fills nodes / root / stackNeed / topCount of `out`.
int build_device_bvh(tinsel_hip* r, const DevMesh& dm, DevMesh& out, int mode)
{
    const int n = dm.numTris;
    // scratch of the library's own sort and scan (tn_sort.h), in bytes
    const size_t sortBytes = sort_scratch_ints((size_t)n)*sizeof(int), scanBytes = scan_scratch_ints((size_t)n)*sizeof(int);
    const size_t N = (size_t)n;
    ScratchPool tmp;
    if (!tmp.reserve(ScratchPool::padded(6*4) + 2*ScratchPool::padded(N*8) + ScratchPool::padded((N - 1)*8) + 2*ScratchPool::padded((2*N - 1)*4) +
                     ScratchPool::padded((2*N - 1)*24) + 7*ScratchPool::padded(N*4) + ScratchPool::padded(sortBytes) + ScratchPool::padded(scanBytes) +
                     ScratchPool::padded(kWalkTopNodes*4) + 256))
        return fail("build_mesh_bvh: device allocation failed");
    uint32_t* bounds = tmp.get<uint32_t>(6);
    unsigned long long* keys = tmp.get<unsigned long long>(N);
    unsigned long long* sorted = tmp.get<unsigned long long>(N);
    int2* children = tmp.get<int2>(N - 1);
    int* parent = tmp.get<int>(2*N - 1);
    float* boxes = tmp.get<float>((2*N - 1)*6);
    int* height = tmp.get<int>(2*N - 1);
    int* visits = tmp.get<int>(N);          // LBVH: the fitting passes' generations; PLOC: nearest neighbours
    int* clustersA = tmp.get<int>(N);
    int* clustersB = tmp.get<int>(N);
    int* keep = tmp.get<int>(N);
    int* offsets = tmp.get<int>(N);
    int* isTop = tmp.get<int>(N);
    int* perm = tmp.get<int>(N);
    unsigned char* sortTmp = tmp.get<unsigned char>(sortBytes);
    unsigned char* scanTmp = tmp.get<unsigned char>(scanBytes);
    int* topIds = tmp.get<int>(kWalkTopNodes);
    int* nextId = tmp.get<int>(2);          // [0] the next internal node id, [1] clusters left after a round
    Node64* nodes = nullptr;
    if (!bounds || !keys || !sorted || !children || !parent || !boxes || !height || !visits || !clustersA || !clustersB || !keep || !offsets || !isTop ||
        !perm || !sortTmp || !scanTmp || !topIds || !nextId || hipMalloc((void**)&nodes, sizeof(Node64)*(N - 1)) != hipSuccess)
        return fail("build_mesh_bvh: device allocation failed");

    const uint32_t init[6] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u };
    const unsigned grid = (unsigned)((n + 255)/256);
    int rc = 0;
    do
    {
        if (hipMemcpyAsync(bounds, init, sizeof(init), hipMemcpyHostToDevice, nullptr) != hipSuccess ||
            hipMemsetAsync(visits, 0, sizeof(int)*(size_t)n, nullptr) != hipSuccess) { rc = fail("build_mesh_bvh: init failed"); break; }
        hipLaunchKernelGGL(k_lbvh_bounds, dim3(grid < 256u ? grid : 256u), dim3(256), 0, nullptr, dm.tris, n, bounds);
        hipLaunchKernelGGL(k_lbvh_keys, dim3(grid), dim3(256), 0, nullptr, dm.tris, n, bounds, keys);
        // keys = Morton code << 32 | triangle index, written in index order: a STABLE sort by the code's bytes (bits 32..63) is the sort by the
        // whole key
        radix_sort_keys(keys, sorted, (size_t)n, 32, 64, reinterpret_cast<int*>(sortTmp), nullptr);
        hipLaunchKernelGGL(k_lbvh_leaves, dim3(grid), dim3(256), 0, nullptr, dm.tris, sorted, n, boxes, height);
        if (mode == TINSEL_BVH_PLOC)
        {
            // agglomerative rounds over the Morton order; the host reads the number of clusters left after every round (8 B)
            const int firstId = n - 2;
            if (hipMemcpyAsync(nextId, &firstId, sizeof(int), hipMemcpyHostToDevice, nullptr) != hipSuccess) { rc = fail("build_mesh_bvh: init failed"); break; }
            hipLaunchKernelGGL(k_ploc_init, dim3(grid), dim3(256), 0, nullptr, n, clustersA);
            int c = n;
            int* cur = clustersA;
            int* nxt = clustersB;
            int rounds = 0;
            while (c > 1 && !rc)
            {
                const unsigned g = (unsigned)((c + 255)/256);
                hipLaunchKernelGGL(k_ploc_nearest, dim3(g), dim3(256), 0, nullptr, (const int*)cur, c, (const float*)boxes, visits);
                hipLaunchKernelGGL(k_ploc_merge, dim3(g), dim3(256), 0, nullptr, cur, c, (const int*)visits, boxes, children, height, nextId, keep);
                exclusive_scan(keep, offsets, (size_t)c, reinterpret_cast<int*>(scanTmp), nullptr);
                hipLaunchKernelGGL(k_ploc_compact, dim3(g), dim3(256), 0, nullptr, (const int*)cur, c, (const int*)keep, (const int*)offsets, nxt, nextId + 1);
                int left = 0;
                if (hipMemcpy(&left, nextId + 1, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail("build_mesh_bvh: read-back failed"); break; }
                if (left >= c || left < 1 || ++rounds > 4096) { rc = fail("build_mesh_bvh: clustering made no progress"); break; }
                c = left;
                std::swap(cur, nxt);
            }
            if (rc)
                break;
        }
        else
        {
            hipLaunchKernelGGL(k_lbvh_hierarchy, dim3(grid), dim3(256), 0, nullptr, sorted, n, children, parent);
            // one pass per tree level (<= 63 for 62-bit keys); look at the root every 16 passes
            int rootGen = 0;
            for (int pass = 2; pass <= 66 && !rootGen; )
            {
                for (int k = 0; k < 16; ++k, ++pass)
                    hipLaunchKernelGGL(k_lbvh_fit_pass, dim3(grid), dim3(256), 0, nullptr, n, pass, children, boxes, height, visits);
                if (hipMemcpy(&rootGen, visits, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
                    break;
            }
            if (!rootGen) { rc = fail("build_mesh_bvh: box fitting did not reach the root"); break; }
        }

        // the tree's top, breadth-first: the host walks the first kWalkTopNodes internal nodes (children: 8 B per node)
        std::vector<int2> hostChildren(N - 1);
        if (hipMemcpy(hostChildren.data(), children, sizeof(int2)*(N - 1), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail("build_mesh_bvh: read-back failed"); break; }
        std::vector<int> topOrder;
        topOrder.reserve(kWalkTopNodes);
        {
            std::vector<int> frontier(1, 0);
            size_t head = 0;
            while (head < frontier.size() && (int)topOrder.size() < kWalkTopNodes)
            {
                const int id = frontier[head++];
                topOrder.push_back(id);
                const int2 ch = hostChildren[(size_t)id];
                if (ch.x < n - 1) frontier.push_back(ch.x);
                if (ch.y < n - 1) frontier.push_back(ch.y);
            }
        }
        const int top = (int)topOrder.size();
        if (hipMemsetAsync(isTop, 0, sizeof(int)*(N - 1), nullptr) != hipSuccess ||
            hipMemcpy(topIds, topOrder.data(), sizeof(int)*(size_t)top, hipMemcpyHostToDevice) != hipSuccess) { rc = fail("build_mesh_bvh: upload failed"); break; }
        int* rank = keep;           // (free again)
        hipLaunchKernelGGL(k_bfs_mark, dim3((unsigned)((top + 255)/256)), dim3(256), 0, nullptr, (const int*)topIds, top, isTop, rank);
        exclusive_scan(isTop, offsets, N - 1, reinterpret_cast<int*>(scanTmp), nullptr);
        hipLaunchKernelGGL(k_bfs_perm, dim3(grid), dim3(256), 0, nullptr, n - 1, top, (const int*)isTop, (const int*)rank, (const int*)offsets, perm);
        hipLaunchKernelGGL(k_lbvh_emit_perm, dim3(grid), dim3(256), 0, nullptr, sorted, n, children, boxes, (const int*)perm, nodes);
        int rootHeight = 0;
        if (hipGetLastError() != hipSuccess || hipMemcpy(&rootHeight, height, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail("build_mesh_bvh: kernels failed"); break; }
        out = dm;
        out.nodes = nodes;
        out.root = 0;                   // perm[0] == 0: the root is the first node of the breadth-first walk
        out.stackNeed = rootHeight + 1;
        out.topCount = top;
        out.numInternal = n - 1;
    } while (false);
    if (rc)
        (void)hipFree(nodes);
    else
        r->lbvhAllocs.push_back(nodes);
    return rc;
}

} // namespace
