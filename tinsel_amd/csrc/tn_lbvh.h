// tn_lbvh.h -- device-side BVH construction for large meshes (opt-in).
//
// The reference builds every mesh BVH on the host with a SAH sweep that std::sorts the items at every node
// (bvh.h:30-263: 172 ms for 107 k triangles) and ships that tree; the parity path keeps it, because the visit
// order of a different tree resolves exact-t ties differently.  This is the alternative for meshes that are
// re-built often or are too large to wait for: a linear BVH over the triangle centroids' 30-bit Morton codes
// (keys made unique with the triangle index), radix-sorted, hierarchy by longest common prefix, boxes fitted
// bottom-up.  One triangle per leaf, like the reference's trees, emitted straight into the traversal layout
// (Node64: both children's boxes and refs per internal node, leaves referenced by triangle index).
//
// All kernels are streaming passes over n triangles: HBM-bound, one launch each.
#pragma once

#include "tn_scene.h"

namespace tn {

// order-preserving float <-> uint (for atomicMin / atomicMax on floats of either sign)
TN_D uint32_t float_ordered(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
TN_D float ordered_float(uint32_t u)
{
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

TN_D void tri_box(const Tri48* tris, int i, float* lo, float* hi)
{
    const float4* tp = reinterpret_cast<const float4*>(tris + i);
    const float4 a = tp[0], b = tp[1], c = tp[2];
    lo[0] = fminf(a.x, fminf(b.x, c.x)); hi[0] = fmaxf(a.x, fmaxf(b.x, c.x));
    lo[1] = fminf(a.y, fminf(b.y, c.y)); hi[1] = fmaxf(a.y, fmaxf(b.y, c.y));
    lo[2] = fminf(a.z, fminf(b.z, c.z)); hi[2] = fmaxf(a.z, fmaxf(b.z, c.z));
}

// bounds[0..2] = min, bounds[3..5] = max of the triangle CENTROIDS (ordered-uint encoded); init to ~0 / 0.
// Fixed grid, grid-stride loop, block reduction: one atomic pair per BLOCK per axis (single-address atomics are slow).
__global__ __launch_bounds__(256) void k_lbvh_bounds(const Tri48* __restrict__ tris, int n, uint32_t* __restrict__ bounds)
{
    uint32_t mn[3] = { 0xffffffffu, 0xffffffffu, 0xffffffffu }, mx[3] = { 0u, 0u, 0u };
    for (int i = blockIdx.x*256 + threadIdx.x; i < n; i += gridDim.x*256)
    {
        float lo[3], hi[3];
        tri_box(tris, i, lo, hi);
        for (int k = 0; k < 3; ++k)
        {
            const uint32_t c = float_ordered(0.5f*(lo[k] + hi[k]));
            mn[k] = c < mn[k] ? c : mn[k];
            mx[k] = c > mx[k] ? c : mx[k];
        }
    }
    __shared__ uint32_t s_mn[4][3], s_mx[4][3];
    for (int k = 0; k < 3; ++k)
    {
        for (int off = 32; off > 0; off >>= 1)
        {
            const uint32_t a = __shfl_down(mn[k], off), b = __shfl_down(mx[k], off);
            mn[k] = a < mn[k] ? a : mn[k];
            mx[k] = b > mx[k] ? b : mx[k];
        }
        if ((threadIdx.x & 63) == 0)
        {
            s_mn[threadIdx.x >> 6][k] = mn[k];
            s_mx[threadIdx.x >> 6][k] = mx[k];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3)
    {
        const int k = threadIdx.x;
        uint32_t a = s_mn[0][k], b = s_mx[0][k];
        for (int w = 1; w < 4; ++w)
        {
            a = s_mn[w][k] < a ? s_mn[w][k] : a;
            b = s_mx[w][k] > b ? s_mx[w][k] : b;
        }
        atomicMin(bounds + k, a);
        atomicMax(bounds + 3 + k, b);
    }
}

TN_D uint32_t expand_bits10(uint32_t v)     // 10 bits -> every third bit
{
    v = (v*0x00010001u) & 0xFF0000FFu;
    v = (v*0x00000101u) & 0x0F00F00Fu;
    v = (v*0x00000011u) & 0xC30C30C3u;
    v = (v*0x00000005u) & 0x49249249u;
    return v;
}

__global__ __launch_bounds__(256) void k_lbvh_keys(const Tri48* __restrict__ tris, int n, const uint32_t* __restrict__ bounds,
                                                   unsigned long long* __restrict__ keys)
{
    const int i = blockIdx.x*256 + threadIdx.x;
    if (i >= n)
        return;
    float lo[3], hi[3];
    tri_box(tris, i, lo, hi);
    uint32_t code = 0;
    for (int k = 0; k < 3; ++k)
    {
        const float mn = ordered_float(bounds[k]), mx = ordered_float(bounds[3 + k]);
        const float ext = mx - mn;
        float u = ext > 0.0f ? (0.5f*(lo[k] + hi[k]) - mn)/ext : 0.0f;
        u = fminf(fmaxf(u*1024.0f, 0.0f), 1023.0f);
        code |= expand_bits10((uint32_t)u) << (2 - k);
    }
    keys[i] = ((unsigned long long)code << 32) | (unsigned long long)(uint32_t)i;
}

// longest common prefix of two sorted keys (unique: the low word is the triangle index); -1 outside the array
TN_D int lbvh_delta(const unsigned long long* keys, int n, int i, int j)
{
    if (j < 0 || j >= n)
        return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));
}

// Node ids: internal nodes 0..n-2 (root = 0), leaf of sorted position j = (n-1) + j.
// children[i] = (left id, right id) of internal node i; parent[id] for every node but the root.
__global__ __launch_bounds__(256) void k_lbvh_hierarchy(const unsigned long long* __restrict__ keys, int n, int2* __restrict__ children,
                                                        int* __restrict__ parent)
{
    const int i = blockIdx.x*256 + threadIdx.x;
    if (i >= n - 1)
        return;
    const int d = (lbvh_delta(keys, n, i, i + 1) - lbvh_delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = lbvh_delta(keys, n, i, i - d);
    int lmax = 2;
    while (lbvh_delta(keys, n, i, i + lmax*d) > dmin)
        lmax *= 2;
    int l = 0;
    for (int t = lmax/2; t >= 1; t /= 2)
        if (lbvh_delta(keys, n, i, i + (l + t)*d) > dmin)
            l += t;
    const int j = i + l*d;
    const int dnode = lbvh_delta(keys, n, i, j);
    int s = 0;
    int t = l;
    do
    {
        t = (t + 1)/2;
        if (lbvh_delta(keys, n, i, i + (s + t)*d) > dnode)
            s += t;
    } while (t > 1);
    const int gamma = i + s*d + (d < 0 ? d : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int left = (lo == gamma) ? (n - 1 + gamma) : gamma;
    const int right = (hi == gamma + 1) ? (n - 1 + gamma + 1) : (gamma + 1);
    children[i] = make_int2(left, right);
    parent[left] = i;
    parent[right] = i;
}

// Box fitting, bottom-up, WITHOUT inter-thread hand-off inside a kernel (an atomic climb needs an agent-scope
// fence per level per thread, which on 8 XCDs with private L2s costs a cache write-back each: 4 ms for 524 k
// triangles).  Instead: leaves first, then one pass per tree level -- an internal node is fitted in pass k when
// both children were finished in an EARLIER pass (gen[child] in [1, k)); kernel boundaries are the fences.
// A tree over 62-bit unique keys is at most 63 levels high, so 63 passes always suffice; the host stops earlier
// when the root is done.   boxes[id] = {min.xyz, max.xyz} for all 2n-1 nodes; gen[] zero-initialised.
__global__ __launch_bounds__(256) void k_lbvh_leaves(const Tri48* __restrict__ tris, const unsigned long long* __restrict__ keys, int n,
                                                     float* __restrict__ boxes, int* __restrict__ height)
{
    const int j = blockIdx.x*256 + threadIdx.x;
    if (j >= n)
        return;
    const int id = n - 1 + j;
    float lo[3], hi[3];
    tri_box(tris, (int)(uint32_t)keys[j], lo, hi);
    float* b = boxes + (size_t)id*6;
    b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2]; b[3] = hi[0]; b[4] = hi[1]; b[5] = hi[2];
    height[id] = 0;
}

__global__ __launch_bounds__(256) void k_lbvh_fit_pass(int n, int pass, const int2* __restrict__ children, float* __restrict__ boxes,
                                                       int* __restrict__ height, int* __restrict__ gen)
{
    const int i = blockIdx.x*256 + threadIdx.x;
    if (i >= n - 1 || gen[i] != 0)
        return;
    const int2 ch = children[i];
    const int gl = ch.x >= n - 1 ? 1 : gen[ch.x];
    const int gr = ch.y >= n - 1 ? 1 : gen[ch.y];
    const bool leftDone = ch.x >= n - 1 || (gl != 0 && gl < pass);
    const bool rightDone = ch.y >= n - 1 || (gr != 0 && gr < pass);
    if (!leftDone || !rightDone)
        return;
    const float* bl = boxes + (size_t)ch.x*6;
    const float* br = boxes + (size_t)ch.y*6;
    float* bo = boxes + (size_t)i*6;
    for (int k = 0; k < 3; ++k)
    {
        bo[k] = fminf(bl[k], br[k]);
        bo[3 + k] = fmaxf(bl[3 + k], br[3 + k]);
    }
    const int hl = height[ch.x], hr = height[ch.y];
    height[i] = 1 + (hl > hr ? hl : hr);
    gen[i] = pass;
}

__global__ __launch_bounds__(256) void k_lbvh_emit(const unsigned long long* __restrict__ keys, int n, const int2* __restrict__ children,
                                                   const float* __restrict__ boxes, Node64* __restrict__ out)
{
    const int i = blockIdx.x*256 + threadIdx.x;
    if (i >= n - 1)
        return;
    const int2 ch = children[i];
    const float* bl = boxes + (size_t)ch.x*6;
    const float* br = boxes + (size_t)ch.y*6;
    Node64 o;
    o.lminx = bl[0]; o.lminy = bl[1]; o.lminz = bl[2]; o.lmaxx = bl[3]; o.lmaxy = bl[4]; o.lmaxz = bl[5];
    o.rminx = br[0]; o.rminy = br[1]; o.rminz = br[2]; o.rmaxx = br[3]; o.rmaxy = br[4]; o.rmaxz = br[5];
    o.left = ch.x >= n - 1 ? (kLeafBit | (uint32_t)keys[ch.x - (n - 1)]) : (uint32_t)ch.x;
    o.right = ch.y >= n - 1 ? (kLeafBit | (uint32_t)keys[ch.y - (n - 1)]) : (uint32_t)ch.y;
    o.pad0 = 0; o.pad1 = 0;
    out[i] = o;
}

// ---------------------------------------------------------------------------
// PLOC (parallel locally-ordered clustering; Meister & Bittner 2018) -- TINSEL_BVH_PLOC: a device build whose trees are close to the
// reference's SAH trees in quality (an LBVH splits where the Morton code says, whatever the boxes look like: twice as deep on the
// 524k-triangle mesh, 13 % slower to render).  Bottom-up and agglomerative over the SAME Morton order: the active clusters sit in
// an array in Morton order; every round each cluster looks kPlocRadius neighbours to either side for the one whose union with it has
// the smallest surface area; two clusters that choose EACH OTHER merge into a new node (box = the union: nothing to fit afterwards,
// height = 1 + max), the array is compacted keeping its order, until one cluster is left.  Node ids as in the LBVH build: leaf of
// sorted position j = (n - 1) + j, internal nodes n - 2 down to 0 in the order they are made, so the last one -- the root -- is 0.

constexpr int kPlocRadius = 8;

TN_D float ploc_area(const float* a, const float* b)
{
    const float dx = fmaxf(a[3], b[3]) - fminf(a[0], b[0]);
    const float dy = fmaxf(a[4], b[4]) - fminf(a[1], b[1]);
    const float dz = fmaxf(a[5], b[5]) - fminf(a[2], b[2]);
    return dx*dy + dy*dz + dz*dx;
}

// clusters[i] = node id of the i-th active cluster; nn[i] = index of its best neighbour within the radius
__global__ __launch_bounds__(256) void k_ploc_nearest(const int* __restrict__ clusters, int c, const float* __restrict__ boxes, int* __restrict__ nn)
{
    const int i = blockIdx.x*256 + threadIdx.x;
    if (i >= c)
        return;
    float bi[6];
    const float* src = boxes + (size_t)clusters[i]*6;
    for (int q = 0; q < 6; ++q)
        bi[q] = src[q];
    float best = 3.0e38f;
    int bestJ = -1;
    const int lo = i - kPlocRadius < 0 ? 0 : i - kPlocRadius, hi = i + kPlocRadius >= c ? c - 1 : i + kPlocRadius;
    for (int j = lo; j <= hi; ++j)
    {
        if (j == i)
            continue;
        const float a = ploc_area(bi, boxes + (size_t)clusters[j]*6);
        if (a < best)           // ties: the LOWEST index wins (j ascends), on both sides of a pair -- so mutual choices still exist
        {
            best = a;
            bestJ = j;
        }
    }
    nn[i] = bestJ;
}

// mutual nearest neighbours merge (the lower index makes the node and stays, the higher one leaves); keep[i] = 1 for clusters that stay
__global__ __launch_bounds__(256) void k_ploc_merge(int* __restrict__ clusters, int c, const int* __restrict__ nn, float* __restrict__ boxes,
                                                    int2* __restrict__ children, int* __restrict__ height, int* __restrict__ nextId, int* __restrict__ keep)
{
    const int i = blockIdx.x*256 + threadIdx.x;
    if (i >= c)
        return;
    const int j = nn[i];
    int stay = 1;
    if (j >= 0 && nn[j] == i)
    {
        if (i < j)
        {
            const int id = atomicSub(nextId, 1);        // n - 2, n - 3, ... 0
            const int a = clusters[i], b = clusters[j];
            children[id] = make_int2(a, b);
            const float* ba = boxes + (size_t)a*6;
            const float* bb = boxes + (size_t)b*6;
            float* bo = boxes + (size_t)id*6;
            for (int q = 0; q < 3; ++q)
            {
                bo[q] = fminf(ba[q], bb[q]);
                bo[3 + q] = fmaxf(ba[3 + q], bb[3 + q]);
            }
            const int ha = height[a], hb = height[b];
            height[id] = 1 + (ha > hb ? ha : hb);
            clusters[i] = id;
        }
        else
            stay = 0;
    }
    keep[i] = stay;
}

// order-preserving compaction: offsets = exclusive scan of keep
__global__ __launch_bounds__(256) void k_ploc_compact(const int* __restrict__ clusters, int c, const int* __restrict__ keep, const int* __restrict__ offsets,
                                                      int* __restrict__ out, int* __restrict__ left)
{
    const int i = blockIdx.x*256 + threadIdx.x;
    if (i < c && keep[i])
        out[offsets[i]] = clusters[i];
    if (i == c - 1)
        *left = offsets[i] + keep[i];           // clusters left after this round: the one word the host reads back
}

__global__ __launch_bounds__(256) void k_ploc_init(int n, int* __restrict__ clusters)
{
    const int j = blockIdx.x*256 + threadIdx.x;
    if (j < n)
        clusters[j] = n - 1 + j;
}

// ---------------------------------------------------------------------------
// Breadth-first numbering of a device-built tree's top (k_walk stages a PREFIX of the node array into LDS, tn_walk.h: any prefix of a
// breadth-first numbering is the top of the tree).  The host walks the first `top` internal nodes breadth-first (children[] downloaded:
// 8 B per node) and uploads their ids in that order; here: isTop flags -> exclusive scan -> perm[id] = rank among the top nodes, or
// top + (id - top nodes before it) for the others (which keep their relative order); emission applies perm to a node's own slot and
// to its internal children's refs.
__global__ __launch_bounds__(256) void k_bfs_mark(const int* __restrict__ topIds, int top, int* __restrict__ isTop, int* __restrict__ rank)
{
    const int k = blockIdx.x*256 + threadIdx.x;
    if (k < top)
    {
        isTop[topIds[k]] = 1;
        rank[topIds[k]] = k;
    }
}

__global__ __launch_bounds__(256) void k_bfs_perm(int numInternal, int top, const int* __restrict__ isTop, const int* __restrict__ rank,
                                                  const int* __restrict__ before, int* __restrict__ perm)
{
    const int id = blockIdx.x*256 + threadIdx.x;
    if (id < numInternal)
        perm[id] = isTop[id] ? rank[id] : top + (id - before[id]);
}

__global__ __launch_bounds__(256) void k_lbvh_emit_perm(const unsigned long long* __restrict__ keys, int n, const int2* __restrict__ children,
                                                        const float* __restrict__ boxes, const int* __restrict__ perm, Node64* __restrict__ out)
{
    const int i = blockIdx.x*256 + threadIdx.x;
    if (i >= n - 1)
        return;
    const int2 ch = children[i];
    const float* bl = boxes + (size_t)ch.x*6;
    const float* br = boxes + (size_t)ch.y*6;
    Node64 o;
    o.lminx = bl[0]; o.lminy = bl[1]; o.lminz = bl[2]; o.lmaxx = bl[3]; o.lmaxy = bl[4]; o.lmaxz = bl[5];
    o.rminx = br[0]; o.rminy = br[1]; o.rminz = br[2]; o.rmaxx = br[3]; o.rmaxy = br[4]; o.rmaxz = br[5];
    o.left = ch.x >= n - 1 ? (kLeafBit | (uint32_t)keys[ch.x - (n - 1)]) : (uint32_t)perm[ch.x];
    o.right = ch.y >= n - 1 ? (kLeafBit | (uint32_t)keys[ch.y - (n - 1)]) : (uint32_t)perm[ch.y];
    o.pad0 = 0; o.pad1 = 0;
    out[perm[i]] = o;
}

// ---------------------------------------------------------------------------
// Refit (tinsel_hip_refit_mesh): the vertices of a mesh moved, its topology did not.  The tree keeps its shape -- the
// reference's SAH tree or a device-built one, both are Node64 arrays -- and every box is recomputed bottom-up from the
// new triangles: a leaf's box is the min / max of its three vertices (Bounds::AddPoint, maths.h), an internal node's
// the union of its children's, exactly what the reference's builder stores for the same tree (min / max do not round).

// new positions -> the pre-gathered triangle records (vertex indices ride in the .w lanes)
__global__ __launch_bounds__(256) void k_refit_tris(Tri48* __restrict__ tris, int numTris, const float* __restrict__ positions)
{
    const int t = blockIdx.x*256 + threadIdx.x;
    if (t >= numTris)
        return;
    Tri48 T = tris[t];
    T.ax = positions[T.i0*3 + 0]; T.ay = positions[T.i0*3 + 1]; T.az = positions[T.i0*3 + 2];
    T.bx = positions[T.i1*3 + 0]; T.by = positions[T.i1*3 + 1]; T.bz = positions[T.i1*3 + 2];
    T.cx = positions[T.i2*3 + 0]; T.cy = positions[T.i2*3 + 1]; T.cz = positions[T.i2*3 + 2];
    tris[t] = T;
}

// One pass per tree level, like the box fitting of the build: a node is refitted once both children were finished in an
// EARLIER pass (leaves always are), so no thread ever reads what another one writes in the same launch.
__global__ __launch_bounds__(256) void k_refit_pass(Node64* __restrict__ nodes, int numNodes, const Tri48* __restrict__ tris,
                                                    float* __restrict__ own, int* __restrict__ gen, int pass)
{
    const int k = blockIdx.x*256 + threadIdx.x;
    if (k >= numNodes || gen[k] != 0)
        return;
    Node64 n = nodes[k];
    float b[2][6];
    const uint32_t child[2] = { n.left, n.right };
    for (int c = 0; c < 2; ++c)
    {
        if (child[c] & kLeafBit)
        {
            const Tri48 T = tris[child[c] & ~kLeafBit];
            b[c][0] = fminf(fminf(T.ax, T.bx), T.cx); b[c][1] = fminf(fminf(T.ay, T.by), T.cy); b[c][2] = fminf(fminf(T.az, T.bz), T.cz);
            b[c][3] = fmaxf(fmaxf(T.ax, T.bx), T.cx); b[c][4] = fmaxf(fmaxf(T.ay, T.by), T.cy); b[c][5] = fmaxf(fmaxf(T.az, T.bz), T.cz);
        }
        else
        {
            const int g = gen[child[c]];
            if (g == 0 || g >= pass)
                return;
            for (int q = 0; q < 6; ++q)
                b[c][q] = own[(size_t)child[c]*6 + q];
        }
    }
    n.lminx = b[0][0]; n.lminy = b[0][1]; n.lminz = b[0][2]; n.lmaxx = b[0][3]; n.lmaxy = b[0][4]; n.lmaxz = b[0][5];
    n.rminx = b[1][0]; n.rminy = b[1][1]; n.rminz = b[1][2]; n.rmaxx = b[1][3]; n.rmaxy = b[1][4]; n.rmaxz = b[1][5];
    nodes[k] = n;
    for (int q = 0; q < 3; ++q)
    {
        own[(size_t)k*6 + q] = fminf(b[0][q], b[1][q]);
        own[(size_t)k*6 + 3 + q] = fmaxf(b[0][3 + q], b[1][3 + q]);
    }
    gen[k] = pass;
}

} // namespace tn
