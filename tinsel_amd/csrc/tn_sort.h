// tn_sort.h -- the two primitives the device BVH builder (tn_lbvh.h) needs besides its own kernels: an exclusive prefix sum of ints and
// a stable least-significant-digit radix sort of 64-bit keys.  Hand-written for wave64 / 64 KB of static LDS; they replaced the rocprim
// calls of rounds 1-4 (which brought ~300 trampoline kernels for other architectures into the library and half of its compile time).
// Neither is on the render path: a mesh's tree is built once (opt-in, tinsel_hip_set_mesh_bvh) -- sizes are 1e3..1e6 elements.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

namespace tn {

// ---- exclusive scan ---------------------------------------------------------------------------------------------------
// Three launches: every workgroup scans its tile of kScanTile ints and leaves the tile's total; ONE workgroup scans the totals;
// every workgroup adds its tile's offset.  In place (out == in) is allowed.
constexpr int kScanBlock = 256;
constexpr int kScanPerThread = 8;
constexpr int kScanTile = kScanBlock*kScanPerThread;

// exclusive scan of the workgroup's kScanBlock values through LDS; returns this thread's prefix, *total the workgroup's sum
__device__ inline int block_scan_excl(int v, int* s_wave /*[kScanBlock/64]*/, int* total)
{
    const int lane = (int)__lane_id(), wave = (int)threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
    {
        const int up = __shfl_up(incl, off);
        if (lane >= off)
            incl += up;
    }
    if (lane == 63)
        s_wave[wave] = incl;
    __syncthreads();
    int base = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < kScanBlock/64; ++w)
    {
        const int t = s_wave[w];
        if (w < wave)
            base += t;
        sum += t;
    }
    __syncthreads();
    *total = sum;
    return base + incl - v;
}

// (`in` and `out` may be the SAME array -- radix_sort_keys scans its digit counts in place -- so neither is __restrict__: ADVICE r05)
__global__ __launch_bounds__(kScanBlock) void k_scan_tiles(const int* in, int* out, int n, int* __restrict__ tileSums)
{
    __shared__ int s_wave[kScanBlock/64];
    const int first = blockIdx.x*kScanTile + (int)threadIdx.x*kScanPerThread;
    int v[kScanPerThread], mine = 0;
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k)
    {
        v[k] = first + k < n ? in[first + k] : 0;
        mine += v[k];
    }
    int total;
    int run = block_scan_excl(mine, s_wave, &total);
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k)
    {
        if (first + k < n)
            out[first + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 0)
        tileSums[blockIdx.x] = total;
}

// ONE workgroup: exclusive scan of the `tiles` tile sums, in place (a thread takes a contiguous run of them)
__global__ __launch_bounds__(kScanBlock) void k_scan_sums(int* __restrict__ tileSums, int tiles)
{
    __shared__ int s_wave[kScanBlock/64];
    const int per = (tiles + kScanBlock - 1)/kScanBlock;
    const int first = (int)threadIdx.x*per;
    int mine = 0;
    for (int k = 0; k < per; ++k)
        if (first + k < tiles)
            mine += tileSums[first + k];
    int total;
    int run = block_scan_excl(mine, s_wave, &total);
    for (int k = 0; k < per; ++k)
    {
        if (first + k < tiles)
        {
            const int t = tileSums[first + k];
            tileSums[first + k] = run;
            run += t;
        }
    }
}

__global__ __launch_bounds__(kScanBlock) void k_scan_add(int* __restrict__ out, int n, const int* __restrict__ tileSums)
{
    const int add = tileSums[blockIdx.x];
    const int first = blockIdx.x*kScanTile + (int)threadIdx.x*kScanPerThread;
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k)
        if (first + k < n)
            out[first + k] += add;
}

inline size_t scan_scratch_ints(size_t n) { return (n + kScanTile - 1)/kScanTile + 1; }

// out[i] = in[0] + .. + in[i - 1]; `scratch`: scan_scratch_ints(n) ints.  Launches only (no synchronisation).
inline void exclusive_scan(const int* in, int* out, size_t n, int* scratch, hipStream_t st)
{
    if (n == 0)
        return;
    const int tiles = (int)((n + kScanTile - 1)/kScanTile);
    hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)tiles), dim3(kScanBlock), 0, st, in, out, (int)n, scratch);
    if (tiles > 1)
    {
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kScanBlock), 0, st, scratch, tiles);
        hipLaunchKernelGGL(k_scan_add, dim3((unsigned)tiles), dim3(kScanBlock), 0, st, out, (int)n, (const int*)scratch);
    }
}

// ---- stable LSD radix sort of 64-bit keys, 8 bits per pass ---------------------------------------------------------------
// A workgroup of kSortBlock threads owns a tile of kSortTile consecutive keys, a thread kSortPerThread consecutive ones: the order
// (tile, thread, position) IS the input order, so counting per (thread, digit) and scattering in that order is stable without any
// ranking inside a wave.  Per pass: k_sort_count (digit counts per tile, digit-major), one exclusive scan over [digit][tile], and
// k_sort_scatter (per-thread counts again in LDS -- u16 [thread][digit], 64 KB -- prefixed over the threads, then the keys go out).
constexpr int kSortBlock = 128;
constexpr int kSortPerThread = 16;
constexpr int kSortTile = kSortBlock*kSortPerThread;        // 2048: a per-(thread, digit) prefix fits 16 bits
constexpr int kSortDigits = 256;

__global__ __launch_bounds__(kSortBlock) void k_sort_count(const unsigned long long* __restrict__ keys, int n, int shift, int tiles, int* __restrict__ counts /*[digit][tile]*/)
{
    __shared__ int s_hist[kSortDigits];
    for (int d = (int)threadIdx.x; d < kSortDigits; d += kSortBlock)
        s_hist[d] = 0;
    __syncthreads();
    const int first = blockIdx.x*kSortTile + (int)threadIdx.x*kSortPerThread;
    for (int k = 0; k < kSortPerThread; ++k)
        if (first + k < n)
            atomicAdd(&s_hist[(int)((keys[first + k] >> shift) & 0xffu)], 1);
    __syncthreads();
    for (int d = (int)threadIdx.x; d < kSortDigits; d += kSortBlock)
        counts[(size_t)d*tiles + blockIdx.x] = s_hist[d];
}

__global__ __launch_bounds__(kSortBlock) void k_sort_scatter(const unsigned long long* __restrict__ keys, unsigned long long* __restrict__ out, int n, int shift, int tiles,
                                                             const int* __restrict__ bases /*[digit][tile], scanned*/)
{
    __shared__ uint16_t s_cnt[kSortBlock][kSortDigits];     // 64 KB
    uint32_t* const zero = reinterpret_cast<uint32_t*>(&s_cnt[0][0]);
    for (int i = (int)threadIdx.x; i < kSortBlock*kSortDigits/2; i += kSortBlock)
        zero[i] = 0u;
    __syncthreads();
    const int t = (int)threadIdx.x;
    const int first = blockIdx.x*kSortTile + t*kSortPerThread;
    unsigned long long key[kSortPerThread];
#pragma unroll
    for (int k = 0; k < kSortPerThread; ++k)
    {
        key[k] = first + k < n ? keys[first + k] : 0ull;
        if (first + k < n)
            s_cnt[t][(int)((key[k] >> shift) & 0xffu)] += 1;        // (this thread's own row)
    }
    __syncthreads();
    // digits d = t and t + 128: prefix of the counts over the threads, in thread order
    for (int d = t; d < kSortDigits; d += kSortBlock)
    {
        uint32_t run = 0;
        for (int u = 0; u < kSortBlock; ++u)
        {
            const uint32_t c = s_cnt[u][d];
            s_cnt[u][d] = (uint16_t)run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSortPerThread; ++k)
    {
        if (first + k < n)
        {
            const int d = (int)((key[k] >> shift) & 0xffu);
            const uint32_t r = s_cnt[t][d];
            s_cnt[t][d] = (uint16_t)(r + 1u);
            out[(size_t)bases[(size_t)d*tiles + blockIdx.x] + r] = key[k];
        }
    }
}

inline size_t sort_scratch_ints(size_t n)
{
    const size_t tiles = (n + kSortTile - 1)/kSortTile;
    return tiles*kSortDigits + scan_scratch_ints(tiles*kSortDigits);
}

// Sorts keys[0, n) by their bits [beginBit, endBit) (both multiples of 8) into `sorted`; keys with equal bits keep their order.  `keys` is
// used as the other ping-pong buffer (its contents are lost); an even number of passes is required so that the result lands in `sorted`
// ... or an odd one: the function copies where needed.  `scratch`: sort_scratch_ints(n) ints.  Launches only.
inline void radix_sort_keys(unsigned long long* keys, unsigned long long* sorted, size_t n, int beginBit, int endBit, int* scratch, hipStream_t st)
{
    if (n == 0)
        return;
    const int tiles = (int)((n + kSortTile - 1)/kSortTile);
    int* counts = scratch;
    int* scanTmp = scratch + (size_t)tiles*kSortDigits;
    unsigned long long* src = keys;
    unsigned long long* dst = sorted;
    for (int shift = beginBit; shift < endBit; shift += 8)
    {
        hipLaunchKernelGGL(k_sort_count, dim3((unsigned)tiles), dim3(kSortBlock), 0, st, (const unsigned long long*)src, (int)n, shift, tiles, counts);
        exclusive_scan(counts, counts, (size_t)tiles*kSortDigits, scanTmp, st);
        hipLaunchKernelGGL(k_sort_scatter, dim3((unsigned)tiles), dim3(kSortBlock), 0, st, (const unsigned long long*)src, dst, (int)n, shift, tiles, (const int*)counts);
        unsigned long long* t = src; src = dst; dst = t;
    }
    if (src != sorted)
        (void)hipMemcpyAsync(sorted, src, n*sizeof(unsigned long long), hipMemcpyDeviceToDevice, st);
}

} // namespace tn
