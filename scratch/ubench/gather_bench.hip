// gather_bench.hip -- how fast can a CU chase random 64-B records (BVH nodes)?  Dependent chains, one per lane.
//   A  per lane: 4 x global_load_dwordx4 of its own record (what ray_mesh / k_walk v1 do)
//   B  quad-cooperative: in step q every lane of a quad loads its 16-B piece of the record of quad-lane q (4 loads, each a
//      64-B contiguous access per quad); NO redistribution (each lane sums what it got) -- isolates the address-rate effect
//   C  B through LDS-DMA (global_load_lds_dwordx4: wave-linear LDS image) + 4 x ds_read_b128 of the lane's own record
//   D  B + 4x4 quad transpose in VALU (DPP)
// build: hipcc --offload-arch=gfx950 -O3 -o gather_bench gather_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include <cstring>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f4* GlobalF4;
typedef __attribute__((address_space(3))) void* LdsPtr;

__device__ __forceinline__ float fsum(f4 a) { return a.x + a.y + a.z + a.w; }

template <int MODE>
__global__ __launch_bounds__(256, 4) void k_chase(const float4* __restrict__ recsIn, uint32_t nrec, int steps, const uint32_t* __restrict__ start, float* __restrict__ out, int padLds)
{
    extern __shared__ f4 s_lds[];     // MODE C: [wave][4][64] float4 = 4 KB per wave (+ padLds bytes to bound occupancy)
    GlobalF4 recs = (GlobalF4)(uintptr_t)recsIn;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t idx = start[blockIdx.x*256 + threadIdx.x];
    float acc = 0.0f;
    for (int s = 0; s < steps; ++s)
    {
        f4 a, b, c, d;
        if (MODE == 0)
        {
            GlobalF4 p = recs + (size_t)idx*4;
            a = p[0]; b = p[1]; c = p[2]; d = p[3];
        }
        else if (MODE == 1 || MODE == 3)
        {
            const int piece = lane & 3;
            const uint32_t i0 = __builtin_amdgcn_mov_dpp(idx, 0x00, 0xf, 0xf, true);   // quad_perm [0,0,0,0]
            const uint32_t i1 = __builtin_amdgcn_mov_dpp(idx, 0x55, 0xf, 0xf, true);   // [1,1,1,1]
            const uint32_t i2 = __builtin_amdgcn_mov_dpp(idx, 0xaa, 0xf, 0xf, true);   // [2,2,2,2]
            const uint32_t i3 = __builtin_amdgcn_mov_dpp(idx, 0xff, 0xf, 0xf, true);   // [3,3,3,3]
            f4 r0 = recs[(size_t)i0*4 + piece];
            f4 r1 = recs[(size_t)i1*4 + piece];
            f4 r2 = recs[(size_t)i2*4 + piece];
            f4 r3 = recs[(size_t)i3*4 + piece];
            if (MODE == 1)
            {
                // no redistribution: lane q needs "its" record's next index: piece 3 of record q lives in lane 3 reg q
                a = r0; b = r1; c = r2; d = r3;
                const float n0 = __builtin_amdgcn_mov_dpp(r0.x, 0xff, 0xf, 0xf, true);
                const float n1 = __builtin_amdgcn_mov_dpp(r1.x, 0xff, 0xf, 0xf, true);
                const float n2 = __builtin_amdgcn_mov_dpp(r2.x, 0xff, 0xf, 0xf, true);
                const float n3 = __builtin_amdgcn_mov_dpp(r3.x, 0xff, 0xf, 0xf, true);
                d.x = piece == 0 ? n0 : piece == 1 ? n1 : piece == 2 ? n2 : n3;
            }
            else
            {
                // 4x4 transpose inside the quad: T[j]@lane i = R[i]@lane j, two butterfly stages per dword
                f4& R0 = r0; f4& R1 = r1; f4& R2 = r2; f4& R3 = r3;
                const bool odd = lane & 1, hi = lane & 2;
#pragma unroll
                for (int w = 0; w < 4; ++w)
                {
                    // stage 1: lanes i^1, registers (0,1) and (2,3)
                    float t = odd ? R0[w] : R1[w];
                    t = __builtin_amdgcn_mov_dpp(t, 0xb1, 0xf, 0xf, true);      // [1,0,3,2]
                    if (odd) R0[w] = t; else R1[w] = t;
                    t = odd ? R2[w] : R3[w];
                    t = __builtin_amdgcn_mov_dpp(t, 0xb1, 0xf, 0xf, true);
                    if (odd) R2[w] = t; else R3[w] = t;
                    // stage 2: lanes i^2, registers (0,2) and (1,3)
                    t = hi ? R0[w] : R2[w];
                    t = __builtin_amdgcn_mov_dpp(t, 0x4e, 0xf, 0xf, true);      // [2,3,0,1]
                    if (hi) R0[w] = t; else R2[w] = t;
                    t = hi ? R1[w] : R3[w];
                    t = __builtin_amdgcn_mov_dpp(t, 0x4e, 0xf, 0xf, true);
                    if (hi) R1[w] = t; else R3[w] = t;
                }
                a = r0; b = r1; c = r2; d = r3;
            }
        }
        else
        {
            const int piece = lane & 3;
            const uint32_t i0 = __builtin_amdgcn_mov_dpp(idx, 0x00, 0xf, 0xf, true);
            const uint32_t i1 = __builtin_amdgcn_mov_dpp(idx, 0x55, 0xf, 0xf, true);
            const uint32_t i2 = __builtin_amdgcn_mov_dpp(idx, 0xaa, 0xf, 0xf, true);
            const uint32_t i3 = __builtin_amdgcn_mov_dpp(idx, 0xff, 0xf, 0xf, true);
            f4* base = s_lds + wave*256;                   // wave-uniform
            __builtin_amdgcn_global_load_lds(recs + (size_t)i0*4 + piece, (LdsPtr)(base + 0), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(recs + (size_t)i1*4 + piece, (LdsPtr)(base + 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(recs + (size_t)i2*4 + piece, (LdsPtr)(base + 128), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(recs + (size_t)i3*4 + piece, (LdsPtr)(base + 192), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // record of quad-lane q of quad k landed at region q, float4 index 4k .. 4k+3
            const f4* mine = base + (lane & 3)*64 + (lane >> 2)*4;
            a = mine[0]; b = mine[1]; c = mine[2]; d = mine[3];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // before the next step's DMA overwrites the image
        }
        acc += fsum(a) + fsum(b) + fsum(c) + d.y;
        idx = __float_as_uint(d.x);
    }
    out[blockIdx.x*256 + threadIdx.x] = acc + (float)idx;
}

int main(int argc, char** argv)
{
    const uint32_t nrec = argc > 1 ? (uint32_t)atoi(argv[1]) : 524288u;
    const int parts = argc > 3 ? atoi(argv[3]) : 1;      // > 1: block b chases inside part (b % parts) of the records only
    const int steps = 256;
    const int blocksPerCU = argc > 2 ? atoi(argv[2]) : 4;
    std::vector<float> recs((size_t)nrec*16);
    std::vector<uint32_t> perm(nrec);
    for (uint32_t i = 0; i < nrec; ++i) perm[i] = i;
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    const uint32_t partLen = nrec/(uint32_t)parts;
    for (int p = 0; p < parts; ++p)
        for (uint32_t i = partLen - 1; i > 0; --i) { uint32_t j = (uint32_t)(rnd() % (i + 1)); std::swap(perm[p*partLen + i], perm[p*partLen + j]); }
    for (uint32_t i = 0; i < nrec; ++i)
    {
        for (int k = 0; k < 16; ++k) recs[(size_t)i*16 + k] = 1e-3f*(float)((i + k) & 1023);
        uint32_t nxt = perm[i];
        memcpy(&recs[(size_t)i*16 + 12], &nxt, 4);
    }
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount*blocksPerCU*4;
    std::vector<uint32_t> start((size_t)grid*256);
    for (size_t t = 0; t < start.size(); ++t) { const uint32_t blk = (uint32_t)(t/256); start[t] = (blk % parts)*partLen + (uint32_t)(rnd() % partLen); }
    float4* dRecs; uint32_t* dStart; float* dOut;
    CHECK(hipMalloc(&dRecs, recs.size()*4)); CHECK(hipMalloc(&dStart, start.size()*4)); CHECK(hipMalloc(&dOut, start.size()*4));
    CHECK(hipMemcpy(dRecs, recs.data(), recs.size()*4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dStart, start.data(), start.size()*4, hipMemcpyHostToDevice));
    const size_t lds = 160*1024/blocksPerCU - 1024;       // bounds residency to blocksPerCU blocks per CU
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* names[4] = { "A per-lane 4 x dwordx4", "B quad-cooperative, no redistribution", "C quad-cooperative via LDS-DMA + ds_read_b128", "D quad-cooperative + DPP transpose" };
    std::vector<float> ref, got(start.size());
    for (int mode = 0; mode < (parts > 1 ? 1 : 4); ++mode)
    {
        for (int rep = 0; rep < 2; ++rep)
        {
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_chase<0>, dim3(grid), dim3(256), lds, 0, dRecs, nrec, steps, dStart, dOut, 0);
            if (mode == 1) hipLaunchKernelGGL(k_chase<1>, dim3(grid), dim3(256), lds, 0, dRecs, nrec, steps, dStart, dOut, 0);
            if (mode == 2) hipLaunchKernelGGL(k_chase<2>, dim3(grid), dim3(256), lds, 0, dRecs, nrec, steps, dStart, dOut, 0);
            if (mode == 3) hipLaunchKernelGGL(k_chase<3>, dim3(grid), dim3(256), lds, 0, dRecs, nrec, steps, dStart, dOut, 0);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipGetLastError());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 1)
            {
                CHECK(hipMemcpy(got.data(), dOut, got.size()*4, hipMemcpyDeviceToHost));
                size_t bad = 0;
                if (mode == 0 || mode == 1) { if (mode == 0) ref = got; }
                else for (size_t i = 0; i < got.size(); ++i) bad += got[i] != ref[i];
                const double visits = (double)grid*256*steps;
                printf("%-52s parts %d nrec %8u  %2d blocks/CU  %8.3f ms  %7.1f G visits/s  %6.2f TB/s of 64-B records  mismatches %zu\n",
                       names[mode], parts, nrec, blocksPerCU, ms, visits/ms*1e-6, visits*64/ms*1e-9, bad);
            }
        }
    }
    return 0;
}
