// valu_bench.hip -- wave64 VALU issue rate on gfx950: cycles per instruction for the instruction classes the path
// tracer is made of (plain f32 add/mul/fma, min/max, compare+select, packed f32).  8 independent chains per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k_valu(float* out, int iters, float seed)
{
    float a[8]; f2 p[8];
    for (int k = 0; k < 8; ++k) { a[k] = seed + threadIdx.x*1e-3f + k; p[k] = f2{ a[k], a[k] + 0.5f }; }
    const float m = 1.0000001f, c = 1e-7f;
    const f2 pm = { m, m }, pc = { c, c };
    for (int i = 0; i < iters; ++i)
    {
#pragma unroll
        for (int r = 0; r < 8; ++r)
        {
#pragma unroll
            for (int k = 0; k < 8; ++k)
            {
                if (MODE == 0) a[k] = __builtin_fmaf(a[k], m, c);                              // v_fma_f32
                if (MODE == 1) a[k] = a[k]*m;                                                  // v_mul_f32
                if (MODE == 2) a[k] = __builtin_fminf(a[k] + c, 1e30f);                        // v_add + v_min
                if (MODE == 3) a[k] = (a[k] < 1e30f) ? a[k] + c : m;                           // v_cmp + v_cndmask (+ add)
                if (MODE == 4) p[k] = __builtin_elementwise_fma(p[k], pm, pc);                 // v_pk_fma_f32
                if (MODE == 5) p[k] = p[k]*pm;                                                 // v_pk_mul_f32
            }
        }
    }
    float s = 0;
    for (int k = 0; k < 8; ++k) s += a[k] + p[k].x + p[k].y;
    out[blockIdx.x*256 + threadIdx.x] = s;
}

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount*8;        // 8 blocks x 4 waves = 32 waves per CU = 8 per SIMD
    float* d; CHECK(hipMalloc(&d, (size_t)grid*256*4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 2000;
    const char* names[6] = { "v_fma_f32", "v_mul_f32", "v_add_f32 + v_min_f32", "v_cmp + v_cndmask + v_add", "v_pk_fma_f32", "v_pk_mul_f32" };
    const int instPer[6] = { 1, 1, 2, 3, 1, 1 };
    for (int mode = 0; mode < 6; ++mode)
        for (int rep = 0; rep < 2; ++rep)
        {
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_valu<0>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f);
            if (mode == 1) hipLaunchKernelGGL(k_valu<1>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f);
            if (mode == 2) hipLaunchKernelGGL(k_valu<2>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f);
            if (mode == 3) hipLaunchKernelGGL(k_valu<3>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f);
            if (mode == 4) hipLaunchKernelGGL(k_valu<4>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f);
            if (mode == 5) hipLaunchKernelGGL(k_valu<5>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep)
            {
                const double winst = (double)grid*4*iters*64*instPer[mode];          // wave-instructions
                const double perSimdPerSec = winst/(ms*1e-3)/(prop.multiProcessorCount*4);
                printf("%-28s %8.3f ms  %6.2f G wave-inst/s per SIMD  => %.2f cycles per wave64 instruction at 2.4 GHz (%d CUs, clock %d MHz)\n",
                       names[mode], ms, perSimdPerSec*1e-9, 2.4e9/perSimdPerSec, prop.multiProcessorCount, prop.clockRate/1000);
            }
        }
    return 0;
}
