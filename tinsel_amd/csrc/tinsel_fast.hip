// tinsel_fast.hip -- the path kernels a second time, under the TOLERANCE arithmetic contract (tinsel_hip_set_arithmetic,
// TINSEL_ARITH_FAST).  Built by tinsel_amd/build.py with
//     -DTN_FAST=1 -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -freciprocal-math -fgpu-flush-denormals-to-zero
// (never -ffinite-math-only: the traversal relies on 1/0 = inf and FLT_MAX sentinels exactly like the reference does).
// What changes against the default arm: FMA contraction everywhere, v_rcp_f32 / v_rsq_f32 / v_sqrt_f32 where the reference
// divides, takes square roots and normalises, the hardware's v_sin / v_cos / v_exp instead of the restated glibc double
// evaluations, hardware min/max in every slab test.  What does not: kernels, queues, launch geometry, RNG streams -- so the
// same seeds give the same paths up to branch flips, and the image stays inside the stated 1e-3 per-pixel L2 of the CPU
// reference (tests/test_gpu_fast.py measures it; bench.py reports fast_l2 beside fast_msamples_s).
// The reference itself ships this trade: `-O3 -ffast-math` (makefile:4), `-use_fast_math -prec-div=false -prec-sqrt=false`
// (tinsel.vcxproj:134).
#ifndef TN_FAST
#error "build with -DTN_FAST=1 (tinsel_amd/build.py)"
#endif
#define tn tn_fast
#include "tn_launch.h"

extern "C" void tinsel_fast_launch_path_kernel(int which, const void* launchArgs, void* stream)
{
    tn_fast::launch_path_kernel(which, *static_cast<const tn_fast::LaunchArgs*>(launchArgs), (hipStream_t)stream);
}

// returns the number of kernels whose dynamic-LDS limit the runtime refused to raise; *first: the first one's name
extern "C" int tinsel_fast_prepare_path_kernels(int sharedMemLimit, const char** first)
{
    const tn_fast::PrepReport rep = tn_fast::prepare_path_kernels(sharedMemLimit);
    if (first)
        *first = rep.first;
    return rep.refused;
}

extern "C" unsigned tinsel_fast_launch_args_size(void) { return (unsigned)sizeof(tn_fast::LaunchArgs); }
