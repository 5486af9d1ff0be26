// tn_launch.h -- the launches of the path kernels, as one function of a plain argument block.
//
// The library carries the path kernels TWICE, from two translation units built with different floating-point contracts:
//   tinsel_hip.hip   (namespace tn)        -ffp-contract=off, IEEE divide / sqrt, glibc's transcendentals restated:
//                                          bit-identical to the CPU oracle -- the default and the parity path;
//   tinsel_fast.hip  (namespace tn_fast)   FMA contraction, v_rcp / v_rsq / v_sqrt, hardware sin / cos / exp:
//                                          the opt-in tolerance arm (tinsel_hip_set_arithmetic), like the reference's
//                                          own `-O3 -ffast-math` / `-use_fast_math` builds (makefile:4, tinsel.vcxproj:134).
// Both include this header (the second one with `tn` renamed), so the two arms differ in arithmetic only: same kernels,
// same queues, same launch geometry.  The host fills a LaunchArgs per launch and hands it to launch_path_kernel of the
// arm in force.
#pragma once

#include "tn_kernels.h"

namespace tn {

enum PathKernel : int
{
    PK_GENERATE = 0, PK_EXTEND, PK_SHADE, PK_SHADOW, PK_BOUNCE, PK_MEGA, PK_WALK, PK_LIGHTS, PK_SWALK_EXTEND, PK_SWALK_SHADOW, PK_STEP,
};

struct LaunchArgs
{
    DevScene scene;
    PathState ps;
    SplitState ss;                  // the wavefront pipelines' dense state (PK_BOUNCE; PK_GENERATE .. PK_SHADE)
    QueueCtl ctl;
    CameraParams cam;
    FrameParams fp;
    const uint32_t* passSeeds;
    const float4* walkRec;          // k_walk's records (null: meshes are walked inline)
    uint32_t walkPrims;
    const uint32_t* order;          // region groups, longest first (k_region_order; null: in index order)
    BinPrims bins;                  // primitives whose leaf-box test sorts the queues
    WalkJob walk;                   // PK_WALK
    SwalkJob swalk;                 // PK_SWALK_*
    int swalkMode;                  // PK_SWALK_*: 0 = 256-thread workgroups, generic pointers; 1 / 2 = 1024-thread workgroups, arena in LDS (2: meshes in HBM too)
    int walkBig;                    // PK_WALK: 1 = 1024-thread workgroups with an LDS-resident tree top (2: two of them per CU, short LDS stacks), 0 = 256-thread ones
    int walkSingle;                 // PK_WALK: ONE walked primitive: k_walk<.., kWalkSingle>; else k_walk_rays (tn_walk.h)
    int shadeSorted;                // PK_SHADE: k_shade_sorted (paths taken class by class) instead of k_shade
    int lightsInExtend;             // PK_EXTEND, arena staged + meshes in HBM: the variant that draws the light samples too (A/B)
    int walkedOnly;                 // PK_EXTEND / PK_SHADOW / PK_STEP: every mesh of the scene is walked by k_walk (2: or is a quad tested in the scan) -> the lean scan variants
    int bounce;
    int bounceEnd;                  // PK_BOUNCE: the launch covers the bounces [bounce, bounceEnd)
    int stackEntries;
    int countDetail;                // detail counters on: the COUNT kernel variants
    int grid;
    uint32_t ldsBytes;              // dynamic LDS of the launch
};

inline void launch_path_kernel(int which, const LaunchArgs& a, hipStream_t st)
{
    const dim3 grid((unsigned)a.grid), block(kBlock);
    const bool lds = a.scene.allInArena != 0, count = a.countDetail != 0;
    // the arena is staged whole but some meshes live in HBM: SceneT's MIXED mode (scene records at compile-time LDS addresses)
    const bool mixed = !lds && !count && a.scene.arenaLdsBytes != 0 && a.scene.arenaLdsBytes == a.scene.arenaBytes;
    switch (which)
    {
    case PK_GENERATE:
        hipLaunchKernelGGL(k_generate, grid, block, 0, st, a.ss, a.ctl, a.cam, a.fp, a.passSeeds, a.scene.primBoxes, a.bins);
        break;
    case PK_EXTEND:
#define TN_LAUNCH2(KERNEL, ...)                                                                                        \
        do {                                                                                                           \
            if (count) { if (lds) hipLaunchKernelGGL((KERNEL<true, true>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<true, false>), __VA_ARGS__); } \
            else       { if (lds) hipLaunchKernelGGL((KERNEL<false, true>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<false, false>), __VA_ARGS__); } \
        } while (0)
        if (a.walkedOnly == 2 && mixed)
            hipLaunchKernelGGL((k_extend<false, true, 2, true>), grid, block, a.ldsBytes, st, a.scene, a.ss, a.ctl, a.bounce, a.stackEntries, a.walkRec, a.walkPrims, a.bins, a.order);
        else if (a.walkedOnly && mixed)
            hipLaunchKernelGGL((k_extend<false, true, 1, true>), grid, block, a.ldsBytes, st, a.scene, a.ss, a.ctl, a.bounce, a.stackEntries, a.walkRec, a.walkPrims, a.bins, a.order);
        else if (a.walkedOnly && !count && !lds)
            hipLaunchKernelGGL((k_extend<false, false, 1>), grid, block, a.ldsBytes, st, a.scene, a.ss, a.ctl, a.bounce, a.stackEntries, a.walkRec, a.walkPrims, a.bins, a.order);
        else if (mixed && a.lightsInExtend)
            hipLaunchKernelGGL((k_extend<false, true, 0, true, true>), grid, block, a.ldsBytes, st, a.scene, a.ss, a.ctl, a.bounce, a.stackEntries, a.walkRec, a.walkPrims, a.bins, a.order);
        else if (mixed)
            hipLaunchKernelGGL((k_extend<false, true, 0, true>), grid, block, a.ldsBytes, st, a.scene, a.ss, a.ctl, a.bounce, a.stackEntries, a.walkRec, a.walkPrims, a.bins, a.order);
        else
            TN_LAUNCH2(k_extend, grid, block, a.ldsBytes, st, a.scene, a.ss, a.ctl, a.bounce, a.stackEntries, a.walkRec, a.walkPrims, a.bins, a.order);
        break;
    case PK_SHADOW:
        if (a.walkedOnly == 2 && mixed)
            hipLaunchKernelGGL((k_shadow<false, true, 2, true>), grid, block, a.ldsBytes, st, a.scene, a.ss, a.ctl, a.bounce, a.stackEntries, a.walkRec, a.walkPrims, a.order);
        else if (a.walkedOnly && mixed)
            hipLaunchKernelGGL((k_shadow<false, true, 1, true>), grid, block, a.ldsBytes, st, a.scene, a.ss, a.ctl, a.bounce, a.stackEntries, a.walkRec, a.walkPrims, a.order);
        else if (a.walkedOnly && !count && !lds)
            hipLaunchKernelGGL((k_shadow<false, false, 1>), grid, block, a.ldsBytes, st, a.scene, a.ss, a.ctl, a.bounce, a.stackEntries, a.walkRec, a.walkPrims, a.order);
        else if (mixed)
            hipLaunchKernelGGL((k_shadow<false, true, 0, true>), grid, block, a.ldsBytes, st, a.scene, a.ss, a.ctl, a.bounce, a.stackEntries, a.walkRec, a.walkPrims, a.order);
        else
            TN_LAUNCH2(k_shadow, grid, block, a.ldsBytes, st, a.scene, a.ss, a.ctl, a.bounce, a.stackEntries, a.walkRec, a.walkPrims, a.order);
        break;
    case PK_MEGA:
        TN_LAUNCH2(k_mega, grid, block, a.ldsBytes, st, a.scene, a.ps, a.ctl, a.cam, a.fp, a.passSeeds, a.stackEntries);
        break;
#undef TN_LAUNCH2
    case PK_LIGHTS:
        if (mixed)
            hipLaunchKernelGGL((k_lights<true, true>), grid, block, a.ldsBytes, st, a.scene, a.ss, a.bounce, a.bins, a.order);
        else if (lds)
            hipLaunchKernelGGL((k_lights<true>), grid, block, a.ldsBytes, st, a.scene, a.ss, a.bounce, a.bins, a.order);
        else
            hipLaunchKernelGGL((k_lights<false>), grid, block, a.ldsBytes, st, a.scene, a.ss, a.bounce, a.bins, a.order);
        break;
    case PK_SHADE:
#define TN_LAUNCH_SHADE(KERNEL)                                                                                        \
        do {                                                                                                           \
            if (mixed) hipLaunchKernelGGL((KERNEL<true, true>), grid, block, a.ldsBytes, st, a.scene, a.ss, a.bounce, a.fp.maxDepth, a.fp.rrStart, a.bins, a.order); \
            else if (lds) hipLaunchKernelGGL((KERNEL<true>), grid, block, a.ldsBytes, st, a.scene, a.ss, a.bounce, a.fp.maxDepth, a.fp.rrStart, a.bins, a.order); \
            else hipLaunchKernelGGL((KERNEL<false>), grid, block, a.ldsBytes, st, a.scene, a.ss, a.bounce, a.fp.maxDepth, a.fp.rrStart, a.bins, a.order); \
        } while (0)
        if (a.shadeSorted) TN_LAUNCH_SHADE(k_shade_sorted); else TN_LAUNCH_SHADE(k_shade);
#undef TN_LAUNCH_SHADE
        break;
    case PK_BOUNCE:
#define TN_LAUNCH_BOUNCE(DEFER)                                                                                        \
        do {                                                                                                           \
            if (count) { if (lds) hipLaunchKernelGGL((k_bounce<true, true, false>), grid, block, a.ldsBytes, st, ka); \
                         else hipLaunchKernelGGL((k_bounce<true, false, false>), grid, block, a.ldsBytes, st, ka); } \
            else       { if (lds) hipLaunchKernelGGL((k_bounce<false, true, DEFER>), grid, block, a.ldsBytes, st, ka); \
                         else hipLaunchKernelGGL((k_bounce<false, false, DEFER>), grid, block, a.ldsBytes, st, ka); } \
        } while (0)
        {
            const BounceKernargs ka = { a.scene, a.ss, a.ctl, a.bounce, a.bounceEnd, a.stackEntries, a.cam, a.fp, a.passSeeds };
            // (the detail-counting variants walk the scene BVH: nothing to defer)
            if (a.scene.deferMeshes)
                TN_LAUNCH_BOUNCE(true);
            else
                TN_LAUNCH_BOUNCE(false);
        }
#undef TN_LAUNCH_BOUNCE
        break;
    case PK_STEP:
        // (LDS, WONLY, MIXED) as k_extend's variants: every mesh walked + arena staged / every mesh walked, generic pointers / arena staged, some
        // meshes inline / whole scene in LDS / generic pointers
#define TN_LAUNCH_STEP(L, W, M) hipLaunchKernelGGL((k_step<L, W, M>), grid, block, a.ldsBytes, st, a.scene, a.ss, a.ctl, a.bounce, a.fp.maxDepth, a.fp.rrStart, a.stackEntries, a.walkRec, a.walkPrims, a.bins, a.order)
        if (a.walkedOnly == 2 && mixed) TN_LAUNCH_STEP(true, 2, true);
        else if (a.walkedOnly && mixed) TN_LAUNCH_STEP(true, 1, true);
        else if (a.walkedOnly && !lds) TN_LAUNCH_STEP(false, 1, false);
        else if (mixed) TN_LAUNCH_STEP(true, 0, true);
        else if (lds) TN_LAUNCH_STEP(true, 0, false);
        else TN_LAUNCH_STEP(false, 0, false);
#undef TN_LAUNCH_STEP
        break;
    case PK_SWALK_EXTEND:
    case PK_SWALK_SHADOW:
#define TN_LAUNCH_SWALK(SH)                                                                                            \
        do {                                                                                                           \
            if (a.swalkMode == 1) hipLaunchKernelGGL((k_swalk<SH, 1024, 1>), grid, dim3(1024), a.ldsBytes, st, a.scene, a.ss, a.ctl, a.bounce, a.swalk); \
            else if (a.swalkMode == 2) hipLaunchKernelGGL((k_swalk<SH, 1024, 2>), grid, dim3(1024), a.ldsBytes, st, a.scene, a.ss, a.ctl, a.bounce, a.swalk); \
            else hipLaunchKernelGGL((k_swalk<SH, 256, 0>), grid, block, a.ldsBytes, st, a.scene, a.ss, a.ctl, a.bounce, a.swalk); \
        } while (0)
        if (which == PK_SWALK_SHADOW) TN_LAUNCH_SWALK(true); else TN_LAUNCH_SWALK(false);
#undef TN_LAUNCH_SWALK
        break;
    case PK_WALK:
        // ONE walked primitive: k_walk (its tree as kernel-argument scalars); several: k_walk_rays (a work item is a ray)
#define TN_LAUNCH_WALK(KERNEL, ...)                                                                                    \
        do {                                                                                                           \
            if (a.walkBig == 2) hipLaunchKernelGGL((KERNEL<1024, 8 __VA_ARGS__>), grid, dim3(1024), a.ldsBytes, st, a.scene, a.walk); \
            else if (a.walkBig) hipLaunchKernelGGL((KERNEL<1024, 4 __VA_ARGS__>), grid, dim3(1024), a.ldsBytes, st, a.scene, a.walk); \
            else hipLaunchKernelGGL((KERNEL<256, 5 __VA_ARGS__>), grid, dim3(256), a.ldsBytes, st, a.scene, a.walk);      \
        } while (0)
#define TN_COMMA ,
        if (a.walkSingle) TN_LAUNCH_WALK(k_walk, TN_COMMA kWalkSingle); else TN_LAUNCH_WALK(k_walk_rays, );
#undef TN_COMMA
#undef TN_LAUNCH_WALK
        break;
    default:
        break;
    }
}

// k_walk (tree tops), k_swalk, k_shade_sorted and k_bounce (shading pools beside a staged arena) ask for more dynamic LDS than the default
// launch limit allows: every variant's limit is raised ONCE, at create, and every result is checked -- a device that grants less fails there
// with the kernel's name instead of at some later launch with a generic error.
struct PrepReport
{
    int refused = 0;                // attributes the runtime refused
    const char* first = nullptr;    // the first kernel it refused
    int segPrefixLds = 0;           // the dynamic LDS k_seg_prefix may ask for (one count per region: the host clamps its grids to that)
};

inline void prep_lds(const void* kernel, const char* name, int bytes, PrepReport& rep)
{
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
    {
        (void)hipGetLastError();        // (a refused attribute must not surface as the next launch's error)
        if (!rep.first)
            rep.first = name;
        ++rep.refused;
    }
}

inline PrepReport prepare_path_kernels(int sharedMemLimit)
{
    PrepReport rep;
#define TN_PREP(...) prep_lds((const void*)__VA_ARGS__, #__VA_ARGS__, sharedMemLimit, rep)
    // (k_seg_prefix has static LDS too)
    {
        PrepReport seg;
        prep_lds((const void*)k_seg_prefix, "k_seg_prefix", sharedMemLimit - 1024, seg);
        rep.segPrefixLds = seg.refused ? (sharedMemLimit < 65536 ? sharedMemLimit : 65536) - 1024 : sharedMemLimit - 1024;
        if (seg.refused) { rep.refused += 1; rep.first = rep.first ? rep.first : "k_seg_prefix"; }
    }
    TN_PREP(k_walk_rays<1024, 4>); TN_PREP(k_walk_rays<1024, 8>); TN_PREP(k_walk_rays<256, 5>);
    TN_PREP(k_walk<1024, 4, kWalkSingle>); TN_PREP(k_walk<1024, 8, kWalkSingle>); TN_PREP(k_walk<256, 5, kWalkSingle>);
    TN_PREP(k_shade_sorted<true, true>); TN_PREP(k_shade_sorted<true>); TN_PREP(k_shade_sorted<false>);
    TN_PREP(k_swalk<false, 1024, 1>); TN_PREP(k_swalk<false, 1024, 2>); TN_PREP(k_swalk<false, 256, 0>);
    TN_PREP(k_swalk<true, 1024, 1>); TN_PREP(k_swalk<true, 1024, 2>); TN_PREP(k_swalk<true, 256, 0>);
    TN_PREP(k_bounce<true, true, false>); TN_PREP(k_bounce<true, false, false>);
    TN_PREP(k_bounce<false, true, false>); TN_PREP(k_bounce<false, false, false>); TN_PREP(k_bounce<false, true, true>); TN_PREP(k_bounce<false, false, true>);
#undef TN_PREP
    return rep;
}

} // namespace tn
