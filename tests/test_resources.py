"""Register / scratch budgets of the hot kernels, as the compiler reported them for THIS build (tinsel_amd/build.py build_verified ->
tinsel_amd/csrc/_obj/resources.json, written by __graft_entry__.build()).  The occupancy every measured number in DESIGN.md rests on is a
property of the build, and it has been lost silently before: until the end of round 4 the double-precision coefficients of the restated
sinf / cosf / expf sat in 24 VGPRs for the whole of every kernel that samples a BSDF, and 152 B of k_bounce's scratch were the constants
of a double atan2 no ray ever calls ("Where the time goes").  No GPU needed: the numbers come from hipcc."""
import hashlib
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "tinsel_amd", "csrc", "_obj")


def _resources():
    path = os.path.join(OBJ, "resources.json")
    obj = os.path.join(OBJ, "tinsel_hip.o")
    if not (os.path.exists(path) and os.path.exists(obj)):
        pytest.skip("no build record in this tree (python -c 'import __graft_entry__ as g; g.build()' writes it)")
    rec = json.load(open(path))
    if rec.get("object_sha256") != hashlib.sha256(open(obj, "rb").read()).hexdigest() or not rec.get("kernels"):
        pytest.skip("resources.json does not describe the object the library was linked from")
    return rec["kernels"]


def test_the_fused_kernel_runs_four_waves_per_simd():
    """k_bounce<COUNT, LDS, DEFER> at 128 VGPRs = four waves per SIMD (round 4: 162-168 and three) with at most a handful
    of loop invariants spilled in the prologue -- 20 B where the whole scene is staged into LDS (cornell's variant; 28 B in the variant with
    deferred mesh walks since the flat scan holds a primitive's record beside its box, round 6: veach 4K +2.7 % all the same).  What bought the
    registers: the kernel's wave-uniform bookkeeping in scalar registers (wave_in_block, tn_kernels.h), slot_pixel's reciprocals from the host."""
    k = _resources()
    variants = [n for n in k if n.startswith("k_bounce<0,")]            # (the <1,..> ones count detail statistics: not a timed path)
    assert len(variants) == 4
    for n in variants:
        lds = n.startswith("k_bounce<0,1,")
        assert k[n]["waves_per_simd"] == 4 and k[n]["vgprs"] <= 128 and k[n]["scratch_bytes"] <= (32 if lds else 80), (n, k[n])


def test_the_shading_kernel_runs_four_waves_per_simd():
    k = _resources()
    for n in ("k_shade<1,1>", "k_shade<1,0>"):                          # staged arena (+ meshes in HBM): glass, the 524k-triangle config
        assert k[n]["waves_per_simd"] == 4 and k[n]["vgprs"] <= 128 and k[n]["scratch_bytes"] == 0, (n, k[n])
    assert k["k_shade<0,0>"]["waves_per_simd"] == 4 and k["k_shade<0,0>"]["scratch_bytes"] <= 32
    for n in (m for m in k if m.startswith("k_shade_sorted<")):
        assert k[n]["waves_per_simd"] == 4 and k[n]["scratch_bytes"] <= 40, (n, k[n])
    assert not [m for m in k if m.startswith("k_shade<") and m.count(",") > 1]      # (the shadow-tracing variants are gone: they lost, round 4)


def test_the_paired_pipelines_kernel_runs_four_waves_without_scratch_where_it_is_the_default():
    """k_step<LDS, WONLY, MIXED> (tn_paired.h): AUTO picks the paired pipeline only where every mesh is walked by k_walk, i.e. the WONLY variants --
    those hold a path's shadow resolve, its closest hit and its shading in 128 VGPRs without a byte of scratch; the others (a mesh walked inline)
    may park a few registers."""
    k = _resources()
    for n in ("k_step<1,1,1>", "k_step<0,1,0>", "k_step<1,2,1>"):          # (<1,2,1>: ... or is a quad tested in the scan, glass.tin when asked for)
        assert k[n]["waves_per_simd"] == 4 and k[n]["vgprs"] <= 128 and k[n]["scratch_bytes"] == 0, (n, k[n])
    for n in ("k_step<1,0,1>", "k_step<1,0,0>", "k_step<0,0,0>"):
        assert k[n]["waves_per_simd"] == 4 and k[n]["scratch_bytes"] <= 64, (n, k[n])


def test_the_lean_scan_kernels_with_quads_keep_their_waves():
    """k_extend / k_shadow<.., WONLY = 2, ..> (every mesh walked by k_walk or a quad tested in the scan by ray_mesh_two_leaves -- glass.tin):
    no register spilled (the shadow kernel's 20 B are the general variants' unused frame), and the shadow kernel still at seven waves per SIMD"""
    k = _resources()
    assert k["k_shadow<0,1,2,1>"]["waves_per_simd"] >= 7 and k["k_shadow<0,1,2,1>"]["vgpr_spills"] == 0 and k["k_shadow<0,1,2,1>"]["scratch_bytes"] <= 20, k["k_shadow<0,1,2,1>"]
    assert k["k_extend<0,1,2,1,1>"]["waves_per_simd"] >= 5 and k["k_extend<0,1,2,1,1>"]["scratch_bytes"] == 0, k["k_extend<0,1,2,1,1>"]


def test_the_library_carries_no_foreign_kernels():
    """405 kernel instantiations, most of them rocprim trampolines for other architectures.  The sort and the scan of the BVH
    builder are the library's own now (tn_sort.h): every kernel in the object is one of tn::k_*."""
    k = _resources()
    assert len(k) <= 250, len(k)
    assert all(n.startswith("k_") for n in k), [n for n in k if not n.startswith("k_")][:3]
    for n in ("k_sort_count", "k_sort_scatter", "k_scan_tiles", "k_scan_sums", "k_scan_add", "k_lbvh_leaves", "k_lbvh_keys"):
        assert n in k, n


def test_the_walk_kernels_keep_eight_waves_per_simd():
    k = _resources()
    for n in ("k_walk<1024,8,2>", "k_walk_rays<1024,8>"):               # one walked primitive / several (the defaults)
        assert k[n]["waves_per_simd"] == 8 and k[n]["vgprs"] <= 64 and k[n]["scratch_bytes"] == 0, (n, k[n])
    for n in (m for m in k if m.startswith("k_swalk<")):
        assert k[n]["scratch_bytes"] == 0 and k[n]["waves_per_simd"] >= 5, (n, k[n])


def test_the_scan_kernels_spill_nothing():
    k = _resources()
    for n in (m for m in k if m.startswith(("k_extend<0,", "k_shadow<0,", "k_lights<", "k_generate", "k_accumulate"))):
        # (k_shadow with inline mesh walks parks 20 B of plane equations; k_extend's staged-arena variant with light sampling trades 36 B
        # for its fifth wave per SIMD: glass, profiles/r04_w_ab_extend5.md)
        assert k[n]["scratch_bytes"] <= (40 if n == "k_extend<0,1,0,1,1>" else 20), (n, k[n])
    for n in ("k_accumulate_piped<3>", "k_accumulate_piped<4>"):       # ten waves a tile: three tiles a CU need 7.5 waves per SIMD
        assert k[n]["waves_per_simd"] == 8 and k[n]["scratch_bytes"] == 0, (n, k[n])
    assert k["k_extend<0,1,0,1,1>"]["waves_per_simd"] >= 5
    assert k["k_extend<0,1,1,1,1>"]["waves_per_simd"] >= 6              # lean scan + light sampling (the 524k-triangle config)
    assert k["k_shadow<0,1,1,1>"]["waves_per_simd"] >= 7


def test_the_build_parses_the_compiler_remarks():
    """build.py's reading of -Rpass-analysis=kernel-resource-usage (no compiler needed: canned text)"""
    import sys
    sys.path.insert(0, ROOT)
    from tinsel_amd import build as hb
    assert hb._kernel_name("_ZN2tn8k_bounceILb0ELb1ELb0EEEvNS_8DevSceneENS_10SplitStateE") == "k_bounce<0,1,0>"
    assert hb._kernel_name("_ZN2tn6k_walkILi1024ELi4ELi6EEEvNS_8DevSceneENS_7WalkJobE") == "k_walk<1024,4,6>"
    assert hb._kernel_name("_ZN2tn13k_lbvh_leavesEPKNS_5Tri48EPKyiPfPi") == "k_lbvh_leaves"           # (a 'v' inside the name: ADVICE r04)
    assert hb._kernel_name("_ZN12_GLOBAL__N_112k_sum_accumsENS_10SumSourcesE") == "_ZN12_GLOBAL__N_112k_sum_accumsENS_10SumSourcesE"
    assert hb._kernel_name("_ZN2tn6k_walkILi1024ELi8ELi2EEEvNS_7WalkJobE") == "k_walk<1024,8,2>"
    assert hb._kernel_name("_ZN2tn10k_generateENS_10SplitStateENS_8QueueCtlE") == "k_generate"
    text = """
a.h:33:1: remark: Function Name: _ZN2tn7k_shadeILb1ELb1ELb0ELb0EEEvNS_8DevSceneE [-Rpass-analysis=kernel-resource-usage]
a.h:33:1: remark:     TotalSGPRs: 106 [-Rpass-analysis=kernel-resource-usage]
a.h:33:1: remark:     VGPRs: 128 [-Rpass-analysis=kernel-resource-usage]
a.h:33:1: remark:     ScratchSize [bytes/lane]: 0 [-Rpass-analysis=kernel-resource-usage]
a.h:33:1: remark:     Occupancy [waves/SIMD]: 4 [-Rpass-analysis=kernel-resource-usage]
a.h:33:1: remark:     SGPRs Spill: 52 [-Rpass-analysis=kernel-resource-usage]
"""
    got = hb._parse_resources(text)
    assert got == {"k_shade<1,1,0,0>": {"sgprs": 106, "vgprs": 128, "scratch_bytes": 0, "waves_per_simd": 4, "sgpr_spills": 52}}
