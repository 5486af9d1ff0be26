"""CPU tests of the drop-in boundary (run with -m "not gpu"): the C-ABI library loads, exports every
symbol include/tinsel_hip.h declares, mirrors the reference PODs byte for byte, opens scene packs on
the host, and FAILS LOUDLY (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import tinsel_amd
from tinsel_amd import abi
from tinsel_amd.renderer import EXPORTED_SYMBOLS, LIB_PATH

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tinsel_hip.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tinsel_(?:hip|pack|image)_\w+)\s*\(", text)))


def test_library_is_built_in_tree():
    assert os.path.exists(LIB_PATH), "run `python -m tinsel_amd.build` (hipcc --offload-arch=gfx950)"
    assert os.path.dirname(LIB_PATH) == os.path.join(ROOT, "tinsel_amd")


def test_every_declared_symbol_is_exported():
    L = tinsel_amd.load_library()
    declared = _declared()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), "include/tinsel_hip.h declares %s but the library does not export it" % name
    # and the python mirror binds exactly the declared set
    assert sorted(EXPORTED_SYMBOLS) == declared


def test_library_contains_gfx950_code_object():
    blob = open(LIB_PATH, "rb").read()
    assert b"gfx950" in blob, "no gfx950 code object embedded"


def test_pod_mirrors_match_reference_layouts():
    # sizes / offsets probed from the reference with g++
    assert C.sizeof(abi.Vec3) == 12 and C.sizeof(abi.Vec4) == 16
    assert C.sizeof(abi.Transform) == 32 and C.sizeof(abi.BVHNode) == 32
    assert C.sizeof(abi.Material) == 128 and abi.Material.eta.offset == 36 and abi.Material.transmission.offset == 80
    assert abi.Material.bump_map.offset == 88 and abi.Material.bump.offset == 112 and abi.Material.bump_tile.offset == 116
    assert C.sizeof(abi.MeshGeometry) == 64 and abi.MeshGeometry.num_vertices.offset == 40 and abi.MeshGeometry.id.offset == 56
    assert C.sizeof(abi.Primitive) == 272 and abi.Primitive.type.offset == 64 and abi.Primitive.geo.offset == 72
    assert abi.Primitive.material.offset == 136 and abi.Primitive.light_samples.offset == 264
    assert C.sizeof(abi.Camera) == 40 and C.sizeof(abi.Filter) == 16 and C.sizeof(abi.Options) == 48
    assert abi.Options.filter.offset == 12 and abi.Options.exposure.offset == 28 and abi.Options.max_depth.offset == 40


@pytest.mark.parametrize("name,nprims,nmesh", [("cornell", 8, 1), ("veach", 9, 1), ("glass", 9, 3), ("features_probe", 9, 2)])
def test_pack_open_on_host(golden_dir, name, nprims, nmesh):
    scene = tinsel_amd.Scene.load_pack(os.path.join(golden_dir, name + ".pack"))
    d = scene.desc
    assert d.num_primitives == nprims and d.num_bvh_nodes == 2*nprims - 1
    prims = C.cast(d.primitives, C.POINTER(abi.Primitive))
    ids = set()
    for i in range(nprims):
        p = prims[i]
        assert p.type in (abi.GEOM_SPHERE, abi.GEOM_PLANE, abi.GEOM_MESH)
        if p.type == abi.GEOM_MESH:
            g = p.geo.mesh
            ids.add(g.id)
            assert g.num_indices % 3 == 0 and g.num_nodes == 2*(g.num_indices//3) - 1
            # pointers were relocated into the blob and are readable
            idx = np.ctypeslib.as_array(C.cast(g.indices, C.POINTER(C.c_int32)), (g.num_indices,))
            assert idx.min() >= 0 and idx.max() < g.num_vertices
            cdf = np.ctypeslib.as_array(C.cast(g.cdf, C.POINTER(C.c_float)), (g.num_indices//3,))
            assert np.all(np.diff(cdf) >= 0) and abs(cdf[-1] - 1.0) < 1e-5
    assert len(ids) == nmesh
    assert bool(d.probe_valid) == name.endswith("probe")
    assert scene.options.width > 0 and scene.options.max_depth > 0


def test_pack_open_rejects_garbage():
    with pytest.raises(tinsel_amd.TinselHipError):
        tinsel_amd.Scene(b"\0" * 512)
    good = open(os.path.join(ROOT, "tests", "golden", "cornell.pack"), "rb").read()
    with pytest.raises(tinsel_amd.TinselHipError):
        tinsel_amd.Scene(good[:300])


def test_pack_open_rejects_offsets_and_counts_that_wrap(golden_dir):
    """A hostile pack must not get pointers outside the blob: offsets near 2^64 (off + bytes wraps), negative counts."""
    import struct
    good = bytearray(open(os.path.join(golden_dir, "cornell.pack"), "rb").read())
    # header: magic 8, version 4, num_primitives 4, num_bvh_nodes 4, num_meshes 4, total_bytes 8, off_primitives 8, off_bvh_nodes 8
    bad = bytearray(good)
    struct.pack_into("<Q", bad, 32, 0xFFFFFFFFFFFFFF00)            # off_primitives: off + bytes wraps past zero
    with pytest.raises(tinsel_amd.TinselHipError):
        tinsel_amd.Scene(bytes(bad))
    bad = bytearray(good)
    struct.pack_into("<I", bad, 12, 0xFFFFFFF0)                    # num_primitives * 272 overflows 32 bits, not 64: out of range
    with pytest.raises(tinsel_amd.TinselHipError):
        tinsel_amd.Scene(bytes(bad))
    # a mesh primitive with negative counts / an offset that wraps
    off_prims = struct.unpack_from("<Q", good, 32)[0]
    nprims = struct.unpack_from("<I", good, 12)[0]
    mesh_at = None
    for i in range(nprims):
        base = off_prims + i*272
        if struct.unpack_from("<i", good, base + 64)[0] == abi.GEOM_MESH:
            mesh_at = base + 72
            break
    assert mesh_at is not None
    bad = bytearray(good)
    struct.pack_into("<i", bad, mesh_at + 40, -5)                  # num_vertices
    with pytest.raises(tinsel_amd.TinselHipError):
        tinsel_amd.Scene(bytes(bad))
    bad = bytearray(good)
    struct.pack_into("<Q", bad, mesh_at + 0, 0xFFFFFFFFFFFFFFF8)   # positions offset
    with pytest.raises(tinsel_amd.TinselHipError):
        tinsel_amd.Scene(bytes(bad))
    tinsel_amd.Scene(bytes(good))                                  # the untouched pack still opens


def test_no_cpu_fallback(golden_dir):
    """Without a visible GPU the constructor must raise; nothing renders on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    scene = tinsel_amd.Scene.load_pack(os.path.join(golden_dir, "cornell.pack"))
    with pytest.raises(tinsel_amd.TinselHipError, match="no HIP device|no CPU fallback"):
        tinsel_amd.create_gpu_renderer(scene)


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "tinsel_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_api" not in text and "libtinsel_oracle" not in text and "libtinsel_ref" not in text, f
                assert not re.search(r'#include\s+"[^"]*oracle', text), f


def test_region_cut_invariants_over_the_whole_range_of_batch_sizes():
    """tinsel_hip_plan_regions (host arithmetic of streaming_grid + cut_regions, no device): however a batch is cut -- uniform regions,
    short regions at the end of a large batch, the equal-share cut of a batch that one resident set takes whole -- the regions cover
    every slot, are whole waves long, come in whole workgroups of four, fit the arrays sized for them, and the short ones are the last."""
    import random
    import tinsel_amd
    rng = random.Random(7)
    sizes = [1, 2, 63, 64, 65, 255, 256, 257, 4095, 4096, 4097, 65536, 256*256*16, 512*512*4, 1024*1024, 1920*1080, 1920*1080*20,
             1024*1024*20, 1024*1024*64, 3840*2160*8, (64 << 20) - 1, 64 << 20, (64 << 20) + 1, 200_000_000, 4_000_000_000]
    sizes += [rng.randrange(1, 1 << rng.randrange(1, 32)) for _ in range(1500)]
    for cus in (256, 304, 64, 8):
        for fused in (True, False):
            for nee in (1, 4):
                for slots in sizes:
                    p = tinsel_amd.plan_regions(slots, cus, nee, fused)
                    n, L, big, S, grid, cap = (p[k] for k in ("num_regions", "region_len", "big_regions", "short_len", "grid", "max_regions"))
                    what = "%d slots, %d CUs, fused %s: %s" % (slots, cus, fused, p)
                    assert n >= 4 and n % 4 == 0 and big % 4 == 0 and 0 < big <= n, what
                    assert L >= 64 and L % 64 == 0 and S >= 64 and S % 64 == 0 and S <= L, what
                    assert grid*4 == n and n <= cap, what
                    covered = big*L + (n - big)*S
                    assert covered >= slots, what
                    assert covered <= slots + cap*64, what          # the arrays' padding: a wave per region
                    assert covered < (1 << 32), what                # positions are 32-bit
                    if not fused:
                        assert big == n, what                       # the split pipeline keeps all regions alike
                    if n > big:                                     # short regions exist: the long ones alone do not cover the batch
                        assert big*L < slots, what


def test_division_by_host_reciprocal_is_exact():
    """tn_kernels.h div_magic (slot_pixel: a path slot -> pass, pixel): n / d as mulhi(n, floor((2^32 - 1) / d)) plus ONE correction, for every
    n < 2^32 and d >= 1.  The host puts the reciprocals into FrameParams; the kernels never divide.  Checked here on the edges and on two
    million random pairs (numpy: the same 32-bit arithmetic)."""
    import numpy as np
    rng = np.random.default_rng(5)
    d = np.concatenate([np.array([1, 2, 3, 7, 64, 4096, 1920, 1080, 1920*1080, 3840*2160, 1 << 20, (1 << 32) - 1], np.uint64),
                        rng.integers(1, 1 << 32, 200000, dtype=np.uint64), rng.integers(1, 1 << 16, 200000, dtype=np.uint64)])
    for n in (np.zeros_like(d), d - np.uint64(1), d, np.minimum(d*np.uint64(3) + np.uint64(1), np.uint64((1 << 32) - 1)), np.full_like(d, (1 << 32) - 1),
              rng.integers(0, 1 << 32, d.size, dtype=np.uint64)):
        m = np.uint64((1 << 32) - 1)//d
        q = (n*m) >> np.uint64(32)
        rem = n - q*d
        fix = rem >= d
        q, rem = q + fix.astype(np.uint64), rem - d*fix.astype(np.uint64)
        assert np.array_equal(q, n//d) and np.array_equal(rem, n % d)
