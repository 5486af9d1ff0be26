"""tinsel_hip_refit_mesh: new vertex positions for a mesh whose topology did not change (deforming / animated meshes;
the reference re-runs its host SAH build for that, mesh.cpp:314-338, and Scene::Build, scene.cpp:4-16).  The trees keep their
shape and every box is recomputed bottom-up: the mesh's own tree on the device, and at the SCENE level the leaf box of every
instance (PrimitiveBounds, intersection.h:906-939) and its ancestors in the scene BVH.

Oracle: the SAME displaced mesh inside a scene pack whose reference BVH nodes were refitted here on the host with numpy
(leaf box = min/max of the triangle's vertices, node box = union of the children's: what Bounds::AddPoint / the builder
store -- min and max do not round), whose area CDF follows Mesh::RebuildCDF's serial order and whose SCENE BVH was refitted
too (leaf box = the oracle's PrimitiveBounds of the refitted mesh -- pinned to the reference's in tests/test_oracle.py --
ancestors = union of their children), rendered by the C oracle on the CPU.  Same trees, same boxes, same triangles => the
GPU must be bit-identical to it.  The displacements leave the original bounds: with stale scene-level boxes the mesh would
be clipped (asserted: the oracle itself renders another image with the stale boxes)."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from tinsel_amd import abi
from tests import oracle_api as oa
from tests.test_gpu_parity import _load

pytestmark = pytest.mark.gpu


def _mesh_views(blob, prim_index):
    """numpy views INTO the (mutable) pack blob of one mesh primitive: positions, indices, nodes, cdf, + the byte offset of .area"""
    off_prims = struct.unpack_from("<Q", blob, 32)[0]
    base = off_prims + prim_index*272
    assert struct.unpack_from("<i", blob, base + 64)[0] == abi.GEOM_MESH
    g = base + 72
    o_pos, o_nrm, o_idx, o_nodes, o_cdf = struct.unpack_from("<5Q", blob, g)
    nv, ni, nn = struct.unpack_from("<3i", blob, g + 40)
    pos = np.frombuffer(blob, np.float32, nv*3, o_pos).reshape(nv, 3)
    idx = np.frombuffer(blob, np.int32, ni, o_idx).reshape(-1, 3)
    nodes = np.frombuffer(blob, np.dtype([("lo", "<f4", 3), ("hi", "<f4", 3), ("left", "<u4"), ("right", "<u4")]), nn, o_nodes)
    cdf = np.frombuffer(blob, np.float32, ni//3, o_cdf)
    return pos, idx, nodes, cdf, g + 52


NODE = np.dtype([("lo", "<f4", 3), ("hi", "<f4", 3), ("left", "<u4"), ("right", "<u4")])


def _refit_scene_level(blob, prim_index):
    """Scene BVH of the pack, in place: the leaf box of the primitive from the oracle's PrimitiveBounds, ancestors = unions."""
    num_nodes = struct.unpack_from("<I", blob, 16)[0]
    off_nodes = struct.unpack_from("<Q", blob, 40)[0]
    nodes = np.frombuffer(blob, NODE, num_nodes, off_nodes)
    P = oa.PortOracle()
    h = P.load_pack(bytes(blob))
    lo, hi = P.primitive_bounds(h, prim_index)
    P.free(h)
    leaf = [k for k in range(num_nodes) if nodes["right"][k] >> 31 and nodes["left"][k] == prim_index]
    assert len(leaf) == 1
    nodes["lo"][leaf[0]], nodes["hi"][leaf[0]] = lo, hi
    order, stack = [], [0]
    while stack:
        k = stack.pop()
        order.append(k)
        if not nodes["right"][k] >> 31:
            stack.append(int(nodes["left"][k])); stack.append(int(nodes["right"][k] & 0x7fffffff))
    for k in reversed(order):
        if not nodes["right"][k] >> 31:
            l, r = int(nodes["left"][k]), int(nodes["right"][k] & 0x7fffffff)
            nodes["lo"][k] = np.minimum(nodes["lo"][l], nodes["lo"][r])
            nodes["hi"][k] = np.maximum(nodes["hi"][l], nodes["hi"][r])


def _refit_pack(blob, prim_index, new_pos, scene_level=True):
    """Host refit of the reference's own 32-B nodes + CDF/area (+ the scene BVH), in place."""
    pos, idx, nodes, cdf, off_area = _mesh_views(blob, prim_index)
    pos[:] = new_pos
    tri_lo = pos[idx].min(axis=1)
    tri_hi = pos[idx].max(axis=1)
    order, stack = [], [0]
    while stack:                                    # post-order over the reference tree
        k = stack.pop()
        order.append(k)
        if not nodes["right"][k] >> 31:
            stack.append(int(nodes["left"][k])); stack.append(int(nodes["right"][k] & 0x7fffffff))
    for k in reversed(order):
        if nodes["right"][k] >> 31:
            t = int(nodes["left"][k])
            nodes["lo"][k], nodes["hi"][k] = tri_lo[t], tri_hi[t]
        else:
            l, r = int(nodes["left"][k]), int(nodes["right"][k] & 0x7fffffff)
            nodes["lo"][k] = np.minimum(nodes["lo"][l], nodes["lo"][r])
            nodes["hi"][k] = np.maximum(nodes["hi"][l], nodes["hi"][r])
    # Mesh::RebuildCDF (mesh.cpp:340-368): serial fp32
    a, b, c = pos[idx[:, 0]], pos[idx[:, 1]], pos[idx[:, 2]]
    ab, ac = (b - a).astype(np.float32), (c - a).astype(np.float32)
    cr = np.stack([ab[:, 1]*ac[:, 2] - ac[:, 1]*ab[:, 2], ab[:, 2]*ac[:, 0] - ab[:, 0]*ac[:, 2], ab[:, 0]*ac[:, 1] - ab[:, 1]*ac[:, 0]], axis=1).astype(np.float32)
    ln = np.sqrt(((cr[:, 0]*cr[:, 0]).astype(np.float32) + (cr[:, 1]*cr[:, 1]).astype(np.float32)).astype(np.float32) + (cr[:, 2]*cr[:, 2]).astype(np.float32)).astype(np.float32)
    areas = (np.float32(0.5)*ln).astype(np.float32)
    total = np.float32(0.0)
    run = np.empty(len(areas), np.float32)
    for t, v in enumerate(areas):
        total = np.float32(total + v)
        run[t] = total
    cdf[:] = (run/total).astype(np.float32)
    struct.pack_into("<f", blob, off_area, float(total))
    if scene_level:
        _refit_scene_level(blob, prim_index)


def _displace(pos, amount):
    p = pos.astype(np.float32).copy()
    p[:, 1] += (amount*np.sin(7.0*p[:, 0].astype(np.float64) + 3.0*p[:, 2].astype(np.float64))).astype(np.float32)
    p[:, 0] *= np.float32(1.0 + 4.0*amount)          # 20 % wider: well outside the original leaf box
    return p


def _mesh_prim(scene):
    prims = C.cast(scene.desc.primitives, C.POINTER(abi.Primitive))
    return [i for i in range(scene.desc.num_primitives) if prims[i].type == abi.GEOM_MESH][0]


@pytest.mark.parametrize("bvh", [abi.BVH_REFERENCE, abi.BVH_LBVH], ids=["reference-tree", "device-built-tree"])
def test_refit_after_displacement_matches_the_oracle_on_the_refitted_pack(bvh):
    import tinsel_amd
    name = "ajax_standin_96"                                        # 18,432 triangles, in HBM, walked by k_walk
    scene, cam, opt, g = _load(name)
    passes = 2
    blob = bytearray(open(os.path.join(oa.GOLDEN, name + ".pack"), "rb").read())
    prim = _mesh_prim(scene)
    old_pos = _mesh_views(blob, prim)[0].copy()
    new_pos = _displace(old_pos, 0.05)
    _refit_pack(blob, prim, new_pos)

    P = oa.PortOracle()
    h = P.load_pack(bytes(blob))
    ref_accum, ref_rad, _ = P.render_seeded(h, cam, opt, 0, passes, want_radiance=True)
    P.free(h)
    assert not np.array_equal(ref_rad, g["radiance"][:passes])      # the displacement is visible
    # ... and it leaves the original bounds: with the scene BVH as loaded (stale leaf box) the oracle itself clips the mesh
    stale = bytearray(open(os.path.join(oa.GOLDEN, name + ".pack"), "rb").read())
    _refit_pack(stale, prim, new_pos, scene_level=False)
    h = P.load_pack(bytes(stale))
    _, stale_rad, _ = P.render_seeded(h, cam, opt, 0, passes, want_radiance=True)
    P.free(h)
    assert not np.array_equal(stale_rad, ref_rad)

    r = tinsel_amd.create_gpu_renderer(scene)
    if bvh == abi.BVH_LBVH:
        r.set_mesh_bvh(abi.BVH_LBVH)
    r.refit_mesh(prim, new_pos)
    r.set_pipeline(abi.PIPELINE_WAVEFRONT_SPLIT)
    r.init(opt.width, opt.height)
    out = r.render(cam, opt, passes=passes)
    rad = r.batch_radiance(passes, opt.height, opt.width)
    same = float((rad == ref_rad).all(axis=-1).mean())
    if bvh == abi.BVH_REFERENCE:
        assert np.array_equal(rad, ref_rad), "%.4f %% of the paths identical" % (100*same)
        assert np.array_equal(out, ref_accum)
    else:
        assert same >= 0.999 and oa.image_l2(out, ref_accum) <= 1e-3    # another tree shape: exact-t ties may resolve differently
        # back to the reference's tree: it was refitted too
        r.set_mesh_bvh(abi.BVH_REFERENCE)
        r.init(opt.width, opt.height); r.set_pass_index(0)
        assert np.array_equal(r.render(cam, opt, passes=passes), ref_accum)
    # refit back to the original vertices restores the original image bit for bit
    r.refit_mesh(prim, old_pos)
    r.init(opt.width, opt.height); r.set_pass_index(0)
    assert np.array_equal(r.render(cam, opt, passes=int(g["passes"])), g["accum"])
    r.close()


def test_refit_of_a_light_mesh_updates_cdf_and_area(monkeypatch):
    """cornell's quad light as a mesh in HBM (tinsel_hip_tuning::small_mesh_bytes = 0): moving its vertices changes the light's area
    (PrimitiveArea -> the light pdf) and its sampling CDF; the oracle renders the refitted pack."""
    import tinsel_amd
    from tinsel_amd import renderer
    monkeypatch.setattr(renderer, "DEFAULT_TUNING", abi.Tuning(small_mesh_bytes=0))
    scene, cam, opt, g = _load("cornell")
    blob = bytearray(open(os.path.join(oa.GOLDEN, "cornell.pack"), "rb").read())
    prim = _mesh_prim(scene)
    pos = _mesh_views(blob, prim)[0].copy()
    new_pos = pos.copy()
    new_pos[:, 0] *= np.float32(1.6)                 # a wider light (it leaves its old box)
    new_pos[0, 2] += np.float32(0.05)                # and no longer a parallelogram: the two triangles differ in area
    _refit_pack(blob, prim, new_pos)
    P = oa.PortOracle()
    h = P.load_pack(bytes(blob))
    ref_accum, ref_rad, _ = P.render_seeded(h, cam, opt, 0, 2, want_radiance=True)
    P.free(h)
    r = tinsel_amd.create_gpu_renderer(scene)
    r.refit_mesh(prim, new_pos)
    r.init(opt.width, opt.height)
    out = r.render(cam, opt, passes=2)
    rad = r.batch_radiance(2, opt.height, opt.width)
    r.close()
    assert np.array_equal(rad, ref_rad), "%d paths differ" % int((rad != ref_rad).any(axis=-1).sum())
    assert np.array_equal(out, ref_accum)


def test_refit_of_a_mesh_in_the_lds_arena():
    """cornell as shipped: the quad light rides in the LDS-staged scene arena (fused kernel, the two-leaf walk).  The refit
    rewrites its records in the arena's copy in HBM, which every launch stages from."""
    import tinsel_amd
    scene, cam, opt, g = _load("cornell")
    blob = bytearray(open(os.path.join(oa.GOLDEN, "cornell.pack"), "rb").read())
    prim = _mesh_prim(scene)
    pos = _mesh_views(blob, prim)[0].copy()
    new_pos = pos.copy()
    new_pos[:, 0] *= np.float32(1.7)                 # a wider light: leaves its old box
    new_pos[0, 2] += np.float32(0.05)
    _refit_pack(blob, prim, new_pos)
    P = oa.PortOracle()
    h = P.load_pack(bytes(blob))
    ref_accum, ref_rad, _ = P.render_seeded(h, cam, opt, 0, 2, want_radiance=True)
    P.free(h)
    r = tinsel_amd.create_gpu_renderer(scene)
    r.refit_mesh(prim, new_pos)
    r.init(opt.width, opt.height)
    out = r.render(cam, opt, passes=2)
    rad = r.batch_radiance(2, opt.height, opt.width)
    assert np.array_equal(rad, ref_rad), "%d paths differ" % int((rad != ref_rad).any(axis=-1).sum())
    assert np.array_equal(out, ref_accum)
    r.refit_mesh(prim, pos)                          # and back
    r.init(opt.width, opt.height); r.set_pass_index(0)
    assert np.array_equal(r.render(cam, opt, passes=int(g["passes"])), g["accum"])
    r.close()


def test_refit_refuses_what_it_cannot_do():
    import tinsel_amd
    scene, cam, opt, g = _load("cornell")
    r = tinsel_amd.create_gpu_renderer(scene)
    prim = _mesh_prim(scene)
    with pytest.raises(tinsel_amd.TinselHipError):
        r.refit_mesh(0 if prim != 0 else 1, np.zeros((4, 3), np.float32))      # not a mesh
    r.close()
    scene, cam, opt, g = _load("ajax_standin_96")
    r = tinsel_amd.create_gpu_renderer(scene)
    with pytest.raises(tinsel_amd.TinselHipError, match="topology"):
        r.refit_mesh(_mesh_prim(scene), np.zeros((5, 3), np.float32))
    r.close()
