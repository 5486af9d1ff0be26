#!/usr/bin/env python3
"""Builds tests/golden/large/ajax_standin.pack (git-ignored, ~75 MB): data/ajax.tin with the MISSING
meshes/ajax.obj (reference .MISSING_LARGE_BLOBS) replaced by a deterministic 524,288-triangle stand-in
(CreateSphere(512,512) + closed-form ripple, then the reference's ImportMesh post-processing and its own
SAH BVHBuilder) -- BASELINE.json configs[2] / SURVEY.md 0.1.  Needs /root/reference."""
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.oracle_api import RefOracle  # noqa: E402
from tinsel_amd import abi  # noqa: E402


def main(slices=512, segments=512, ref="/root/reference"):
    R = RefOracle()
    h = R.load_tin(os.path.join(ref, "data/ajax.tin"))        # loads sphere light + plane; the mesh primitive is dropped
    gloss = abi.Material()                                      # `material gloss` of data/ajax.tin:22-28
    gloss.color = abi.Vec3(0.95, 0.9, 0.9)
    gloss.specular, gloss.roughness, gloss.metallic = 1.0, 0.025, 0.0
    gloss.clearcoat_gloss = 1.0
    gloss.bump_tile = abi.Vec3(10.0, 10.0, 10.0)
    R.lib.ref_scene_add_standin_mesh(h, slices, segments, C.c_float(2.0), C.byref(gloss), 1)
    out = os.path.join(HERE, "large", "ajax_standin.pack" if slices == 512 else "ajax_standin_%d.pack" % slices)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    n = R.write_pack(h, out)
    print("wrote %s (%.1f MB)" % (out, n/1e6))


def aphrodite(subdivisions=1, ref="/root/reference"):
    """tests/golden/large/ajax_aphrodite.pack (git-ignored, ~40 MB): data/ajax.tin with meshes/ajax.obj replaced by the largest mesh the
    reference ships, data/meshes/Aphrodite_from_jotero_com.obj (a scan: 106,846 triangles), 1 -> 4 midpoint-subdivided to 427,384 -- SURVEY.md
    0.1's other option for BASELINE.json configs[2].  Import, post-processing and the SAH tree are the reference's own (oracle/ref_harness.cpp
    ref_scene_add_obj_mesh): an IRREGULAR tree, where the stand-in above is a regular tessellation.  subdivisions = 0: the mesh as shipped
    (ajax_aphrodite_0.pack, 10 MB)."""
    R = RefOracle()
    R.lib.ref_scene_add_obj_mesh.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int]
    h = R.load_tin(os.path.join(ref, "data/ajax.tin"))
    gloss = abi.Material()
    gloss.color = abi.Vec3(0.95, 0.9, 0.9)
    gloss.specular, gloss.roughness, gloss.metallic = 1.0, 0.025, 0.0
    gloss.clearcoat_gloss = 1.0
    gloss.bump_tile = abi.Vec3(10.0, 10.0, 10.0)
    # (the scan's up axis is z: the mesh data's axes are shifted by two places, z -> y, so that the primitive keeps ajax.tin's identity rotation)
    tris = R.lib.ref_scene_add_obj_mesh(h, os.path.join(ref, "data/meshes/Aphrodite_from_jotero_com.obj").encode(), subdivisions, 2, C.c_float(2.0), C.byref(gloss), 1)
    assert tris > 0, "ImportMesh failed"
    out = os.path.join(HERE, "large", "ajax_aphrodite.pack" if subdivisions == 1 else "ajax_aphrodite_%d.pack" % subdivisions)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    n = R.write_pack(h, out)
    print("wrote %s (%.1f MB, %d triangles)" % (out, n/1e6, tris))


if __name__ == "__main__":
    a = sys.argv[1:]
    if a[:1] == ["aphrodite"]:
        aphrodite(*(int(x) for x in a[1:2]))
    else:
        main(*(int(x) for x in a[:2]), *(a[2:3]))
