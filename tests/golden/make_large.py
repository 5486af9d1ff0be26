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


if __name__ == "__main__":
    a = sys.argv[1:]
    main(*(int(x) for x in a[:2]), *(a[2:3]))
