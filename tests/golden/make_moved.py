#!/usr/bin/env python3
"""Fixtures for tinsel_hip_set_primitive_transform / tinsel_hip_rebuild_scene: primitives MOVE, the reference re-runs its own
Scene::Build (scene.cpp:4-16) and renders.  Runs only where /root/reference is mounted; the GPU box uses the committed file.

For every scene in MOVES the committed <name>.pack is loaded by the reference, the listed primitives get new start / end transforms
(oracle/ref_harness.cpp: ref_scene_set_transform = mutate + Scene::Build), and

  moved.golden.npz   per scene:  <name>_index [K], <name>_start / <name>_end [K,32] (the Transform structs as bytes),
                     <name>_nodes [(2P-1)*32] (the reference's rebuilt scene BVH as bytes), <name>_radiance / <name>_accum (the
                     reference's PathTrace + AddSample on the moved scene), <name>_camera / <name>_options / <name>_passes.

The GPU test applies the same transforms to a renderer created from the ORIGINAL pack and rebuilds the scene level."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tests.oracle_api import RefOracle  # noqa: E402
from tinsel_amd import abi  # noqa: E402

# name -> (W, H, passes, [(primitive, translate start xyz, translate end xyz, scale factor)])
#   many_spheres  203 primitives: the scene BVH is WALKED (k_swalk), so the rebuilt tree is what rays traverse; two spheres leave their old
#                 leaf boxes by several radii, one of them becomes a MOVING primitive (start != end: motion blur), one mesh-free scene
#   features      flat scan (the tree decides ties only): a mesh instance moves and grows, a sphere starts to move during the exposure, the
#                 mesh LIGHT moves and shrinks (PrimitiveArea, hence every light pdf, follows)
#   cornell       the light mesh moves down into the box: every shadow ray changes
MOVES = {
    "many_spheres": (128, 96, 3, [(7, (1.5, 0.6, -1.0), (1.5, 0.6, -1.0), 1.0), (40, (-0.8, 0.3, 0.5), (-0.2, 0.5, 0.5), 1.0), (150, (0.0, 1.0, 0.0), (0.0, 1.0, 0.0), 1.5)]),
    "features": (96, 64, 4, [(6, (0.4, 0.1, -0.3), (0.4, 0.1, -0.3), 1.25), (4, (-0.5, 0.2, 0.2), (-0.3, 0.2, 0.2), 1.0), (2, (0.2, -0.3, 0.0), (0.2, -0.3, 0.0), 0.8)]),
    "cornell": (64, 64, 4, [(5, (0.2, -0.5, 0.1), (0.2, -0.5, 0.1), 1.0), (6, (0.3, 0.0, 0.2), (0.3, 0.0, 0.2), 1.0)]),
}


def struct_bytes(s):
    return np.frombuffer(bytes(s), dtype=np.uint8).copy()


def main():
    R = RefOracle()
    L = R.lib
    L.ref_scene_set_transform.argtypes = [C.c_void_p, C.c_int, C.POINTER(abi.Transform), C.POINTER(abi.Transform)]
    L.ref_scene_get_bvh.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    out = {}
    for name, (W, H, passes, moves) in MOVES.items():
        h = R.load_pack(os.path.join(HERE, name + ".pack"))
        idx, starts, ends = [], [], []
        for (i, ds, de, scale) in moves:
            p = R.primitive(h, i)
            s, e = abi.Transform.from_buffer_copy(bytes(p.start_transform)), abi.Transform.from_buffer_copy(bytes(p.end_transform))
            s.p.x += ds[0]; s.p.y += ds[1]; s.p.z += ds[2]; s.s *= scale
            e.p.x += de[0]; e.p.y += de[1]; e.p.z += de[2]; e.s *= scale
            assert L.ref_scene_set_transform(h, i, C.byref(s), C.byref(e)) == 0
            idx.append(i); starts.append(struct_bytes(s)); ends.append(struct_bytes(e))
        n = L.ref_scene_get_bvh(h, None, 0)
        nodes = (abi.BVHNode*n)()
        L.ref_scene_get_bvh(h, C.cast(nodes, C.c_void_p), n)
        g = np.load(os.path.join(HERE, name + ".golden.npz"))
        cam = abi.Camera.from_buffer_copy(g["camera"].tobytes())
        opt = abi.Options.from_buffer_copy(g["options"].tobytes())
        opt.width, opt.height = W, H
        accum, rad, _ = R.render_seeded(h, cam, opt, 0, passes, want_accum=True, want_radiance=True, threads=8)
        R.free(h)
        changed = float((rad != g["radiance"]).any(axis=-1).mean()) if rad.shape == g["radiance"].shape else float("nan")
        print("%-14s %d primitives moved, %d nodes, %.1f %% of the paths differ from the unmoved scene" % (name, len(idx), n, 100*changed))
        out[name + "_index"] = np.array(idx, np.int32)
        out[name + "_start"] = np.stack(starts); out[name + "_end"] = np.stack(ends)
        out[name + "_nodes"] = np.frombuffer(bytes(nodes), dtype=np.uint8).copy()
        out[name + "_radiance"] = rad; out[name + "_accum"] = accum
        out[name + "_camera"] = struct_bytes(cam); out[name + "_options"] = struct_bytes(opt); out[name + "_passes"] = np.int32(passes)
    np.savez_compressed(os.path.join(HERE, "moved.golden.npz"), **out)


if __name__ == "__main__":
    main()
