#!/usr/bin/env python3
"""Frames of an animation for the headless drivers' BATCH mode (main.cpp:104-118: a `%d` in the scene file name; :314-327: PNG per frame, renderer
deleted and re-created).  Runs only where /root/reference is mounted; the GPU box uses the committed files.

cornell.pack is loaded by the reference; per frame two primitives get new start / end transforms -- a sphere that travels across the box and
MOVES during the exposure (start != end: motion blur, data/motionblur.tin-style), and the light mesh drifting sideways (every shadow ray and
the light's sampling change) -- by oracle/ref_harness.cpp ref_scene_set_transform (mutate + the reference's own Scene::Build), and the frame is
written as a pack of its own:

  tests/golden/anim_cornell_<k>.pack   k = 0..3   (3 KB each: the scene as the reference would load frame k's .tin)
"""
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tests.oracle_api import RefOracle  # noqa: E402
from tinsel_amd import abi  # noqa: E402

FRAMES = 4
SPHERE, LIGHT = 6, 5           # primitives of cornell.pack: a sphere, the quad light mesh


def frame_moves(k):
    """[(primitive, translation of the start transform, of the end transform)] of frame k: the sphere's end of frame k is its start of frame k + 1"""
    x0, x1 = 0.12*k, 0.12*(k + 1)
    return [(SPHERE, (x0, 0.02*k, 0.0), (x1, 0.02*(k + 1), 0.0)), (LIGHT, (0.05*k, 0.0, -0.03*k), (0.05*k, 0.0, -0.03*k))]


def main():
    R = RefOracle()
    L = R.lib
    L.ref_scene_set_transform.argtypes = [C.c_void_p, C.c_int, C.POINTER(abi.Transform), C.POINTER(abi.Transform)]
    for k in range(FRAMES):
        h = R.load_pack(os.path.join(HERE, "cornell.pack"))
        for (i, ds, de) in frame_moves(k):
            p = R.primitive(h, i)
            s, e = abi.Transform.from_buffer_copy(bytes(p.start_transform)), abi.Transform.from_buffer_copy(bytes(p.end_transform))
            s.p.x += ds[0]; s.p.y += ds[1]; s.p.z += ds[2]
            e.p.x += de[0]; e.p.y += de[1]; e.p.z += de[2]
            assert L.ref_scene_set_transform(h, i, C.byref(s), C.byref(e)) == 0
        out = os.path.join(HERE, "anim_cornell_%d.pack" % k)
        n = R.write_pack(h, out)
        R.free(h)
        print("wrote %s (%d B)" % (out, n))


if __name__ == "__main__":
    main()
