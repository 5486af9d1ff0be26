#!/usr/bin/env python3
"""Regenerates tests/golden/ from the reference itself.  Runs ONLY where
/root/reference is mounted (this container); the GPU box uses the committed files.

For every scene:
  <name>.pack                scene pack written by the reference's own loader + Scene::Build
                             (oracle/ref_harness.cpp: ref_scene_write_pack)
  <name>.golden.npz          outputs of the reference's PathTrace (render.cpp:230) under the
                             per-path seed contract, at a small size:
                               radiance [passes,H,W,3]  per-path radiance
                               accum    [H,W,4]         AddSample'd framebuffer (render.cpp:401-445)
                               normals  [H,W,4]         CpuRenderer eNormals mode (render.cpp:494-515)
                             + the camera/options used (as raw bytes).
Usage:  python tests/golden/make_golden.py [--ref /root/reference]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tests.oracle_api import RefOracle  # noqa: E402
from tinsel_amd import abi  # noqa: E402

# name -> (tin path relative to reference data dir or absolute, golden W, H, passes, maxDepth override or None)
SCENES = {
    "cornell": ("data/cornell.tin", 64, 64, 4, None),
    "veach": ("data/veach.tin", 64, 64, 3, None),
    "glass": ("data/glass.tin", 64, 64, 3, 12),
    "simple": ("data/simple.tin", 64, 32, 3, None),
    "conservation": ("data/conservation.tin", 64, 32, 3, None),
    "furnace": ("data/furnace.tin", 32, 32, 2, 32),
    "emitter": ("data/emitter.tin", 48, 48, 2, None),
    "gloss": ("data/gloss.tin", 64, 64, 3, None),
    # the reference's own motion-blur scene: the octopus mesh (10,214 triangles, in HBM: walked by k_walk) turns half a revolution about y
    # while the shutter is open (motionblur.tin:16-17, 68), a sphere light with 10 samples
    "motionblur": ("data/motionblur.tin", 64, 64, 3, None),
    "features": (os.path.join(HERE, "scenes", "features.tin"), 96, 64, 4, None),
    # 203 primitives: scene-level BVH walk instead of the flat scan, arena too large for LDS (HBM-resident scene)
    "many_spheres": (os.path.join(HERE, "scenes", "many_spheres.tin"), 128, 96, 3, None),
    # one primitive (root of the scene BVH is a leaf), 7x5 frame, maxDepth 1, emissive light hit directly
    "one_sphere": (os.path.join(HERE, "scenes", "one_sphere.tin"), 7, 5, 5, None),
}


def struct_bytes(s):
    return np.frombuffer(bytes(s), dtype=np.uint8).copy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("scenes", nargs="*", help="regenerate only these .tin-derived fixtures")
    args = ap.parse_args()

    R = RefOracle()
    only = [a for a in sys.argv[1:] if not a.startswith("-") and a in SCENES]
    for name, (tin, W, H, passes, depth) in SCENES.items():
        if only and name not in only:
            continue
        path = tin if os.path.isabs(tin) else os.path.join(args.ref, tin)
        h = R.load_tin(path)
        if name == "features":
            pass
        pack = os.path.join(HERE, name + ".pack")
        R.write_pack(h, pack)
        make_outputs(R, h, name, W, H, passes, depth)
        R.free(h)

    if only:
        return

    # features + procedural probe: exercises ProbeSample/ProbePdf MIS (render.cpp:107-144,370-380)
    h = R.load_tin(os.path.join(HERE, "scenes", "features.tin"))
    R.lib.ref_scene_set_procedural_probe(h, 64, 32)
    R.write_pack(h, os.path.join(HERE, "features_probe.pack"))
    make_outputs(R, h, "features_probe", 96, 64, 4, None)

    # data/ajax.tin with a deterministic 18,432-triangle stand-in for the missing ajax.obj (see make_large.py for
    # the 524,288-triangle config): a real SAH mesh BVH (deep stack), gloss material, sphere light
    import subprocess
    subprocess.run([sys.executable, os.path.join(HERE, "make_large.py"), "96", "96", args.ref], check=True)
    os.replace(os.path.join(HERE, "large", "ajax_standin_96.pack"), os.path.join(HERE, "ajax_standin_96.pack"))
    h = R.load_pack(os.path.join(HERE, "ajax_standin_96.pack"))
    make_outputs(R, h, "ajax_standin_96", 96, 96, 3, None)
    R.free(h)

    # cornell + probe (open box is lit by the probe through the camera side)
    h = R.load_tin(os.path.join(args.ref, "data/cornell.tin"))
    R.lib.ref_scene_set_procedural_probe(h, 64, 32)
    R.write_pack(h, os.path.join(HERE, "cornell_probe.pack"))
    make_outputs(R, h, "cornell_probe", 64, 64, 3, None)


def make_outputs(R, h, name, W, H, passes, depth):
    cam, opt = R.camera_options(h)
    opt.width, opt.height = W, H
    if depth is not None:
        opt.max_depth = depth
    accum, rad, _ = R.render_seeded(h, cam, opt, 0, passes, want_accum=True, want_radiance=True, threads=8)
    nopt = opt.copy()
    nopt.mode = abi.MODE_NORMALS
    normals, _ = R.render_faithful(h, cam, nopt, 1)
    np.savez_compressed(os.path.join(HERE, name + ".golden.npz"), radiance=rad, accum=accum, normals=normals,
                        camera=struct_bytes(cam), options=struct_bytes(opt), passes=np.int32(passes))
    finite = np.isfinite(rad).all()
    print("%-16s %3dx%-3d passes=%d depth=%d mean radiance %s finite=%s" % (
        name, W, H, passes, opt.max_depth, rad.mean(axis=(0, 1, 2)), finite))


if __name__ == "__main__":
    main()
