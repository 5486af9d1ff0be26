#!/usr/bin/env python3
"""Random-scene parity corpus: tests/golden/fuzz.golden.npz.

32 small random scenes (seeded generator below: 3-12 primitives of every type -- planes, spheres, inline meshes, all
with random poses, a third of them moving -- random Disney materials over the whole parameter space incl. transmission,
absorption, subsurface, clearcoat and metals, 1-3 emitters with 1-3 light samples each, random sky, camera, depth,
filter and clamp) are loaded by the reference's own loader + Scene::Build and rendered by the reference's PathTrace
under the per-path seed contract.  The file keeps, per scene, the scene pack (as bytes) and the per-path radiance and
framebuffer -- so the GPU box needs neither the reference nor the .tin files.  Needs /root/reference.

Usage:  python tests/golden/make_fuzz.py            (deterministic: the same file every time)"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.oracle_api import RefOracle  # noqa: E402

NUM_SCENES = 32

MESHES = """
mesh quad
{
	verts 4
	-0.5 0 0.5
	0.5 0 0.5
	0.5 0 -0.5
	-0.5 0 -0.5

	tris 2
	0 2 1
	0 3 2
}

mesh tetra
{
	verts 4
	-1.0 0.0 -0.7
	1.0 0.0 -0.7
	0.0 0.0 1.0
	0.0 1.4 0.0

	tris 4
	0 2 1
	0 1 3
	1 2 3
	2 0 3
}

mesh wedge
{
	verts 6
	-0.5 0 -0.5
	0.5 0 -0.5
	0.5 0 0.5
	-0.5 0 0.5
	-0.5 0.8 -0.5
	0.5 0.8 -0.5

	tris 8
	0 1 2
	0 2 3
	0 4 5
	0 5 1
	3 2 5
	3 5 4
	0 3 4
	1 5 2
}
"""


def quat(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    return q


def f3(v):
    return "%.6g %.6g %.6g" % tuple(float(x) for x in v)


def f4(v):
    return "%.6g %.6g %.6g %.6g" % tuple(float(x) for x in v)


def material(rng, name, emitter):
    lines = ["material %s" % name, "{"]
    if emitter:
        lines.append("\temission %s" % f3(rng.uniform(2.0, 25.0, 3)))
        lines.append("\tcolor %s" % f3(rng.uniform(0.0, 0.6, 3)*(rng.random() < 0.5)))
    else:
        lines.append("\tcolor %s" % f3(rng.uniform(0.05, 0.95, 3)))
    kind = rng.integers(0, 6)
    lines.append("\troughness %.6g" % float(rng.choice([0.001, 0.02, 0.1, 0.3, 0.6, 1.0]) if rng.random() < 0.5 else rng.uniform(0.0, 1.0)))
    lines.append("\tspecular %.6g" % float(rng.uniform(0.0, 1.0)))
    lines.append("\tmetallic %.6g" % float(rng.choice([0.0, 1.0, rng.uniform(0, 1)])))
    if kind == 1:
        lines.append("\tsubsurface %.6g" % float(rng.uniform(0.1, 1.0)))
    if kind == 2:
        lines.append("\tclearcoat %.6g" % float(rng.uniform(0.1, 1.0)))
        lines.append("\tclearcoatGloss %.6g" % float(rng.uniform(0.0, 1.0)))
    if kind in (3, 4):
        lines.append("\ttransmission %.6g" % float(rng.choice([1.0, 0.9, rng.uniform(0.2, 1.0)])))
        if rng.random() < 0.7:
            lines.append("\teta %.6g" % float(rng.uniform(1.05, 2.2)))
        if kind == 4:
            lines.append("\tabsorption %s" % f3(rng.uniform(0.0, 3.0, 3)))
    if rng.random() < 0.2:
        lines.append("\tspecularTint %.6g" % float(rng.uniform(0, 1)))
        lines.append("\tsheen %.6g" % float(rng.uniform(0, 1)))
    lines.append("}")
    return "\n".join(lines)


def primitive(rng, kind, mat, light_samples):
    lines = ["primitive", "{", "\ttype %s" % ("mesh" if kind in ("quad", "tetra", "wedge") else kind)]
    if kind == "plane":
        n = rng.normal(size=3)
        if rng.random() < 0.6:
            n = np.eye(3)[rng.integers(0, 3)]*rng.choice([-1.0, 1.0])
        n = n/np.linalg.norm(n)
        lines.append("\tplane %s %.6g" % (f3(n), float(rng.uniform(0.5, 3.0))))
    else:
        p = rng.uniform(-2.0, 2.0, 3)
        if rng.random() < 0.33:
            lines.append("\tposition %s , %s" % (f3(p), f3(p + rng.uniform(-0.5, 0.5, 3))))
        else:
            lines.append("\tposition %s" % f3(p))
        if rng.random() < 0.6:
            q = quat(rng)
            if rng.random() < 0.3:
                lines.append("\trotation %s , %s" % (f4(q), f4(quat(rng))))
            else:
                lines.append("\trotation %s" % f4(q))
        s = float(rng.uniform(0.4, 1.8))
        if rng.random() < 0.25:
            lines.append("\tscale %.6g , %.6g" % (s, s*float(rng.uniform(0.7, 1.4))))
        elif rng.random() < 0.6:
            lines.append("\tscale %.6g" % s)
        if kind == "sphere":
            lines.append("\tradius %.6g" % float(rng.uniform(0.2, 1.0)))
        else:
            lines.append("\tmesh %s" % kind)
    lines.append("\tmaterial %s" % mat)
    if light_samples:
        lines.append("\tlightSamples %d" % light_samples)
    lines.append("}")
    return "\n".join(lines)


def scene_text(seed, nprim=None, max_planes=3):
    """nprim / max_planes: override the number of non-light primitives (default: 3-10, drawn) and the cap on planes -- the large flat-scan scenes of
    tests/test_fuzz.py (up to 64 primitives); the random stream is the same either way"""
    rng = np.random.default_rng(1000 + seed)
    W, H = int(rng.choice([24, 32, 40])), int(rng.choice([16, 24, 32]))
    out = ["# fuzz scene %d" % seed, "options", "{", "\twidth %d" % W, "\theight %d" % H,
           "\tmaxDepth %d" % int(rng.choice([1, 2, 3, 4, 6, 8])),
           "\tfilter %s %.6g %.6g" % (rng.choice(["gaussian", "box"]), float(rng.choice([0.5, 0.75, 1.0, 1.5, 2.0, 2.5])), float(rng.uniform(0.5, 3.0)))]
    if rng.random() < 0.3:
        out.append("\tclamp %.6g" % float(rng.uniform(0.5, 8.0)))
    out += ["}", "", "camera", "{", "\tposition %s" % f3(rng.uniform(-1, 1, 3) + np.array([0, 0.5, 6.0])),
            "\ttarget %s" % f3(rng.uniform(-0.5, 0.5, 3)), "\tfov %.6g" % float(rng.uniform(30, 70))]
    if rng.random() < 0.7:
        a = float(rng.uniform(0.0, 0.5))
        out += ["\tshutterstart %.6g" % a, "\tshutterend %.6g" % (a + float(rng.uniform(0.05, 1.0)))]
    out += ["}", "", "sky", "{", "\thorizon %s" % f3(rng.uniform(0, 1.2, 3)), "\tzenith %s" % f3(rng.uniform(0, 1.2, 3)), "}", ""]
    nmat = int(rng.integers(2, 6))
    nlight = int(rng.integers(1, 4))
    for m in range(nmat):
        out.append(material(rng, "m%d" % m, False))
    for m in range(nlight):
        out.append(material(rng, "e%d" % m, True))
    out.append(MESHES)
    drawn = int(rng.integers(3, 11))
    nprim = drawn if nprim is None else int(nprim)
    kinds = ["plane", "sphere", "sphere", "quad", "tetra", "wedge"]
    planes = 0
    for k in range(nprim):
        kind = kinds[int(rng.integers(0, len(kinds)))]
        if kind == "plane":
            planes += 1
            if planes > max_planes:
                kind = "sphere"
        out.append(primitive(rng, kind, "m%d" % int(rng.integers(0, nmat)), 0))
    for m in range(nlight):
        kind = ["sphere", "quad", "tetra"][int(rng.integers(0, 3))]     # planes cannot be sampled (intersection.h:901)
        out.append(primitive(rng, kind, "e%d" % m, int(rng.integers(1, 4))))
    return "\n\n".join(out) + "\n"


def main():
    R = RefOracle()
    d = tempfile.mkdtemp()
    out = {"count": np.int32(NUM_SCENES)}
    for k in range(NUM_SCENES):
        tin = os.path.join(d, "fuzz%02d.tin" % k)
        open(tin, "w").write(scene_text(k))
        h = R.load_tin(tin)
        pack = os.path.join(d, "fuzz%02d.pack" % k)
        R.write_pack(h, pack)
        cam, opt = R.camera_options(h)
        passes = 2
        accum, rad, _ = R.render_seeded(h, cam, opt, 3*k, passes, want_accum=True, want_radiance=True, threads=8)
        R.free(h)
        out["pack_%02d" % k] = np.frombuffer(open(pack, "rb").read(), np.uint8)
        out["radiance_%02d" % k] = rad
        out["accum_%02d" % k] = accum
        out["first_pass_%02d" % k] = np.int32(3*k)
        print("fuzz %02d: %dx%d depth %d prims %d finite %s mean %.4f" % (k, opt.width, opt.height, opt.max_depth, R.num_primitives(h) if False else -1,
                                                                           bool(np.isfinite(rad).all()), float(np.nanmean(rad))))
    np.savez_compressed(os.path.join(HERE, "fuzz.golden.npz"), **out)
    print("wrote fuzz.golden.npz (%.1f KB)" % (os.path.getsize(os.path.join(HERE, "fuzz.golden.npz"))/1e3))


if __name__ == "__main__":
    main()
