#!/usr/bin/env python3
"""What a frame of an animation costs to get READY: the reference's batch mode re-creates its renderer per frame (main.cpp:318-327: loader,
Scene::Build, a new GpuRenderer -- every mesh converted and uploaded again); the headless drivers here keep one renderer and rewrite the moved
primitives (HipRenderer.update_scene).  Makes N frames of data/ajax.tin + the 427,384-triangle Aphrodite scan with the statue translating and
turning (oracle/_ref on this box: ref_scene_set_transform = mutate + the reference's own Scene::Build), runs the batch, and times a re-create
per frame beside it.  Needs tests/golden/large/ajax_aphrodite.pack and oracle/_ref."""
import ctypes as C
import math
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.oracle_api import RefOracle  # noqa: E402
from tinsel_amd import Scene, abi, create_gpu_renderer  # noqa: E402

FRAMES = 4
PACK = os.path.join(ROOT, "tests", "golden", "large", "ajax_aphrodite.pack")


def main():
    tmp = tempfile.mkdtemp(prefix="tinsel_anim_")
    R = RefOracle()
    R.lib.ref_scene_set_transform.argtypes = [C.c_void_p, C.c_int, C.POINTER(abi.Transform), C.POINTER(abi.Transform)]
    for k in range(FRAMES):
        h = R.load_pack(PACK)
        p = R.primitive(h, 1)                                       # the mesh
        s = abi.Transform.from_buffer_copy(bytes(p.start_transform))
        a = 0.15*k
        s.p.x += 0.1*k
        s.r.x, s.r.y, s.r.z, s.r.w = 0.0, math.sin(a/2), 0.0, math.cos(a/2)      # turning about y
        assert R.lib.ref_scene_set_transform(h, 1, C.byref(s), C.byref(s)) == 0
        R.write_pack(h, os.path.join(tmp, "f%d.pack" % k))
        R.free(h)
    env = dict(os.environ, PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, "-m", "tinsel_amd.headless", "-spp=16", "-width=1920", "-height=1080", os.path.join(tmp, "f%d.pack")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    print("\n".join(l for l in p.stdout.splitlines() if l.startswith("frame ") or " frames;" in l))
    if p.returncode:
        print(p.stderr[-2000:])
    # the reference's pattern: a NEW renderer per frame (pack already in memory both ways: the file read is the same for both)
    times = []
    for k in range(FRAMES):
        scene = Scene.load_pack(os.path.join(tmp, "f%d.pack" % k))
        t0 = time.perf_counter()
        r = create_gpu_renderer(scene)
        r.init(1920, 1080)
        times.append((time.perf_counter() - t0)*1e3)
        r.close()
    print("re-create per frame (tinsel_hip_create + init, pack in memory): " + ", ".join("%.1f ms" % t for t in times))
    # and the in-place update alone, pack in memory
    scenes = [Scene.load_pack(os.path.join(tmp, "f%d.pack" % k)) for k in range(FRAMES)]
    r = create_gpu_renderer(scenes[0])
    ups = []
    for k in range(1, FRAMES):
        t0 = time.perf_counter()
        ok = r.update_scene(scenes[k - 1], scenes[k])
        r.init(1920, 1080)
        ups.append((time.perf_counter() - t0)*1e3)
        assert ok
    print("update in place per frame (scene_delta's comparison of 40 MB of mesh arrays + set_primitive_transform + rebuild_scene + init): " + ", ".join("%.1f ms" % t for t in ups))
    # ... and what a caller pays who KNOWS what moved (an animation system): the two C-ABI calls alone
    from tinsel_amd.renderer import scene_delta
    direct = []
    for k in list(range(1, FRAMES)) + [0]:
        moves, nodes = scene_delta(scenes[k - 1], scenes[k])
        t0 = time.perf_counter()
        for i, s, e in moves:
            r.set_primitive_transform(i, s, e)
        r.rebuild_scene(nodes)
        direct.append((time.perf_counter() - t0)*1e3)
    r.close()
    print("tinsel_hip_set_primitive_transform + tinsel_hip_rebuild_scene alone: " + ", ".join("%.3f ms" % t for t in direct))


if __name__ == "__main__":
    main()
