#!/usr/bin/env python3
"""SURVEY.md 8(f) rank 4, measured: motion blur -- "static instances free, moving ones cheap" -- and NLM.

  1. the reference's own data/motionblur.tin (tests/golden/motionblur.pack: the octopus mesh turns half a revolution about y while the
     shutter is open) at 1920x1080, as written, and its STATIC twin -- the same renderer after tinsel_hip_set_primitive_transform(mesh,
     start, start) + tinsel_hip_rebuild_scene -- : Msamples/s, kernel ms, the share of the rays that walk the moving mesh, and the
     difference per such ray (a moving primitive costs its rays one InterpolateTransform -- nlerp + normalise, maths.h:1566-1569 -- where
     a static one reads a pre-interpolated pose);
  2. the features fixture (two moving primitives among nine) the same way;
  3. NonLocalMeansFilter (nlm.cpp:33-73) at 1920x1080, radius 1 and 2: the device kernels' time against the reference's own function on
     one host core (it is a serial double loop).
Prints a markdown table."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import tinsel_amd  # noqa: E402
from tinsel_amd import abi  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def timed(r, cam, opt, passes):
    r.init(opt.width, opt.height)
    r.reserve(passes, opt.max_depth)
    r.render(cam, opt, passes=2, readback=False)
    best = None
    r.enable_kernel_timing(True)
    for _ in range(5):
        r.reset_stats()
        t0 = time.perf_counter()
        r.render(cam, opt, passes=passes, readback=False)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, r.stats(), {k: round(v[2], 3) for k, v in r.kernel_times().items()}, r.queue_counts())
    r.enable_kernel_timing(False)
    return best


def scene_pair(name, W, H, depth, passes):
    scene = tinsel_amd.Scene.load_pack(os.path.join(GOLD, name + ".pack"))
    cam, opt = scene.camera, scene.options.copy()
    opt.width, opt.height, opt.mode = W, H, abi.MODE_PATHTRACE
    if depth:
        opt.max_depth = depth
    prims = C.cast(scene.desc.primitives, C.POINTER(abi.Primitive))
    movers = [i for i in range(scene.desc.num_primitives) if bytes(prims[i].start_transform) != bytes(prims[i].end_transform)]
    rows = []
    r = tinsel_amd.create_gpu_renderer(scene)
    walked = r.walked_prims
    for label in ("as written (%d moving of %d primitives)" % (len(movers), scene.desc.num_primitives), "static twin (end := start)"):
        dt, st, kms, (alive, _) = timed(r, cam, opt, passes)
        rows.append((label, passes*W*H/dt/1e6, st["rays"]/dt/1e6, st["rays"]/max(1, st["samples"]), kms, dt))
        for i in movers:
            s = abi.Transform.from_buffer_copy(bytes(prims[i].start_transform))
            r.set_primitive_transform(i, s, s)
        r.rebuild_scene()
    r.close()
    print("\n### %s %dx%d maxDepth %d, %d passes per batch (%d primitive(s) walked by k_walk)\n" % (name, W, H, opt.max_depth, passes, walked))
    print("| scene | Msamples/s | Mrays/s | rays per sample | kernel ms of the batch |\n|---|---|---|---|---|")
    for label, ms_, mr, rps, kms, dt in rows:
        print("| %s | %.1f | %.1f | %.2f | %s |" % (label, ms_, mr, rps, kms))
    a, b = rows
    print("\nmoving / static: %.3f x the time (%.2f ms against %.2f ms per batch)" % (a[5]/b[5], a[5]*1e3, b[5]*1e3))


def nlm(W, H):
    from tests import oracle_api as oa
    scene = tinsel_amd.Scene.load_pack(os.path.join(GOLD, "cornell.pack"))
    opt = scene.options.copy()
    opt.width, opt.height = W, H
    r = tinsel_amd.create_gpu_renderer(scene)
    r.init(W, H)
    rng = np.random.default_rng(1)
    acc = (rng.random((H, W, 4), dtype=np.float32)*4 + 0.1).astype(np.float32)
    r.write_accum(acc, 0)
    print("\n### NonLocalMeansFilter %dx%d (nlm.cpp:33-73): device kernels against the reference's function on one host core\n" % (W, H))
    print("| radius | k_nlm_means + k_nlm, ms per frame | reference on the host, ms per frame (1 core) | ratio | images equal |\n|---|---|---|---|---|")
    R = oa.RefOracle() if oa.have_ref() else None
    for radius in (1, 2):
        r.enable_kernel_timing(True)
        for _ in range(4):
            r.present(opt, nlm_width=radius, nlm_falloff=200.0, readback=False)
        kt = r.kernel_times()
        r.enable_kernel_timing(False)
        dev_ms = (kt["k_nlm_means"][1] + kt["k_nlm"][1])/max(1, kt["k_nlm"][0])
        dev = r.present(opt, nlm_width=radius, nlm_falloff=200.0)
        filtered = r.present(opt, nlm_width=0)
        host_ms, same = float("nan"), "-"
        if R is not None:
            t0 = time.perf_counter()
            out = R.nlm(filtered, 200.0, radius)
            host_ms = (time.perf_counter() - t0)*1e3
            same = str(bool(np.array_equal(out, dev)))
        print("| %d | %.3f | %.1f | %.0f x | %s |" % (radius, dev_ms, host_ms, host_ms/dev_ms if dev_ms > 0 else float("nan"), same))
    r.close()


if __name__ == "__main__":
    scene_pair("motionblur", 1920, 1080, 4, 16)
    scene_pair("features", 1920, 1080, 6, 16)
    nlm(1920, 1080)
