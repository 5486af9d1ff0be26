#!/usr/bin/env python3
"""k_walk section profile (library built with -DTN_WALK_PROF, loaded through TINSEL_HIP_LIB): renders `passes` passes of a
scene pack at WxH and lets tinsel_hip_destroy print the counters."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinsel_amd
from tinsel_amd import abi
pack, W, H, depth, passes = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
scene = tinsel_amd.Scene.load_pack(os.path.join(ROOT, "tests", "golden", pack + ".pack"))
cam, opt = scene.camera, scene.options.copy()
opt.width, opt.height, opt.max_depth, opt.mode = W, H, depth, abi.MODE_PATHTRACE
r = tinsel_amd.create_gpu_renderer(scene)
r.init(W, H)
r.reserve(passes, depth)
r.render(cam, opt, passes=2, readback=False)
r.enable_kernel_timing(True)
t0 = time.perf_counter()
r.render(cam, opt, passes=passes, readback=False)
dt = time.perf_counter() - t0
print(pack, "%.1f Msamples/s" % (passes*W*H/dt/1e6), {k: round(v[2], 2) for k, v in r.kernel_times().items()}, flush=True)
r.close()
