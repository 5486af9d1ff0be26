import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, tinsel_amd
from tinsel_amd import abi
def speed(pack, W, H, depth, passes, arith):
    scene = tinsel_amd.Scene.load_pack(os.path.join(ROOT, "tests/golden", pack + ".pack"))
    cam, opt = scene.camera, scene.options.copy(); opt.width, opt.height, opt.max_depth, opt.mode = W, H, depth, abi.MODE_PATHTRACE
    r = tinsel_amd.create_gpu_renderer(scene); r.set_arithmetic(arith); r.init(W, H); r.reserve(passes, depth); r.render(cam, opt, passes=passes, readback=False)
    r.enable_kernel_timing(True)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); r.render(cam, opt, passes=passes, readback=False); ts.append(time.perf_counter() - t0)
    kt = r.kernel_times(); r.close()
    return "%.0f (k_shade %.1f ms)" % (passes*W*H/min(ts)/1e6, kt.get("k_shade", (0, 0))[1])
print(os.environ.get("TINSEL_HIP_LIB", "default"))
for a, n in ((abi.ARITH_EXACT, "exact"), (abi.ARITH_FAST, "fast")):
    print(n, "ajax", speed("large/ajax_standin", 1920, 1080, 4, 32, a), "glass", speed("glass", 1920, 1080, 12, 16, a), "veach4k", speed("veach", 3840, 2160, 4, 8, a), "features", speed("features", 1920, 1080, 6, 8, a))
