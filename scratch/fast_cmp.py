import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, tinsel_amd
from tinsel_amd import abi
from tests.oracle_api import image_l2
def run(pack, W, H, depth, spp):
    scene = tinsel_amd.Scene.load_pack(os.path.join(ROOT, "tests/golden", pack + ".pack"))
    cam, opt = scene.camera, scene.options.copy(); opt.width, opt.height, opt.max_depth, opt.mode = W, H, depth, abi.MODE_PATHTRACE
    imgs = {}
    for a in (abi.ARITH_EXACT, abi.ARITH_FAST):
        r = tinsel_amd.create_gpu_renderer(scene); r.set_arithmetic(a); r.init(W, H); imgs[a] = r.render(cam, opt, passes=spp); r.close()
    return image_l2(imgs[0], imgs[1])
def speed(pack, W, H, depth, passes):
    scene = tinsel_amd.Scene.load_pack(os.path.join(ROOT, "tests/golden", pack + ".pack"))
    cam, opt = scene.camera, scene.options.copy(); opt.width, opt.height, opt.max_depth, opt.mode = W, H, depth, abi.MODE_PATHTRACE
    out = []
    for a in (abi.ARITH_EXACT, abi.ARITH_FAST):
        r = tinsel_amd.create_gpu_renderer(scene); r.set_arithmetic(a); r.init(W, H); r.reserve(passes, depth); r.render(cam, opt, passes=passes, readback=False)
        t0 = time.perf_counter(); r.render(cam, opt, passes=passes, readback=False); r.render(cam, opt, passes=passes, readback=False); dt = (time.perf_counter() - t0)/2; r.close()
        out.append(passes*W*H/dt/1e6)
    return out
print(os.environ.get("TINSEL_HIP_LIB"))
print("L2 @256spp: cornell256 %.2e glass %.2e veach %.2e features_probe %.2e" % (run("cornell", 256, 256, 4, 256), run("glass", 240, 135, 12, 256), run("veach", 240, 135, 4, 256), run("features_probe", 192, 128, 6, 256)))
print("Msamples/s exact/fast: cornell1024 %s glass1080 %s veach4k %s" % (speed("cornell", 1024, 1024, 4, 64), speed("glass", 1920, 1080, 12, 16), speed("veach", 3840, 2160, 4, 8)))
