import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, tinsel_amd
scene = tinsel_amd.Scene.load_pack(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/cornell.pack"))
cam, opt = scene.camera, scene.options.copy()
for W in (256, 512):
    opt.width = opt.height = W
    r = tinsel_amd.create_gpu_renderer(scene, 0)
    r.init(W, W)
    r.render(cam, opt, passes=1, readback=False)
    for passes, readback in ((1, False), (1, True), (16, False)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            r.render(cam, opt, passes=passes, readback=readback)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0)/n
        print("%dx%d passes=%d readback=%s: %.3f ms per call (%.1f Msamples/s)" % (W, W, passes, readback, dt*1e3, passes*W*W/dt/1e6))
    r.close()
