import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tinsel_amd
from tinsel_amd import abi
scene = tinsel_amd.Scene.load_pack("tests/golden/%s.pack" % (sys.argv[1] if len(sys.argv) > 1 else "cornell"))
cam, opt = scene.camera, scene.options.copy()
opt.width = opt.height = 2048
opt.mode = abi.MODE_NORMALS
r = tinsel_amd.create_gpu_renderer(scene); r.init(opt.width, opt.height)
r.render_async(cam, opt, passes=1); torch.cuda.synchronize()
r.enable_kernel_timing(True)
for _ in range(10): r.render_async(cam, opt, passes=1)
torch.cuda.synchronize()
kt = r.kernel_times()
print(kt, "Grays/s %.2f" % (10*opt.width*opt.height/(kt['k_normals'][1]*1e-3)/1e9))
