"""Dev tool: is the speed of the split pipeline's streaming kernels a property of the PROCESS or of the path state's allocation?  One process,
the renderer (and with it every array of the path state) created and destroyed several times, with decoy allocations of growing size held
in between so that the state lands somewhere else each time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tinsel_amd
from tinsel_amd import abi
scene = tinsel_amd.Scene.load_pack(os.path.join(ROOT, "tests", "golden", "glass.pack"))
opt = scene.options.copy()
opt.width, opt.height, opt.max_depth = 1920, 1080, 12
opt.mode = abi.MODE_PATHTRACE
decoys = []
for it in range(8):
    r = tinsel_amd.create_gpu_renderer(scene, 0)
    accum = torch.zeros((opt.height, opt.width, 4), dtype=torch.float32, device="cuda")
    r.init(opt.width, opt.height, accum_tensor=accum)
    r.reserve(20, opt.max_depth)
    s = torch.cuda.current_stream().cuda_stream
    r.render_async(scene.camera, opt, passes=20, stream=s); torch.cuda.synchronize()
    r.enable_kernel_timing(True)
    r.render_async(scene.camera, opt, passes=20, stream=s); torch.cuda.synchronize()
    kt = {k: round(v[1], 2) for k, v in r.kernel_times().items() if v[0]}
    print("state %d (decoys held: %.1f GiB): %s" % (it, sum(d.numel() for d in decoys)/(1 << 30), kt))
    r.close()
    del accum
    if it % 2 == 1:
        decoys.append(torch.empty(int((0.7 + 0.9*it)*(1 << 30)), dtype=torch.uint8, device="cuda"))
