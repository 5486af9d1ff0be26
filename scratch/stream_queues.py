"""Dev tool: do the split pipeline's streaming kernels run at a different rate on a different HIP stream (hardware queue) of the same process?"""
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
r = tinsel_amd.create_gpu_renderer(scene, 0)
accum = torch.zeros((opt.height, opt.width, 4), dtype=torch.float32, device="cuda")
r.init(opt.width, opt.height, accum_tensor=accum)
r.reserve(20, opt.max_depth)
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(5)]
for rnd in range(2):
    for i, s in enumerate(streams):
        torch.cuda.synchronize()
        r.enable_kernel_timing(False)
        r.render_async(scene.camera, opt, passes=20, stream=s.cuda_stream); torch.cuda.synchronize()
        r.enable_kernel_timing(True)
        t0 = time.perf_counter()
        r.render_async(scene.camera, opt, passes=20, stream=s.cuda_stream); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        kt = {k: round(v[1], 2) for k, v in r.kernel_times().items() if v[0]}
        print("round %d stream %d: %.2f ms  %s" % (rnd, i, dt*1e3, kt))
