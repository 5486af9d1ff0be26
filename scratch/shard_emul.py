"""One shard of N, alone on the device: what a rank's tile numbering costs against the one-shard row-major numbering (weak scaling: N x passes
over 1/N of the pixels = the same number of paths).  usage: shard_emul.py scene width height world [maxdepth]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tinsel_amd
from tinsel_amd import abi

name, W, H, world = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
scene = tinsel_amd.Scene.load_pack(os.path.join(ROOT, "tests", "golden", name + ".pack"))
opt = scene.options.copy()
opt.width, opt.height = W, H
if len(sys.argv) > 5:
    opt.max_depth = int(sys.argv[5])
opt.mode = abi.MODE_PATHTRACE
steps = 20


def measure(tile, rank=0):
    r = tinsel_amd.create_gpu_renderer(scene, 0)
    n = 1 if tile == 0 else world
    if tile:
        r.set_shard(rank, world, tile)
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    r.init(W, H, accum_tensor=accum)
    st = torch.cuda.current_stream().cuda_stream
    r.reserve(steps*n, opt.max_depth)
    r.render_async(scene.camera, opt, passes=5*n, stream=st); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        r.render_async(scene.camera, opt, passes=steps*n, stream=st); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    r.enable_kernel_timing(True)
    r.render_async(scene.camera, opt, passes=steps*n, stream=st); torch.cuda.synchronize()
    kt = {k: round(v[1], 3) for k, v in r.kernel_times().items() if v[0]}
    r.close()
    return best, kt


base, kt = measure(0)
print("| %s %dx%d | one shard, row-major | %.2f ms | 1.000 | %s |" % (name, W, H, base*1e3, kt))
for tile in (32, 64, 128):
    for rank in (0, 3):
        t, kt = measure(tile, rank)
        print("| %s %dx%d | rank %d of %d, tile %d | %.2f ms | %.3f | %s |" % (name, W, H, rank, world, tile, t*1e3, base/t, kt))
