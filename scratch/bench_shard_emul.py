"""Dev tool: what ONE rank of an N-GPU weak-scaling run does, on one GPU (no RCCL): rank R of N renders steps*N passes
over its 1/N of the tiles.  Compare its wall time with the N = 1 run of the same steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tinsel_amd
from tinsel_amd import abi
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
scene = tinsel_amd.Scene.load_pack(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/cornell.pack"))
cam, opt = scene.camera, scene.options.copy()
opt.width = opt.height = 1024
for world in (1, 2, 4, 8):
    r = tinsel_amd.create_gpu_renderer(scene, 0)
    if world > 1:
        r.set_shard(world//2, world, 32)
        r.set_batch_paths((8 << 20)*world)
    r.init(1024, 1024)
    r.enable_kernel_timing(True)
    r.render(cam, opt, passes=8*world, readback=False)          # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r.render(cam, opt, passes=steps*world, readback=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = r.stats()
    print("N=%d  rank %d: %.2f ms for %d passes -> %.1f Msamples/s per rank (x%d = %.0f aggregate), kernels %s" % (
        world, world//2, dt*1e3, steps*world, steps*1024*1024/dt/1e6, world, world*steps*1024*1024/dt/1e6,
        {k: round(v[2], 2) for k, v in r.kernel_times().items()}))
    r.close()
