import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tinsel_amd
from tinsel_amd import abi
scene = tinsel_amd.Scene.load_pack(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/cornell.pack"))
cam, opt = scene.camera, scene.options.copy()
opt.width = opt.height = 1024
for world, rank, tile, passes in ((1, 0, 32, 64), (8, 4, 32, 64), (8, 4, 1024, 64), (8, 0, 1024, 64), (8, 4, 128, 64), (8, 4, 16, 64)):
    r = tinsel_amd.create_gpu_renderer(scene, 0)
    if world > 1:
        r.set_shard(rank, world, tile)
        r.set_batch_paths((8 << 20)*world)
    r.init(1024, 1024)
    r.enable_kernel_timing(True)
    r.render(cam, opt, passes=passes, readback=False)
    r.render(cam, opt, passes=passes, readback=False)
    print(world, rank, tile, passes, r.kernel_times(), r.stats()["samples"])
    r.close()
