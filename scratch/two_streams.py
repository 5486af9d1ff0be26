"""Dev experiment: do two independent batches on two streams overlap usefully (tails of one under the other)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tinsel_amd
name, W, H, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
scene = tinsel_amd.Scene.load_pack(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/%s.pack" % name))
cam, opt = scene.camera, scene.options.copy()
opt.width, opt.height = W, H
if name == "glass": opt.max_depth = 12
def make():
    r = tinsel_amd.create_gpu_renderer(scene, 0); r.init(W, H); r.reserve(steps, opt.max_depth); return r
a, b = make(), make()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for r, s in ((a, s1), (b, s2)):
    r.render_async(cam, opt, passes=8, stream=s.cuda_stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
a.render_async(cam, opt, passes=steps, stream=s1.cuda_stream)
torch.cuda.synchronize(); t1 = time.perf_counter() - t0
t0 = time.perf_counter()
a.render_async(cam, opt, passes=steps//2, stream=s1.cuda_stream)
b.render_async(cam, opt, passes=steps//2, stream=s2.cuda_stream)
torch.cuda.synchronize(); t2 = time.perf_counter() - t0
print("%s %dx%d %d passes: one stream %.2f ms (%.1f Msamples/s); two streams x half each %.2f ms (%.1f Msamples/s)" % (
    name, W, H, steps, t1*1e3, steps*W*H/t1/1e6, t2*1e3, steps*W*H/t2/1e6))
