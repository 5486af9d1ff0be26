"""Dev tool: where does a whole frame differ from the reference?  GPU frame against the oracle's, the differing pixels, and for a window around
the first of them the per-path radiance pass by pass (GPU against oracle).   python scratch/find_diff.py pack W H depth spp [lib.so]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 6:
    os.environ["TINSEL_HIP_LIB"] = sys.argv[6]
import numpy as np
import tinsel_amd
from tinsel_amd import abi
from tests import oracle_api as oa
pack, W, H, depth, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
first = int(os.environ.get("FIND_DIFF_FIRST_PASS", "0"))     # (the fuzz scenes are rendered from pass index = their seed)
scene = tinsel_amd.Scene.load_pack(pack)
cam, opt = scene.camera, scene.options.copy()
opt.width, opt.height, opt.mode = W, H, abi.MODE_PATHTRACE
if depth > 0:
    opt.max_depth = depth
r = tinsel_amd.create_gpu_renderer(scene)
r.init(W, H)
r.set_pass_index(first)
out = r.render(cam, opt, passes=spp)
O = oa.RefOracle()
h = O.load_pack(pack)
cache = "/tmp/find_diff_%s_%d_%d_%d_%d.npy" % (os.path.basename(pack), W, H, depth, spp)
if os.path.exists(cache):
    want = np.load(cache)
else:
    want, _, _ = O.render_seeded(h, cam, opt, first, spp)
    np.save(cache, want)
ys, xs = np.nonzero((out != want).any(axis=-1))
print("lib", os.environ.get("TINSEL_HIP_LIB", "in-tree"), ":", len(ys), "pixels differ:", list(zip(ys.tolist(), xs.tolist()))[:8])
for y, x in list(zip(ys.tolist(), xs.tolist()))[:4]:
    print("  pixel", (y, x), "gpu", out[y, x], "cpu", want[y, x])
if len(ys):
    y0, x0 = int(ys[0]), int(xs[0])
    wy0, wy1, wx0, wx1 = max(0, y0 - 3), min(H, y0 + 4), max(0, x0 - 3), min(W, x0 + 4)
    _, rad_cpu, _ = O.render_seeded(h, cam, opt, first, spp, window=(wx0, wy0, wx1, wy1), want_accum=False, want_radiance=True)
    for p in range(spp):
        r.init(W, H)
        r.set_pass_index(first + p)
        r.render(cam, opt, passes=1, readback=False)
        g = r.batch_radiance(1, H, W)[0, wy0:wy1, wx0:wx1]
        c = rad_cpu[p]
        bad = np.nonzero((g != c).any(axis=-1))
        for yy, xx in zip(bad[0].tolist(), bad[1].tolist()):
            print("  pass", first + p, "pixel", (wy0 + yy, wx0 + xx), "gpu", g[yy, xx], "cpu", c[yy, xx], "bits", g[yy, xx].view(np.uint32), c[yy, xx].view(np.uint32))
O.free(h)
r.close()
