"""Dev tool: section cycle shares of k_bounce (library built with -DTN_PROFILE_SECTIONS)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tinsel_amd import Scene, create_gpu_renderer, abi
name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
W = H = 1024
scene = Scene.load_pack("tests/golden/%s.pack" % name)
cam, opt = scene.camera, scene.options
opt.width, opt.height = W, H
if len(sys.argv) > 2: opt.max_depth = int(sys.argv[2])
r = create_gpu_renderer(scene)
r.set_pipeline(abi.PIPELINE_WAVEFRONT)
r.init(W, H)
r.render(cam, opt, passes=8)
r.reset_stats()
r.render(cam, opt, passes=8)
s = r.stats()
v = [s["internal_visits"], s["tri_tests"], s["prim_tests"], s["shadow_rays"], s["_6"], s["_7"]]
names = ["load/generate", "closest trace", "hit begin + NEE prepare/resolve", "NEE trace", "store/append/loop", "bsdf_step/on_miss"]
tot = float(sum(v))
for n, x in zip(names, v):
    print("%-34s %5.1f %%" % (n, 100.0*x/tot))
print("rays", s["rays"], "samples", s["samples"])
