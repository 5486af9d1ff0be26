"""Dev tool: cycle shares inside trace_flat (library built with -DTN_PROFILE_TRACE=1 closest / =2 NEE)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinsel_amd import Scene, create_gpu_renderer, abi
name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
W = H = 1024
scene = Scene.load_pack("tests/golden/%s.pack" % name)
cam, opt = scene.camera, scene.options
opt.width, opt.height = W, H
r = create_gpu_renderer(scene)
r.set_pipeline(abi.PIPELINE_WAVEFRONT)
r.init(W, H)
r.render(cam, opt, passes=8)
r.reset_stats()
r.render(cam, opt, passes=8)
s = r.stats()
v = [s["internal_visits"], s["tri_tests"], s["prim_tests"], s["shadow_rays"], s["_6"], s["_7"]]
names = ["box tests", "planes", "spheres", "meshes", "loop/bookkeeping", "BVH fallback walk"]
tot = float(sum(v))
for n, x in zip(names, v):
    print("%-20s %5.1f %%  %8.0f cycles/wave-item" % (n, 100.0*x/tot, x/ (s["rays"]/64.0)))
