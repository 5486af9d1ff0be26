"""Dev tool: cycle shares of trace_flat's sections inside k_bounce (library built with -DTN_PROFILE_TRACE=1: the closest-hit traces, =2: the
shadow traces).  usage: trace_sections.py scene"""
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
names = ["leaf boxes + loop", "planes", "spheres", "meshes (inline / deferred walks)", "loop head", "fallback BVH walk"]
tot = float(sum(v))
for n, x in zip(names, v):
    print("%-34s %5.1f %%  %.3e" % (n, 100.0*x/max(tot, 1.0), x))
