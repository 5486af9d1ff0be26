"""Dev tool: BVH node visits / triangle tests per EXTENSION ray, by bounce (renders with maxDepth 1, 2, ... under the
detail counters and differences the totals).  Shadow rays are counted with their bounce."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinsel_amd
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
scene = tinsel_amd.Scene.load_pack(os.path.join(root, "tests/golden/large/ajax_standin.pack"))
cam, opt = scene.camera, scene.options.copy()
opt.width, opt.height = 1920, 1080
prev = None
for depth in (1, 2, 3, 4):
    opt.max_depth = depth
    r = tinsel_amd.create_gpu_renderer(scene, 0)
    r.set_detail_counters(True)
    r.init(opt.width, opt.height)
    r.render(cam, opt, passes=2, readback=False)
    s = r.stats()
    r.close()
    cur = {k: s[k] for k in ("rays", "shadow_rays", "internal_visits", "tri_tests", "prim_tests")}
    d = cur if prev is None else {k: cur[k] - prev[k] for k in cur}
    print("bounce %d: rays %9d (shadow %9d)  node visits/ray %.2f  tri tests/ray %.2f  prim tests/ray %.2f" % (
        depth - 1, d["rays"], d["shadow_rays"], d["internal_visits"]/max(1, d["rays"]), d["tri_tests"]/max(1, d["rays"]), d["prim_tests"]/max(1, d["rays"])))
    prev = cur
