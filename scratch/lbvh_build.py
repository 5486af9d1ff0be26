import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinsel_amd import Scene, create_gpu_renderer, abi
scene = Scene.load_pack(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/large/ajax_standin.pack"))
r = create_gpu_renderer(scene)
for i in range(5):
    print(r.set_mesh_bvh(abi.BVH_LBVH))
    r.set_mesh_bvh(abi.BVH_REFERENCE)
r.close()
