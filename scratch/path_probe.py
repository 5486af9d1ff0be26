"""Dev tool: ONE path (pixel, pass) of a scene through every pipeline and at every maxDepth, against the reference and the C restatement.
   python scratch/path_probe.py pack W H y x pass maxDepth"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tinsel_amd
from tinsel_amd import abi
from tests import oracle_api as oa
pack, W, H, y, x, p, D = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
scene = tinsel_amd.Scene.load_pack(pack)
cam = scene.camera
R, P = oa.RefOracle(), oa.PortOracle()
hr, hp = R.load_pack(pack), P.load_pack(pack)
for depth in range(1, D + 1):
    opt = scene.options.copy()
    opt.width, opt.height, opt.max_depth, opt.mode = W, H, depth, abi.MODE_PATHTRACE
    _, rr, _ = R.render_seeded(hr, cam, opt, p, 1, window=(x, y, x + 1, y + 1), want_accum=False, want_radiance=True)
    _, rp, _ = P.render_seeded(hp, cam, opt, p, 1, window=(x, y, x + 1, y + 1), want_accum=False, want_radiance=True)
    line = "depth %d ref %s port %s" % (depth, rr[0, 0, 0].view(np.uint32), rp[0, 0, 0].view(np.uint32))
    for name, pipe, tune in (("fused", abi.PIPELINE_WAVEFRONT, None), ("split", abi.PIPELINE_WAVEFRONT_SPLIT, None), ("mega", abi.PIPELINE_MEGAKERNEL, None),
                             ("split-bvh", abi.PIPELINE_WAVEFRONT_SPLIT, abi.Tuning(flat_scan=0))):
        r = tinsel_amd.create_gpu_renderer(scene, 0, tune)
        r.set_pipeline(pipe)
        r.init(W, H)
        r.set_pass_index(p)
        r.render(cam, opt, passes=1, readback=False)
        g = r.batch_radiance(1, H, W)[0, y, x]
        r.close()
        line += " | %s %s%s" % (name, g.view(np.uint32), "" if np.array_equal(g, rr[0, 0, 0]) else " DIFFERS")
    print(line)
    print("   ref", rr[0, 0, 0], "port", rp[0, 0, 0])
