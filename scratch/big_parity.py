"""Dev tool: whole frames of the OTHER scenes (every fixture and shipped scene that is not a BASELINE configuration) at hundreds of millions of
paths each against the reference on this box: pixels that differ.   python scratch/big_parity.py [scale]   (scale 1 = ~3e8 paths per scene)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tinsel_amd
from tinsel_amd import abi
from tests import oracle_api as oa
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
G = oa.GOLDEN
SCENES = [  # pack, W, H, spp
    ("features", 1280, 720, 320), ("features_probe", 1280, 720, 256), ("gloss", 1024, 1024, 256), ("cornell_probe", 1024, 1024, 256),
    ("motionblur", 1280, 720, 256), ("many_spheres", 1024, 768, 320), ("emitter", 512, 512, 512), ("furnace", 512, 512, 512),
    ("conservation", 512, 512, 512), ("simple", 512, 512, 512), ("one_sphere", 512, 512, 512),
    ("large/env_loft", 1280, 720, 256), ("large/env", 1280, 720, 256), ("large/example", 1280, 720, 256),
    ("large/table", 1280, 720, 192), ("large/transmission", 1280, 720, 192), ("large/meshlight", 1280, 720, 192),
]
R = oa.RefOracle()
total = 0
for name, W, H, spp in SCENES:
    pack = os.path.join(G, name + ".pack")
    if not os.path.exists(pack):
        print("%-22s not on this box" % name)
        continue
    spp = max(1, int(spp*scale))
    scene = tinsel_amd.Scene.load_pack(pack)
    cam, opt = scene.camera, scene.options.copy()
    opt.width, opt.height, opt.mode = W, H, abi.MODE_PATHTRACE
    r = tinsel_amd.create_gpu_renderer(scene)
    r.init(W, H)
    t0 = time.perf_counter()
    out = r.render(cam, opt, passes=spp)
    tg = time.perf_counter() - t0
    r.close()
    h = R.load_pack(pack)
    t0 = time.perf_counter()
    want, _, _ = R.render_seeded(h, cam, opt, 0, spp)
    tc = time.perf_counter() - t0
    R.free(h)
    bad = (out != want).any(axis=-1) & ~(np.isnan(out).any(axis=-1) & np.isnan(want).any(axis=-1))
    nanmis = int((np.isnan(out) != np.isnan(want)).any(axis=-1).sum())
    total += W*H*spp
    print("%-22s %4dx%-4d depth %2d spp %4d = %.2e paths: %d pixels differ (%d with NaN on one side only); GPU %.1f s, CPU %.0f s" % (
        name, W, H, opt.max_depth, spp, W*H*spp, int(bad.sum()), nanmis, tg, tc), flush=True)
    if bad.any():
        ys, xs = np.nonzero(bad)
        print("    first:", list(zip(ys.tolist(), xs.tolist()))[:6])
print("total %.2e paths" % total)
