"""Dev tool: the fuzz generator's random scenes (tests/golden/make_fuzz.py: every primitive and material kind, moving primitives, 3-13 primitives)
at a real frame size -- 320 x 240 x 32 spp = 2.5e6 paths per scene instead of the suite's ~1e3 -- against the reference on this box.
   python scratch/fuzz_at_scale.py first count [W H spp]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tinsel_amd
from tinsel_amd import abi
from tests import oracle_api as oa
from tests.golden.make_fuzz import scene_text
first, count = int(sys.argv[1]), int(sys.argv[2])
W, H, spp = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (320, 240, 32)
R = oa.RefOracle()
d = tempfile.mkdtemp()
bad, total = [], 0
for k in range(first, first + count):
    tin = os.path.join(d, "f%d.tin" % k)
    open(tin, "w").write(scene_text(k))
    h = R.load_tin(tin)
    pack = os.path.join(d, "f%d.pack" % k)
    R.write_pack(h, pack)
    cam, opt = R.camera_options(h)
    opt.width, opt.height = W, H
    want, _, _ = R.render_seeded(h, cam, opt, k, spp)
    R.free(h)
    scene = tinsel_amd.Scene.load_pack(pack)
    for pipe in (abi.PIPELINE_AUTO, abi.PIPELINE_WAVEFRONT_SPLIT):
        r = tinsel_amd.create_gpu_renderer(scene, 0, abi.Tuning(walk_min_tris=0, small_mesh_bytes=0) if pipe == abi.PIPELINE_WAVEFRONT_SPLIT else None)
        r.set_pipeline(pipe)
        r.init(W, H)
        r.set_pass_index(k)
        out = r.render(cam, opt, passes=spp)
        r.close()
        both_nan = np.isnan(out) & np.isnan(want)
        diff = ((out != want) & ~both_nan).any(axis=-1)
        if diff.any():
            bad.append((k, pipe, int(diff.sum())))
    total += W*H*spp
    os.remove(tin); os.remove(pack)
print("seeds %d..%d at %dx%dx%d spp: %.2e paths, scenes that differ (seed, pipeline, pixels): %s" % (first, first + count - 1, W, H, spp, total, bad))
