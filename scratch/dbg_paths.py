import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tinsel_amd
from tinsel_amd import abi
from tests.oracle_api import GOLDEN, image_l2
names = sys.argv[1:] or ["features"]
for name in names:
    g = np.load(os.path.join(GOLDEN, name + ".golden.npz"))
    scene = tinsel_amd.Scene.load_pack(os.path.join(GOLDEN, name + ".pack"))
    cam = abi.Camera.from_buffer_copy(g["camera"].tobytes()); opt = abi.Options.from_buffer_copy(g["options"].tobytes())
    passes = int(g["passes"])
    for pipe in (0, 1):
        r = tinsel_amd.create_gpu_renderer(scene); r.set_pipeline(pipe); r.init(opt.width, opt.height)
        out = r.render(cam, opt, passes=passes)
        rad = r.batch_radiance(passes, opt.height, opt.width)
        ref = g["radiance"]
        d = np.abs(rad - ref).max(axis=-1)
        rel = d/np.maximum(1e-3, np.abs(ref).max(axis=-1))
        bad = rel > 1e-3
        exact = (rad == ref).all(axis=-1)
        print("%s pipe=%d L2=%.3e paths=%d exact=%d bad=%d (%.2e)" % (name, pipe, image_l2(out, g["accum"]), bad.size, exact.sum(), bad.sum(), bad.mean()))
        idx = np.argwhere(bad)[:12]
        for s, j, i in idx:
            print("   pass %d pix (%d,%d) gpu %s ref %s" % (s, i, j, rad[s, j, i], ref[s, j, i]))
        # normals-mode primitive map to classify
        r.close()
