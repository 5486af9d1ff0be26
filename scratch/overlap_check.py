"""Dev check: a batch's passes as two overlapped chunks (tinsel_hip_tuning::overlap = 1) against one chunk, at full size: same accumulator bit for bit?  how long?
   python scratch/overlap_check.py [scene pack under tests/golden] [pipeline] [repeats]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tinsel_amd import Scene, create_gpu_renderer, abi
name = sys.argv[1] if len(sys.argv) > 1 else "large/transmission"
pipe = {"auto": abi.PIPELINE_AUTO, "split": abi.PIPELINE_WAVEFRONT_SPLIT, "paired": abi.PIPELINE_WAVEFRONT_PAIRED}[sys.argv[2] if len(sys.argv) > 2 else "auto"]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
scene = Scene.load_pack("tests/golden/%s.pack" % name)
cam, opt = scene.camera, scene.options
opt.width, opt.height = 1920, 1080
ref = None
for tune in (None, abi.Tuning(overlap=1)):
    r = create_gpu_renderer(scene, 0, tune)
    r.set_pipeline(pipe)
    r.init(opt.width, opt.height)
    times = []
    outs = []
    for k in range(reps):
        r.init(opt.width, opt.height)
        t0 = time.time()
        out = r.render(cam, opt, passes=20)
        times.append(time.time() - t0)
        outs.append(out.copy())
    r.close()
    same = all(np.array_equal(outs[0], o) for o in outs)
    if ref is None:
        ref = outs[0]
    print("%s overlap=%s: render calls %s s; repeats identical: %s; equals one-chunk: %s" % (name, tune is not None, " ".join("%.3f" % t for t in times), same, np.array_equal(ref, outs[0])))
