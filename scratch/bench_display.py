"""Display-stage kernels at 4K: HIP-event time per launch and the streaming rate against the 8 TB/s HBM peak.
Algorithmic bytes per pixel: k_present 16 in + 16 out; k_nlm_means 16 + 16; k_nlm 32 in (image + means) + 16 out."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tinsel_amd import Scene, create_gpu_renderer
W, H = 3840, 2160
scene = Scene.load_pack(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/cornell.pack"))
opt = scene.options.copy(); opt.width, opt.height = W, H
r = create_gpu_renderer(scene)
r.init(W, H)
rng = np.random.default_rng(1)
acc = (rng.random((H, W, 4), dtype=np.float32)*4 + 0.1).astype(np.float32)
r.write_accum(acc, 0)
r.enable_kernel_timing(True)
res = {}
for it in range(6):
    r.present(opt, nlm_width=1, nlm_falloff=200.0, readback=False)
times = r.kernel_times()
bytes_px = {"k_present": 32, "k_nlm_means": 32, "k_nlm": 48}
out = {}
for name, b in bytes_px.items():
    launches, ms = times[name][:2]
    us = 1e3*ms/launches
    out[name] = {"launches": launches, "avg_us": us, "GB_s": W*H*b/us/1e3, "frac_of_8TBs": W*H*b/us/1e3/8000.0}
print(json.dumps({"display_4k": out}))
