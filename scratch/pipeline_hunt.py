"""Dev tool: the BASELINE configurations at ~2e9 paths through EVERY pipeline and a few tunings, against the AUTO pipeline's frame (which
profiles/r06_3i_big_spp_parity.txt holds to the reference at the same sizes): pixels that differ.   python scratch/pipeline_hunt.py [scale]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tinsel_amd
from tinsel_amd import abi
from tests import oracle_api as oa
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
G = oa.GOLDEN
CONFIGS = [("cornell", 1024, 1024, 4, 2048), ("large/ajax_standin", 1920, 1080, 4, 1024), ("large/ajax_aphrodite", 1920, 1080, 4, 1024),
           ("glass", 1920, 1080, 12, 1024), ("veach", 3840, 2160, 4, 256)]
ARMS = [("split", abi.PIPELINE_WAVEFRONT_SPLIT, {}), ("paired", abi.PIPELINE_WAVEFRONT_PAIRED, {}), ("mega", abi.PIPELINE_MEGAKERNEL, {}),
        ("split, every mesh walked", abi.PIPELINE_WAVEFRONT_SPLIT, {"walk_min_tris": 0, "small_mesh_bytes": 0}),
        ("paired, every mesh walked", abi.PIPELINE_WAVEFRONT_PAIRED, {"walk_min_tris": 0, "small_mesh_bytes": 0}),
        ("split, scene level by k_swalk", abi.PIPELINE_WAVEFRONT_SPLIT, {"flat_scan": 0}),
        ("split, meshes inline (no k_walk)", abi.PIPELINE_WAVEFRONT_SPLIT, {"walk": 0}),
        ("auto, arena in HBM", abi.PIPELINE_AUTO, {"lds_scene": 0})]
for name, W, H, depth, spp in CONFIGS:
    pack = os.path.join(G, name + ".pack")
    if not os.path.exists(pack):
        continue
    spp = max(1, int(spp*scale))
    scene = tinsel_amd.Scene.load_pack(pack)
    cam, opt = scene.camera, scene.options.copy()
    opt.width, opt.height, opt.max_depth, opt.mode = W, H, depth, abi.MODE_PATHTRACE
    def render(pipe, tune):
        r = tinsel_amd.create_gpu_renderer(scene, 0, abi.Tuning(**tune) if tune else None)
        r.set_pipeline(pipe)
        r.init(W, H)
        t0 = time.perf_counter()
        out = r.render(cam, opt, passes=spp)
        dt = time.perf_counter() - t0
        r.close()
        return out, dt
    want, dt = render(abi.PIPELINE_AUTO, {})
    print("%s %dx%d depth %d spp %d (%.2e paths): auto %.1f s" % (name, W, H, depth, spp, W*H*spp, dt), flush=True)
    for label, pipe, tune in ARMS:
        if pipe == abi.PIPELINE_MEGAKERNEL and W*H*spp > 1.2e9 and scale >= 1.0:
            s2 = spp   # (the megakernel is slow but it is the arm least like the others: keep it)
        out, dt = render(pipe, tune)
        bad = (out != want).any(axis=-1)
        print("    %-34s %6.1f s: %d pixels differ%s" % (label, dt, int(bad.sum()), "" if not bad.any() else "  first " + str(list(zip(*[a.tolist() for a in np.nonzero(bad)]))[:4])), flush=True)
