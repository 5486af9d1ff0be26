"""Child process of tests/test_gpu_switches.py: renders a few fixtures through both wavefront pipelines with the environment it was started in (the
library reads most of its A/B switches once per process) and compares per-path radiance and the framebuffer with the golden files, bit for bit.
Prints one line per (fixture, pipeline): `ok` or the number of paths that differ; exit status 1 on any difference."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tinsel_amd import Scene, abi, create_gpu_renderer  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
FIXTURES = sys.argv[1].split(",") if len(sys.argv) > 1 else ["cornell", "veach", "glass", "features", "ajax_standin_96", "many_spheres"]
PIPELINES = [("wavefront", abi.PIPELINE_WAVEFRONT), ("split", abi.PIPELINE_WAVEFRONT_SPLIT)]      # (the default pipeline is one of the two)


def main():
    bad = 0
    for name in FIXTURES:
        g = np.load(os.path.join(GOLDEN, name + ".golden.npz"))
        scene = Scene.load_pack(os.path.join(GOLDEN, name + ".pack"))
        cam = abi.Camera.from_buffer_copy(g["camera"].tobytes())
        opt = abi.Options.from_buffer_copy(g["options"].tobytes())
        passes = int(g["passes"])
        for label, pipe in PIPELINES:
            r = create_gpu_renderer(scene)
            r.set_pipeline(pipe)
            r.init(opt.width, opt.height)
            out = r.render(cam, opt, passes=passes)
            rad = r.batch_radiance(passes, opt.height, opt.width)
            r.close()
            diff = int((rad != g["radiance"]).any(axis=-1).sum())
            same = diff == 0 and np.array_equal(out, g["accum"])
            print("%s/%s: %s" % (name, label, "ok" if same else "%d paths differ, framebuffer %s" % (diff, np.array_equal(out, g["accum"]))), flush=True)
            bad += 0 if same else 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
