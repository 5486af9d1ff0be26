"""Child process of tests/test_gpu_switches.py: renders a few fixtures through both wavefront pipelines with the tinsel_hip_tuning given on the
command line (--tuning '{"field": value, ...}': create-time fields go to tinsel_hip_create_tuned, the rest through tinsel_hip_set_tuning as
well, so both entry points are driven) and compares per-path radiance and the framebuffer with the golden files, bit for bit.
Prints one line per (fixture, pipeline): `ok` or the number of paths that differ; exit status 1 on any difference.
--describe a,b: prints what the library decided for fixtures a, b with the DEFAULT tuning (pipeline facts + an image digest)."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tinsel_amd import Scene, abi, create_gpu_renderer  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
DEFAULT_FIXTURES = ["cornell", "veach", "glass", "features", "ajax_standin_96", "many_spheres"]
PIPELINES = [("wavefront", abi.PIPELINE_WAVEFRONT), ("split", abi.PIPELINE_WAVEFRONT_SPLIT)]      # (the default pipeline is one of the two)


def load(name):
    g = np.load(os.path.join(GOLDEN, name + ".golden.npz"))
    scene = Scene.load_pack(os.path.join(GOLDEN, name + ".pack"))
    cam = abi.Camera.from_buffer_copy(g["camera"].tobytes())
    opt = abi.Options.from_buffer_copy(g["options"].tobytes())
    return g, scene, cam, opt, int(g["passes"])


def describe(fixtures):
    for name in fixtures:
        g, scene, cam, opt, passes = load(name)
        r = create_gpu_renderer(scene)
        r.init(opt.width, opt.height)
        out = r.render(cam, opt, passes=passes)
        alive, nee = r.queue_counts()
        print("describe %s walked=%d stack=%d nee=%d tuning=%s alive=%s image=%s" % (
            name, r.walked_prims, r.stack_entries, r.nee_per_path, json.dumps(r.get_tuning().as_dict(), sort_keys=True), alive,
            hashlib.sha1(out.tobytes()).hexdigest()), flush=True)
        r.close()
    return 0


def main():
    args = sys.argv[1:]
    if args[:1] == ["--describe"]:
        return describe(args[1].split(","))
    fields, fixtures = {}, DEFAULT_FIXTURES
    while args:
        a = args.pop(0)
        if a == "--tuning":
            fields = json.loads(args.pop(0))
        else:
            fixtures = a.split(",")
    tuning = abi.Tuning(**fields)
    per_render = {k: v for k, v in fields.items() if k not in abi.Tuning.CREATE_FIELDS}
    bad = 0
    for name in fixtures:
        g, scene, cam, opt, passes = load(name)
        for label, pipe in PIPELINES:
            if label == "wavefront":
                r = create_gpu_renderer(scene, 0, tuning)                    # everything through create
            else:
                r = create_gpu_renderer(scene, 0, abi.Tuning(**{k: v for k, v in fields.items() if k in abi.Tuning.CREATE_FIELDS}))
                r.set_tuning(abi.Tuning(**per_render))                      # the per-render part through the setter
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
