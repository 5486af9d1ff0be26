"""Headless stand-in for the caller of the boundary -- the role src/main.cpp plays behind GLUT.

    python -m tinsel_amd.headless [-spp=N] [-width=W] [-height=H] [-exposure=E] [-maxdepth=D]
                                  [-nlm=RADIUS[,FALLOFF]] [-rr=BOUNCE] [-out=image.png|image.pfm] [-save=state.npz] [-resume=state.npz]
                                  scene.pack

Conventions kept from main.cpp:
  * the LAST argument is the input file (main.cpp:97-101); here a scene pack written by the reference's own loader +
    Scene::Build (tests/golden/make_golden.py; .tin parsing is the reference's loader and stays there -- the C++ shim
    shim/tinsel_headless.cpp takes .tin directly);
  * `-key=value` overrides are applied after the scene's own options (main.cpp:143-149), `-spp` sets
    options.maxSamples;
  * a frame = 16 calls of Renderer::Render, one more sample per pixel each (main.cpp:242-250), then the display
    stage (normalise, ToneMap, LinearToSrgb, optional NonLocalMeansFilter: main.cpp:258-282) and one progress line
    "<samples> render: (ms) total: (ms)" (main.cpp:303);
  * when the sample count reaches maxSamples the image is written with WritePng's conversion (main.cpp:307-312).
  * BATCH / animation mode (main.cpp:104-118, 314-327): a `%` in the file name makes it a printf pattern over the frame index 0, 1, 2, ...;
    every frame renders maxSamples and is written to `<frame file>.png` (or `-out=pattern-with-%d`), until a frame's file does not exist.
    The reference deletes its renderer and re-runs Init per frame -- loader, Scene::Build, a new GpuRenderer: every mesh uploaded again.
    Here ONE renderer lives through the batch: where frame k + 1 is frame k with other primitive transforms (a rigid animation) the moved
    primitives' records are rewritten in place and the scene level rebuilt from the frame's own nodes (HipRenderer.update_scene:
    tinsel_hip_set_primitive_transform + tinsel_hip_rebuild_scene) -- the frames' PNGs are those of fresh renderers byte for byte
    (tests/test_gpu_display.py) -- and only a frame that differs in more is re-created.  Each frame prints what it cost to get ready.
Beyond it: `.pfm` output of the normalised linear image (PfmSave layout), and -save / -resume of the accumulator
(tinsel_hip_write_accum) so a long render can be continued bit-exactly.

No CPU fallback: without a GPU and the HIP library this exits with the library's error.
"""
import os
import sys
import time

import numpy as np

from . import abi
from .display import write_pfm, write_png
from .renderer import Scene, create_gpu_renderer

FRAME_PASSES = 16          # numSamples of main.cpp:240


def parse_args(argv):
    if len(argv) < 2:
        raise SystemExit(__doc__)
    cfg = {"file": argv[-1], "out": None, "nlm": 0, "nlm_falloff": 200.0, "save": None, "resume": None, "over": {}}
    for a in argv[1:-1]:
        if not a.startswith("-") or "=" not in a:
            raise SystemExit("unrecognised argument %r\n%s" % (a, __doc__))
        k, v = a[1:].split("=", 1)
        if k in ("spp", "width", "height", "maxdepth", "rr"):
            cfg["over"][k] = int(v)
        elif k == "exposure":
            cfg["over"][k] = float(v)
        elif k == "nlm":
            parts = v.split(",")
            cfg["nlm"] = int(parts[0])
            if len(parts) > 1:
                cfg["nlm_falloff"] = float(parts[1])
        elif k in ("out", "save", "resume"):
            cfg[k] = v
        else:
            raise SystemExit("unrecognised option -%s\n%s" % (k, __doc__))
    return cfg


def apply_overrides(scene, over):
    cam, opt = scene.camera, scene.options
    if "spp" in over:
        opt.max_samples = over["spp"]
    elif opt.max_samples >= 2**31 - 1:
        # the interactive reference renders until closed (maxSamples = INT_MAX, main.cpp:189); a batch run needs an end
        opt.max_samples = 64
        print("no -spp given and the scene sets no sample limit: rendering 64 spp")
    opt.width = over.get("width", opt.width)
    opt.height = over.get("height", opt.height)
    opt.max_depth = over.get("maxdepth", opt.max_depth)
    opt.exposure = over.get("exposure", opt.exposure)
    return cam, opt


def render_frame(r, cam, opt, cfg, samples=0):
    """the progressive loop of main.cpp:242-303 up to maxSamples; returns (presented image, samples)"""
    image = None
    while samples < opt.max_samples:
        ts = time.perf_counter()
        n = min(FRAME_PASSES, opt.max_samples - samples)
        r.render(cam, opt, passes=n, readback=False)
        tr = time.perf_counter()
        image = r.present(opt, cfg["nlm"], cfg["nlm_falloff"])
        samples += n
        te = time.perf_counter()
        print("%d render: (%.4fms) total: (%.4fms)" % (samples, (tr - ts)*1000.0, (te - ts)*1000.0), flush=True)
    if image is None:
        image = r.present(opt, cfg["nlm"], cfg["nlm_falloff"])
    return image, samples


def batch(cfg):
    """main.cpp's batch mode (:104-118, :314-327) with ONE renderer for the whole animation where the frames allow it."""
    over = cfg["over"]
    r, prev, index, ready_ms = None, None, 0, []
    while True:
        name = cfg["file"] % index
        if not os.path.exists(name):
            if index == 0:
                raise SystemExit("Couldn't open %s for reading." % name)       # (main.cpp:131-135)
            break
        t0 = time.perf_counter()
        scene = Scene.load_pack(name)
        cam, opt = apply_overrides(scene, over)
        how = "created"
        if r is not None and r.update_scene(prev, scene):
            how = "updated in place"
        else:
            if r is not None:
                r.close()
                how = "re-created (the frame differs in more than transforms)"
            r = create_gpu_renderer(scene)
            if over.get("rr", 0) > 0:
                r.set_russian_roulette(over["rr"])
        r.init(opt.width, opt.height)
        r.set_pass_index(0)             # every frame starts its seeds where a fresh renderer would
        ms = (time.perf_counter() - t0)*1000.0
        ready_ms.append((how, ms))
        print("frame %d: %s: renderer %s in %.3fms" % (index, name, how, ms), flush=True)
        image, _ = render_frame(r, cam, opt, cfg)
        out = (cfg["out"] % index) if (cfg["out"] and "%" in cfg["out"]) else name + ".png"      # (main.cpp:113-115: input + ".png")
        write_png(out, image)
        print("wrote %s" % out, flush=True)
        prev = scene
        index += 1
    if r is not None:
        r.close()
    inplace = [ms for how, ms in ready_ms[1:] if how == "updated in place"]
    print("%d frames; first renderer ready in %.3fms%s" % (index, ready_ms[0][1],
          "; %d later frames updated in place in %.3fms on average (the reference re-creates: the first frame's cost every time)" % (
              len(inplace), sum(inplace)/len(inplace)) if inplace else ""))
    return 0


def main(argv=None):
    cfg = parse_args(sys.argv if argv is None else argv)
    if "%" in cfg["file"]:
        return batch(cfg)
    t0 = time.perf_counter()
    scene = Scene.load_pack(cfg["file"])
    cam, opt = apply_overrides(scene, cfg["over"])
    over = cfg["over"]

    r = create_gpu_renderer(scene)
    if over.get("rr", 0) > 0:
        r.set_russian_roulette(over["rr"])       # opt-in; not the reference's behaviour (tinsel_hip.h)
    r.init(opt.width, opt.height)
    print("Created renderer in %fms" % ((time.perf_counter() - t0)*1000.0))

    samples = 0
    if cfg["resume"]:
        st = np.load(cfg["resume"])
        samples = int(st["samples"])
        r.write_accum(st["accum"], samples)

    image, samples = render_frame(r, cam, opt, cfg, samples)

    if cfg["save"]:
        np.savez(cfg["save"], accum=r.read_accum(), samples=np.int64(samples))
    if cfg["out"]:
        if cfg["out"].endswith(".pfm"):
            a = r.read_accum()
            with np.errstate(all="ignore"):
                write_pfm(cfg["out"], a[..., :3]/a[..., 3:4])
        else:
            write_png(cfg["out"], image)
        print("wrote %s" % cfg["out"])
    st = r.stats()
    print("%d samples, %d rays" % (st["samples"], st["rays"]))
    r.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
