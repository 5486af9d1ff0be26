"""GPU parity of the display stage: tinsel_hip_present (normalise + ToneMap + LinearToSrgb
+ NonLocalMeansFilter kernels), the headless frame loop, and accumulator save / resume -- against
tests/golden/display.golden.npz (outputs of the reference's own code) and, when oracle/_ref travelled, the
reference run live on the same accumulator.  Everything is bit-identical, down to the bytes of the PNG file."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from tinsel_amd import abi
from tests.oracle_api import GOLDEN, png_pixels
from tests.test_display import same_bits

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAVE_REF = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libtinsel_ref.so"))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "display.golden.npz"))


def _renderer(width, height):
    from tinsel_amd import Scene, create_gpu_renderer
    scene = Scene.load_pack(os.path.join(GOLDEN, "cornell.pack"))
    r = create_gpu_renderer(scene)
    r.init(width, height)
    return scene, r


@pytest.mark.parametrize("tag", ["a", "b"])
def test_present_matches_reference(gold, tag):
    accum = gold["accum_" + tag]
    scene, r = _renderer(accum.shape[1], accum.shape[0])
    r.write_accum(accum, 0)
    opt = scene.options.copy()
    opt.width, opt.height = accum.shape[1], accum.shape[0]
    opt.exposure = float(gold["exposure_" + tag])
    out = r.present(opt)
    assert same_bits(out, gold["filtered_" + tag])
    # eNormals / eComplexity present the raw pixels (main.cpp:258)
    opt.mode = abi.MODE_NORMALS
    raw = r.present(opt)
    assert same_bits(raw, accum)
    r.close()


def test_nlm_matches_reference(gold):
    accum = gold["accum_a"]
    scene, r = _renderer(accum.shape[1], accum.shape[0])
    r.write_accum(accum, 0)
    opt = scene.options.copy()
    opt.width, opt.height = accum.shape[1], accum.shape[0]
    assert np.array_equal(gold["nlm_in_a"], gold["filtered_a"])
    assert same_bits(r.present(opt, nlm_width=1, nlm_falloff=200.0), gold["nlm_r1_a"])
    assert same_bits(r.present(opt, nlm_width=2, nlm_falloff=50.0), gold["nlm_r2_a"])
    assert same_bits(r.present(opt), gold["filtered_a"])        # and back to unfiltered
    r.close()


def test_display_leaf_functions_are_host_libm_bit_for_bit(tmp_path):
    """powf(x, 2.2), powf(x, 1/2.2) over the whole non-negative float range and expf over [-110, 90] on the device
    equal THIS host's libm to the last bit (8 M samples + specials)."""
    scene, r = _renderer(16, 16)
    rng = np.random.default_rng(3)
    n = 8_000_000
    bits = rng.integers(0, 0x7f800000, n, dtype=np.uint32)
    bits[:8] = [0, 1, 0x007fffff, 0x00800000, 0x3f800000, 0x7f7fffff, 0x7f800000, 0x3b83126f]
    x = bits.view(np.float32).copy()
    x[8:12] = [-0.0, -1.5, np.nan, -np.inf]
    y = (rng.random(n)*200.0 - 110.0).astype(np.float32)
    y[:6] = [0.0, -np.inf, np.inf, np.nan, 88.72284, -103.972084]
    out = r.leaf(8, 0, n, 4, rows=np.stack([x, y], axis=1))
    src = tmp_path / "l.c"
    src.write_text("#include <math.h>\nvoid f(int n,const float*x,const float*y,float*o){for(int i=0;i<n;i++){o[3*i]=powf(x[i],2.2f);"
                   "o[3*i+1]=powf(x[i],1.0f/2.2f);o[3*i+2]=expf(y[i]);}}")
    so = tmp_path / "l.so"
    subprocess.run(["gcc", "-O1", "-fno-builtin", "-shared", "-fPIC", "-o", str(so), str(src), "-lm"], check=True)
    L = C.CDLL(str(so))
    ref = np.zeros((n, 3), np.float32)
    L.f(n, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), ref.ctypes.data_as(C.c_void_p))
    for k, name in enumerate(["powf(x,2.2)", "powf(x,1/2.2)", "expf"]):
        assert same_bits(out[:, k], ref[:, k]), name
    r.close()


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref not built")
def test_frame_loop_against_live_reference(tmp_path):
    """Render -> present -> NLM -> PNG, against the reference's PathTrace + AddSample + ToneMap/LinearToSrgb +
    NonLocalMeansFilter + WritePng run here on the host: same float images, same file bytes."""
    from tests.oracle_api import RefOracle
    from tinsel_amd.display import png_bytes, quantize_rgb8
    R = RefOracle()
    W, H, spp = 160, 120, 8
    scene, r = _renderer(W, H)
    cam, opt = scene.camera, scene.options.copy()
    opt.width, opt.height, opt.exposure = W, H, 1.3
    h = R.load_pack(os.path.join(GOLDEN, "cornell.pack"))
    ref_accum, _, _ = R.render_seeded(h, cam, opt, 0, spp)
    R.free(h)
    ref_filtered = R.present(ref_accum, opt.exposure, opt.limit)
    ref_nlm = R.nlm(ref_filtered, 200.0, 1)
    R.write_png(ref_nlm, str(tmp_path / "ref.png"))

    accum = r.render(cam, opt, passes=spp)
    assert np.array_equal(accum, ref_accum)
    assert same_bits(r.present(opt), ref_filtered)
    mine = r.present(opt, nlm_width=1, nlm_falloff=200.0)
    assert same_bits(mine, ref_nlm)
    assert png_bytes(quantize_rgb8(mine)) == open(tmp_path / "ref.png", "rb").read()
    r.close()


def _headless(*args):
    env = dict(os.environ, PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, "-m", "tinsel_amd.headless"] + list(args), cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    return p.stdout


def test_headless_cli_and_resume(tmp_path, gold):
    """The main.cpp-style driver end to end: CLI overrides, 16-pass frames, PNG out; a render interrupted at 16 spp,
    saved and resumed to 40 spp writes the same file as the uninterrupted one."""
    pack = os.path.join(GOLDEN, "cornell.pack")
    full = tmp_path / "full.png"
    out = _headless("-spp=40", "-width=96", "-height=64", "-maxdepth=3", "-exposure=0.8", "-out=%s" % full, pack)
    lines = [l for l in out.splitlines() if " render: (" in l]
    assert [int(l.split()[0]) for l in lines] == [16, 32, 40]          # frames of 16 passes (main.cpp:240-250)
    _headless("-spp=16", "-width=96", "-height=64", "-maxdepth=3", "-exposure=0.8", "-save=%s" % (tmp_path / "s.npz"), pack)
    part = tmp_path / "part.png"
    _headless("-spp=40", "-width=96", "-height=64", "-maxdepth=3", "-exposure=0.8", "-resume=%s" % (tmp_path / "s.npz"),
              "-out=%s" % part, pack)
    assert open(full, "rb").read() == open(part, "rb").read()
    px = png_pixels(open(full, "rb").read())
    assert px.shape == (64, 96, 3) and px.mean() > 20

    if HAVE_REF:
        from tests.oracle_api import RefOracle
        from tinsel_amd import Scene
        R = RefOracle()
        scene = Scene.load_pack(pack)
        cam, opt = scene.camera, scene.options.copy()
        opt.width, opt.height, opt.max_depth, opt.exposure = 96, 64, 3, 0.8
        h = R.load_pack(pack)
        ref_accum, _, _ = R.render_seeded(h, cam, opt, 0, 40)
        R.free(h)
        R.write_png(R.present(ref_accum, 0.8, opt.limit), str(tmp_path / "ref.png"))
        assert open(full, "rb").read() == open(tmp_path / "ref.png", "rb").read()

    pfm = tmp_path / "x.pfm"
    _headless("-spp=4", "-width=32", "-height=16", "-out=%s" % pfm, pack)
    data = open(pfm, "rb").read()
    assert data.startswith(b"PF\n32 16\n-") and len(data) == data.index(b"\n", 10) + 1 + 32*16*12


def test_headless_batch_mode_one_renderer_for_an_animation(tmp_path):
    """main.cpp's batch / animation mode (:104-118 a `%d` file name, :314-327 PNG per frame, renderer deleted and re-created per frame) the
    MI355X way: ONE renderer for the batch, frames that differ in primitive transforms only are updated in place
    (tinsel_hip_set_primitive_transform + tinsel_hip_rebuild_scene from the frame's own nodes).  Four frames of a cornell box whose sphere moves
    DURING each exposure and whose light drifts (tests/golden/make_animation.py: the reference's own Scene::Build per frame), then a fifth
    frame that is another scene altogether: every PNG must be, byte for byte, the file a fresh renderer writes for that frame alone."""
    import shutil
    for k in range(4):
        shutil.copy(os.path.join(GOLDEN, "anim_cornell_%d.pack" % k), tmp_path / ("anim_%d.pack" % k))
    shutil.copy(os.path.join(GOLDEN, "veach.pack"), tmp_path / "anim_4.pack")
    args = ["-spp=24", "-width=96", "-height=64"]
    out = _headless(*args, str(tmp_path / "anim_%d.pack"))
    ready = [l for l in out.splitlines() if l.startswith("frame ")]
    assert len(ready) == 5, out
    assert "renderer created" in ready[0] and all("updated in place" in l for l in ready[1:4]) and "re-created" in ready[4], ready
    assert "5 frames" in out and "3 later frames updated in place" in out
    frames = []
    for k in range(5):
        batch_png = tmp_path / ("anim_%d.pack.png" % k)                 # (main.cpp:113-115: input file + ".png")
        fresh_png = tmp_path / ("fresh_%d.png" % k)
        _headless(*args, "-out=%s" % fresh_png, str(tmp_path / ("anim_%d.pack" % k)))
        a, b = open(batch_png, "rb").read(), open(fresh_png, "rb").read()
        assert a == b, "frame %d: the batch's PNG differs from a fresh renderer's" % k
        frames.append(a)
    assert len(set(frames)) == 5                                        # (the frames really differ)
    # -out with a %d: the caller's own naming
    _headless(*args, "-out=%s" % (tmp_path / "f%02d.png"), str(tmp_path / "anim_%d.pack"))
    assert open(tmp_path / "f02.png", "rb").read() == frames[2]
