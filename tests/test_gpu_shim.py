"""Drop-in check on the GPU box: the reference's OWN loader / Scene::Build / Renderer interface driving
CreateGpuRenderer() through shim/hip_renderer.cpp (binary built where the reference is mounted)."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "shim", "_build", "tinsel_headless")


@pytest.mark.skipif(not os.path.exists(EXE), reason="shim/_build/tinsel_headless not built (needs /root/reference)")
def test_reference_main_loop_drives_the_hip_backend():
    scene = os.path.join(ROOT, "tests", "golden", "scenes", "features.tin")
    out = subprocess.run([EXE, scene, "-spp=256", "-cpuspp=64", "-width=96", "-height=64"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"mean radiance gpu ([\d.]+) cpu ([\d.]+) ; per-pixel L2 between the two estimates ([\d.e+-]+)", out.stdout)
    assert m, out.stdout
    gpu, cpu, l2 = float(m.group(1)), float(m.group(2)), float(m.group(3))
    # two independent Monte-Carlo estimates of the same image (different RNG streams): means agree to ~1 %
    assert abs(gpu/cpu - 1) < 0.03, out.stdout
    assert l2 < 0.5


@pytest.mark.skipif(not os.path.exists(EXE), reason="shim/_build/tinsel_headless not built (needs /root/reference)")
def test_cxx_caller_writes_the_same_png_as_the_python_caller(tmp_path):
    """The C++ caller (reference loader + Scene::Build + shim + the device display stage + the reference's OWN
    WritePng) and the Python caller (scene pack + tinsel_amd.display's writer) produce byte-identical files."""
    import sys
    a, b = tmp_path / "cxx.png", tmp_path / "py.png"
    tin = os.path.join(ROOT, "tests", "golden", "scenes", "features.tin")
    out = subprocess.run([EXE, tin, "-spp=24", "-width=96", "-height=64", "-png=%s" % a], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and os.path.exists(a), out.stdout + out.stderr
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-m", "tinsel_amd.headless", "-spp=24", "-width=96", "-height=64", "-out=%s" % b,
                          os.path.join(ROOT, "tests", "golden", "features.pack")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert open(a, "rb").read() == open(b, "rb").read()


@pytest.mark.skipif(not os.path.exists(EXE), reason="shim/_build/tinsel_headless not built (needs /root/reference)")
def test_cxx_batch_mode_updates_one_renderer_in_place(tmp_path):
    """The reference's batch mode (main.cpp:104-118, 314-327) through the C++ caller: frame%d.tin, the reference's own loader + Scene::Build per
    frame, ONE HipRenderer updated in place where only transforms changed (HipRendererUpdateScene), the reference's own WritePng.  Frames: this
    repo's features.tin with its moving glass sphere further along every frame (and moving during the exposure); a last frame with another
    material forces the re-create path.  Every PNG equals, byte for byte, the one a run on that frame alone writes."""
    base = open(os.path.join(ROOT, "tests", "golden", "scenes", "features.tin")).read()
    moving = "position -1.3 0.5 0.2 , -0.9 0.7 0.4"
    assert base.count(moving) == 1
    for k in range(4):
        x0, x1 = -1.3 + 0.4*k, -0.9 + 0.4*k
        text = base.replace(moving, "position %.2f 0.5 0.2 , %.2f 0.7 0.4" % (x0, x1))
        if k == 3:
            assert text.count("roughness 0.15") == 1
            text = text.replace("roughness 0.15", "roughness 0.35")         # (the gold material: more than a transform)
        (tmp_path / ("f%d.tin" % k)).write_text(text)
    args = ["-spp=24", "-width=96", "-height=64"]
    out = subprocess.run([EXE, str(tmp_path / "f%d.tin")] + args, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    ready = [l for l in out.stdout.splitlines() if l.startswith("frame ")]
    assert len(ready) == 4 and "renderer created" in ready[0] and "updated in place" in ready[1] and "updated in place" in ready[2] and "re-created" in ready[3], "\n".join(ready) + out.stderr[-1500:]
    pngs = []
    for k in range(4):
        alone = tmp_path / ("alone%d.png" % k)
        one = subprocess.run([EXE, str(tmp_path / ("f%d.tin" % k))] + args + ["-png=%s" % alone], capture_output=True, text=True, timeout=300)
        assert one.returncode == 0, one.stdout + one.stderr
        a, b = open(tmp_path / ("f%d.tin.png" % k), "rb").read(), open(alone, "rb").read()
        assert a == b, "frame %d" % k
        pngs.append(a)
    assert len(set(pngs)) == 4
