"""tinsel_hip_group (include/tinsel_hip.h): N GPUs behind ONE Renderer inside the library -- host thread per device,
pixel-tile shards with rank-local path slots, one reduce of the float4 accumulator per read-back, then D2H.

This pool's GPU boxes have ONE device, so the multi-member code runs under the library's validation switch
TINSEL_HIP_GROUP_ONE_DEVICE=1 (all members on device 0; the reduce is a device-local sum in rank order instead of
ncclReduce -- everything else, threads included, is the real path).  On a box with >= 2 devices the same tests run over
RCCL (the switch is only set when fewer devices are visible than members)."""
import os
import re
import subprocess

import numpy as np
import pytest

from tinsel_amd import abi
from tests.test_gpu_parity import _load, _render

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _group(scene, n, tile, monkeypatch):
    import torch
    from tinsel_amd import HipRendererGroup
    if torch.cuda.device_count() < n:
        monkeypatch.setenv("TINSEL_HIP_GROUP_ONE_DEVICE", "1")
    return HipRendererGroup(scene, n, tile)


@pytest.mark.parametrize("n,tile", [(2, 64), (3, 16), (8, 32)])
def test_group_image_equals_the_single_gpu_image(n, tile, monkeypatch):
    scene, cam, opt, g = _load("features")
    whole, st = _render(scene, cam, opt, 3, abi.PIPELINE_AUTO)
    grp = _group(scene, n, tile, monkeypatch)
    assert grp.num_gpus == n
    grp.init(opt.width, opt.height)
    out = grp.render(cam, opt, passes=3)
    samples = sum(grp.member_stats(k)["samples"] for k in range(n))
    rays = sum(grp.member_stats(k)["rays"] for k in range(n))
    grp.close()
    assert samples == 3*opt.width*opt.height and rays == st["rays"]        # every path traced exactly once, by exactly one member
    np.testing.assert_allclose(out, whole, rtol=1e-5, atol=1e-6)             # same terms per pixel, grouped by member


def test_group_render_is_progressive_and_never_double_counts(monkeypatch):
    """Render x 3 with a read-back each time == one Render of 3 passes: the reduce leaves the members' accumulators alone."""
    scene, cam, opt, g = _load("cornell")
    grp = _group(scene, 2, 32, monkeypatch)
    grp.init(opt.width, opt.height)
    for _ in range(3):
        out = grp.render(cam, opt, passes=1)
    grp.init(opt.width, opt.height)          # Init zeroes: the next frame starts from nothing ...
    grp2 = grp.render(cam, opt, passes=1)
    grp.close()
    whole, _ = _render(scene, cam, opt, 3, abi.PIPELINE_AUTO)
    np.testing.assert_allclose(out, whole, rtol=1e-5, atol=1e-6)
    assert grp2[..., 3].max() < 0.5*whole[..., 3].max()        # ... (one pass of weight, not four)


@pytest.mark.parametrize("n,tile", [(2, 64), (3, 16)])
def test_group_normals_mode_is_the_image_not_n_times_it(n, tile, monkeypatch):
    """eNormals has no weights to divide by (main.cpp:256 shows the buffer as it is): every member writes its own pixels and
    zero elsewhere, so the group's sum is the reference's image exactly (x + 0 + 0 is x)."""
    scene, cam, opt, g = _load("features")
    nopt = opt.copy()
    nopt.mode = abi.MODE_NORMALS
    grp = _group(scene, n, tile, monkeypatch)
    grp.init(opt.width, opt.height)
    out = grp.render(cam, nopt, passes=1)
    out2 = grp.render(cam, nopt, passes=1)        # overwrite semantics: a second call changes nothing
    grp.close()
    assert np.array_equal(out, g["normals"])
    assert np.array_equal(out2, g["normals"])


@pytest.mark.parametrize("n,mode", [(2, abi.LOOKAHEAD_ON), (3, abi.LOOKAHEAD_ON), (2, abi.LOOKAHEAD_PIN_OUTPUT), (3, abi.LOOKAHEAD_PIN_OUTPUT)])
def test_group_lookahead_changes_no_bit(n, mode, monkeypatch):
    """tinsel_hip_group_set_lookahead: the reference's call pattern (Render = 1 pass + full read-back) at N members -- every
    member traces batches of future calls of its shard, the next call's snapshots are reduced while this call's image is
    copied out.  Every returned image must equal the plain group's bit for bit: through hits, misses (options change
    mid-stream), a present in between, member introspection (drops the speculation) and an Init."""
    scene, cam, opt, g = _load("features")
    plain = _group(scene, n, 32, monkeypatch); plain.init(opt.width, opt.height)
    ahead = _group(scene, n, 32, monkeypatch); ahead.init(opt.width, opt.height)
    ahead.set_lookahead(mode)
    out = np.empty((opt.height, opt.width, 4), np.float32)
    deeper = opt.copy(); deeper.max_depth = opt.max_depth + 2
    script = [opt, opt, opt, deeper, deeper, opt, opt, opt, opt]          # hits, a miss, hits, a miss, hits
    for k, o in enumerate(script):
        want = plain.render(cam, o, passes=1)
        got = ahead.render(cam, o, output=out, passes=1)
        assert np.array_equal(got, want), "call %d: %d pixels differ, max abs difference %.3e" % (k, int((got != want).any(axis=-1).sum()), float(np.abs(got - want).max()))
        if k == 2:
            assert np.array_equal(ahead.present(o), plain.present(o))       # the committed sum only, speculation intact
        if k == 6:
            assert ahead.member_stats(0)["samples"] >= plain.member_stats(0)["samples"]     # (counters may run ahead)
    for _ in range(3):                                                      # two passes per call
        want = plain.render(cam, opt, passes=2)
        got = ahead.render(cam, opt, output=out, passes=2)
        assert np.array_equal(got, want)
    # a call without read-back in between, then a new frame
    plain.render(cam, opt, passes=1, readback=False); ahead.render(cam, opt, passes=1, readback=False)
    assert np.array_equal(ahead.render(cam, opt, output=out, passes=1), plain.render(cam, opt, passes=1))
    plain.init(opt.width, opt.height); ahead.init(opt.width, opt.height)
    assert np.array_equal(ahead.render(cam, opt, output=out, passes=1), plain.render(cam, opt, passes=1))
    assert np.array_equal(ahead.render(cam, opt, output=out, passes=1), plain.render(cam, opt, passes=1))
    plain.close(); ahead.close()


def test_group_lookahead_through_the_one_rank_rccl_arm(monkeypatch):
    """The look-ahead reduce itself (ncclReduce of a SNAPSHOT into totalNext from the member's thread, ordered behind the
    snapshot's event) with one participant: bit-identical to the reference."""
    from tinsel_amd import HipRendererGroup
    monkeypatch.delenv("TINSEL_HIP_GROUP_ONE_DEVICE", raising=False)
    monkeypatch.setenv("TINSEL_HIP_GROUP_FORCE_RCCL", "1")
    scene, cam, opt, g = _load("cornell")
    grp = HipRendererGroup(scene, 1)
    grp.init(opt.width, opt.height)
    grp.set_lookahead(abi.LOOKAHEAD_ON)
    out = np.empty((opt.height, opt.width, 4), np.float32)
    for _ in range(int(g["passes"])):
        grp.render(cam, opt, output=out, passes=1)
    grp.close()
    assert np.array_equal(out, g["accum"])


def test_group_of_one_is_one_renderer_bit_for_bit():
    from tinsel_amd import HipRendererGroup
    scene, cam, opt, g = _load("cornell")
    grp = HipRendererGroup(scene, 1)
    grp.init(opt.width, opt.height)
    out = grp.render(cam, opt, passes=int(g["passes"]))
    grp.close()
    assert np.array_equal(out, g["accum"])


def test_one_rank_rccl_reduce_executes(monkeypatch):
    """The RCCL arm itself on a single-GPU box: dlopen(librccl.so.1), ncclCommInitAll over one device and a 1-rank
    ncclReduce of the accumulator into the separate `total` on the member's stream, from the member's thread
    (TINSEL_HIP_GROUP_FORCE_RCCL: the calls a multi-GPU group makes, with one participant)."""
    from tinsel_amd import HipRendererGroup
    monkeypatch.delenv("TINSEL_HIP_GROUP_ONE_DEVICE", raising=False)
    monkeypatch.setenv("TINSEL_HIP_GROUP_FORCE_RCCL", "1")
    scene, cam, opt, g = _load("cornell")
    grp = HipRendererGroup(scene, 1)
    grp.init(opt.width, opt.height)
    half = int(g["passes"])//2
    grp.render(cam, opt, passes=half)
    out = grp.render(cam, opt, passes=int(g["passes"]) - half)
    grp.close()
    assert np.array_equal(out, g["accum"])          # a sum over one rank is a copy: still bit-identical to the reference


def test_group_refuses_more_gpus_than_visible(monkeypatch):
    import torch
    import tinsel_amd
    monkeypatch.delenv("TINSEL_HIP_GROUP_ONE_DEVICE", raising=False)
    scene, cam, opt, g = _load("one_sphere")
    with pytest.raises(tinsel_amd.TinselHipError, match="GPUs requested"):
        tinsel_amd.HipRendererGroup(scene, torch.cuda.device_count() + 1)


EXE = os.path.join(ROOT, "shim", "_build", "tinsel_headless")


@pytest.mark.skipif(not os.path.exists(EXE), reason="shim/_build/tinsel_headless not built (needs /root/reference)")
def test_reference_caller_gets_n_gpus_through_the_cxx_shim():
    """The reference's own loader + Renderer interface, CreateGpuRenderer() from shim/hip_renderer.cpp with 2 members
    (min(2, visible) real devices; on this pool's boxes the one-device validation switch): no Python in the loop."""
    import torch
    env = dict(os.environ, TINSEL_HIP_NUM_GPUS="2")
    if torch.cuda.device_count() < 2:
        env["TINSEL_HIP_GROUP_ONE_DEVICE"] = "1"
    scene = os.path.join(ROOT, "tests", "golden", "scenes", "features.tin")
    out = subprocess.run([EXE, scene, "-spp=256", "-cpuspp=64", "-width=96", "-height=64"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "CreateGpuRenderer: 2 GPUs" in out.stderr, out.stderr
    m = re.search(r"mean radiance gpu ([\d.]+) cpu ([\d.]+) ; per-pixel L2 between the two estimates ([\d.e+-]+)", out.stdout)
    assert m, out.stdout
    gpu, cpu = float(m.group(1)), float(m.group(2))
    assert abs(gpu/cpu - 1) < 0.03, out.stdout
    one = subprocess.run([EXE, scene, "-spp=256", "-cpuspp=1", "-width=96", "-height=64"], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, TINSEL_HIP_NUM_GPUS="1"))
    m1 = re.search(r"mean radiance gpu ([\d.]+)", one.stdout)
    assert m1 and abs(float(m1.group(1))/gpu - 1) < 1e-4, one.stdout + out.stdout     # same paths, summed in another order
