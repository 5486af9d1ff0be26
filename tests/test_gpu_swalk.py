"""k_swalk (tinsel_amd/csrc/tn_swalk.h): the scene-level walk with ray replacement (QueryBVH under Trace, reference
intersection.h:751-799 / render.cpp:17-62), which replaces k_extend / k_shadow where the scene does not fit the wave-uniform flat
scan.  By default that is many_spheres (203 primitives) only; with the flat scan switched off (tinsel_hip_tuning::flat_scan = 0, the
library's A/B knob) EVERY fixture and the 32-scene fuzz corpus go through it in the split pipeline: planes, spheres, meshes
walked inline on the stack above the scene level, moving primitives, several shadow rays per bounce, probes, one-primitive
scenes whose root is a leaf.  Per-path radiance and framebuffer must be the reference's bit for bit."""
import os

import numpy as np
import pytest

from tinsel_amd import abi
from tests import oracle_api as oa
from tests.test_gpu_parity import SCENES, _load

pytestmark = pytest.mark.gpu


@pytest.fixture
def no_flat_scan(monkeypatch):
    from tinsel_amd import renderer
    monkeypatch.setattr(renderer, "DEFAULT_TUNING", abi.Tuning(flat_scan=0))


def _render_split(scene, cam, opt, passes, first_pass=0):
    from tinsel_amd import create_gpu_renderer
    r = create_gpu_renderer(scene)
    r.set_pipeline(abi.PIPELINE_WAVEFRONT_SPLIT)
    r.enable_kernel_timing(True)
    r.init(opt.width, opt.height)
    r.set_pass_index(first_pass)
    out = r.render(cam, opt, passes=passes)
    rad = r.batch_radiance(passes, opt.height, opt.width)
    st = r.stats()
    r.close()
    return out, rad, st


@pytest.mark.parametrize("name", SCENES)
def test_every_fixture_through_the_scene_walk_matches_the_reference(name, no_flat_scan):
    scene, cam, opt, g = _load(name)
    passes = int(g["passes"])
    out, rad, st = _render_split(scene, cam, opt, passes)
    assert st["samples"] == passes*opt.width*opt.height
    assert np.array_equal(rad, g["radiance"]), "%s: %d paths differ" % (name, int((rad != g["radiance"]).any(axis=-1).sum()))
    assert np.array_equal(out, g["accum"])


def test_scene_walk_counts_the_rays_of_the_inline_kernels(monkeypatch):
    """many_spheres as shipped: k_swalk (default) against k_extend / k_shadow (tinsel_hip_tuning::scene_walk = 0) -- same image, same rays."""
    scene, cam, opt, g = _load("many_spheres")
    out_a, rad_a, st_a = _render_split(scene, cam, opt, 3)
    from tinsel_amd import renderer
    monkeypatch.setattr(renderer, "DEFAULT_TUNING", abi.Tuning(scene_walk=0))
    out_b, rad_b, st_b = _render_split(scene, cam, opt, 3)
    assert np.array_equal(out_a, out_b) and np.array_equal(rad_a, rad_b)
    assert st_a["rays"] == st_b["rays"] and st_a["shadow_rays"] == st_b["shadow_rays"]


def test_fuzz_corpus_through_the_scene_walk(no_flat_scan):
    import tinsel_amd
    corpus = np.load(os.path.join(oa.GOLDEN, "fuzz.golden.npz"))
    bad = []
    for k in range(int(corpus["count"])):
        scene = tinsel_amd.Scene(corpus["pack_%02d" % k].tobytes())
        out, rad, _ = _render_split(scene, scene.camera, scene.options, 2, int(corpus["first_pass_%02d" % k]))
        if not np.array_equal(rad, corpus["radiance_%02d" % k]) or not np.array_equal(out, corpus["accum_%02d" % k]):
            bad.append(k)
    assert not bad, "scenes that differ through k_swalk: %s" % bad
