"""k_walk (tinsel_amd/csrc/tn_walk.h): the dedicated mesh-walk kernel of the split pipeline must give, bit for bit, the
closest hits of the inline IntersectRayMesh walk (reference intersection.h:661-749) -- i.e. the reference's radiance.

By default, in a scene that has a mesh too large for the LDS arena, every mesh of 8 triangles or more lives in HBM and is
handed to k_walk, which the committed fixtures exercise with one or two meshes and one NEE ray per bounce.  Here the thresholds are lowered through tinsel_hip_tuning (walk_min_tris, small_mesh_bytes) so that
EVERY mesh of EVERY fixture and of the 32-scene fuzz corpus goes through it: several walked primitives per scene,
several shadow rays per bounce, moving meshes, one-triangle trees, rays that miss the leaf box."""
import os

import numpy as np
import pytest

from tinsel_amd import abi
from tests import oracle_api as oa
from tests.test_gpu_parity import SCENES, _load

pytestmark = pytest.mark.gpu


@pytest.fixture
def walk_everything(monkeypatch):
    from tinsel_amd import renderer
    monkeypatch.setattr(renderer, "DEFAULT_TUNING", abi.Tuning(walk_min_tris=0, small_mesh_bytes=0))


def _render_split(scene, cam, opt, passes, first_pass=0):
    from tinsel_amd import create_gpu_renderer
    r = create_gpu_renderer(scene)
    walked = r.walked_prims
    r.set_pipeline(abi.PIPELINE_WAVEFRONT_SPLIT)
    r.init(opt.width, opt.height)
    r.set_pass_index(first_pass)
    out = r.render(cam, opt, passes=passes)
    rad = r.batch_radiance(passes, opt.height, opt.width)
    r.close()
    return out, rad, walked


@pytest.mark.parametrize("name", SCENES)
def test_every_mesh_through_k_walk_matches_the_reference(name, walk_everything):
    scene, cam, opt, g = _load(name)
    passes = int(g["passes"])
    out, rad, walked = _render_split(scene, cam, opt, passes)
    assert np.array_equal(rad, g["radiance"]), "%s: %d paths differ with %d walked primitives" % (
        name, int((rad != g["radiance"]).any(axis=-1).sum()), walked)
    assert np.array_equal(out, g["accum"])


def test_fixtures_cover_multi_mesh_and_multi_shadow_ray_walks(walk_everything):
    """The point of the lowered thresholds: some fixture must walk >= 2 primitives with >= 2 NEE rays per bounce."""
    from tinsel_amd import create_gpu_renderer
    seen = {}
    for name in SCENES:
        scene, cam, opt, g = _load(name)
        r = create_gpu_renderer(scene)
        seen[name] = (r.walked_prims, r.nee_per_path)
        r.close()
    assert any(w >= 2 and k >= 2 for w, k in seen.values()), seen
    assert any(w >= 1 for w, k in seen.values())


def test_default_thresholds_walk_only_large_meshes():
    # glass.tin: the 1280-triangle sphere and the 12-triangle cube are walked, the 2-triangle lamp rides in the LDS arena
    from tinsel_amd import create_gpu_renderer
    for name, expect in (("ajax_standin_96", 1), ("glass", 2), ("cornell", 0)):
        scene, cam, opt, g = _load(name)
        r = create_gpu_renderer(scene)
        assert r.walked_prims == expect, name
        r.close()


def test_fuzz_corpus_through_k_walk(walk_everything):
    import tinsel_amd
    corpus = np.load(os.path.join(oa.GOLDEN, "fuzz.golden.npz"))
    bad, walked_total = [], 0
    for k in range(int(corpus["count"])):
        scene = tinsel_amd.Scene(corpus["pack_%02d" % k].tobytes())
        out, rad, walked = _render_split(scene, scene.camera, scene.options, 2, int(corpus["first_pass_%02d" % k]))
        walked_total += walked
        if not np.array_equal(rad, corpus["radiance_%02d" % k]) or not np.array_equal(out, corpus["accum_%02d" % k]):
            bad.append(k)
    assert not bad, "scenes that differ through k_walk: %s" % bad
    assert walked_total > 0
