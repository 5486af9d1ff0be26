"""The reference's other shipped scene files through the C-ABI: data/table.tin (15 primitives, seven meshes of up to 30,240 triangles,
maxDepth 8), transmission.tin (seven walked meshes behind glass, maxDepth 16), meshlight.tin (a 36,752-triangle mesh AS the light:
LightSample's triangle CDF search, render.cpp:107-144 / mesh.h), example.tin and env.tin (probe-lit meshes).

tests/golden/make_reference_scenes.py loads each .tin with the reference's own loader + Scene::Build and renders the golden with the
reference's PathTrace / AddSample; the packs (1-7 MB, git-ignored under tests/golden/large/) travel with the tree, the goldens are
committed.  Bar: per-path radiance and framebuffer BIT-IDENTICAL, every pipeline."""
import os

import numpy as np
import pytest

from tinsel_amd import abi
from tests import oracle_api as oa

pytestmark = pytest.mark.gpu

REF_SCENES = ["table", "transmission", "meshlight", "example", "env"]


def _pack(name):
    return os.path.join(oa.GOLDEN, "large", name + ".pack")


def _load(name):
    import tinsel_amd
    if not os.path.exists(_pack(name)):
        pytest.skip("tests/golden/large/%s.pack not generated (tests/golden/make_reference_scenes.py)" % name)
    g = np.load(os.path.join(oa.GOLDEN, name + ".golden.npz"))
    scene = tinsel_amd.Scene.load_pack(_pack(name))
    cam = abi.Camera.from_buffer_copy(g["camera"].tobytes())
    opt = abi.Options.from_buffer_copy(g["options"].tobytes())
    return scene, cam, opt, g


@pytest.mark.parametrize("pipeline", [abi.PIPELINE_AUTO, abi.PIPELINE_WAVEFRONT, abi.PIPELINE_MEGAKERNEL, abi.PIPELINE_WAVEFRONT_SPLIT],
                         ids=["auto", "wavefront", "mega", "split"])
@pytest.mark.parametrize("name", REF_SCENES)
def test_shipped_scene_matches_golden(name, pipeline):
    import tinsel_amd
    scene, cam, opt, g = _load(name)
    passes = int(g["passes"])
    r = tinsel_amd.create_gpu_renderer(scene)
    r.set_pipeline(pipeline)
    r.init(opt.width, opt.height)
    out = r.render(cam, opt, passes=passes)
    rad = r.batch_radiance(passes, opt.height, opt.width)
    st = r.stats()
    r.close()
    assert st["samples"] == passes*opt.width*opt.height
    assert np.abs(g["radiance"]).max() > 0, "a black golden proves nothing"
    exact = np.all(rad == g["radiance"], axis=-1)
    assert exact.all(), "%d of %d paths are not bit-identical to the reference" % ((~exact).sum(), exact.size)
    assert np.array_equal(out, g["accum"]), "framebuffer differs (L2 %.3e)" % oa.image_l2(out, g["accum"])


@pytest.mark.parametrize("name", REF_SCENES)
def test_shipped_scene_normals(name):
    import tinsel_amd
    scene, cam, opt, g = _load(name)
    nopt = opt.copy()
    nopt.mode = abi.MODE_NORMALS
    r = tinsel_amd.create_gpu_renderer(scene)
    r.init(opt.width, opt.height)
    out = r.render(cam, nopt, passes=1)
    r.close()
    assert np.array_equal(out, g["normals"])


@pytest.mark.parametrize("mode", [abi.BVH_LBVH, abi.BVH_PLOC], ids=["lbvh", "ploc"])
@pytest.mark.parametrize("name", ["table", "meshlight", "transmission"])
def test_shipped_scene_under_device_built_trees(name, mode):
    """the device-side mesh builders on real many-mesh scenes: a different tree, the same closest hits except on exact ties between
    triangles -- so almost every path stays bit-identical and the image is inside the north_star bar"""
    import tinsel_amd
    scene, cam, opt, g = _load(name)
    passes = int(g["passes"])
    r = tinsel_amd.create_gpu_renderer(scene)
    ms = r.set_mesh_bvh(mode)
    r.init(opt.width, opt.height)
    out = r.render(cam, opt, passes=passes)
    rad = r.batch_radiance(passes, opt.height, opt.width)
    r.close()
    same = np.all(rad == g["radiance"], axis=-1).mean()
    l2 = oa.image_l2(out, g["accum"])
    print("%s: device build %.3f ms; paths bit-identical to the reference trees: %.4f %%, per-pixel L2 %.3e" % (name, ms, 100*same, l2))
    assert same >= 0.999 and l2 <= 1e-3
