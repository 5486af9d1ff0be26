"""Opt-in Russian roulette (tinsel_hip_set_russian_roulette; not in the reference, whose loop runs every path to
maxDepth): every pipeline is bit-identical to the C oracle with the same rule restated (port_set_russian_roulette), the
estimate stays unbiased, deep configs get cheaper, and with the switch off nothing changes."""
import os
import subprocess

import numpy as np
import pytest

from tinsel_amd import abi
from tests import oracle_api as oa

pytestmark = pytest.mark.gpu


def _load(name):
    import tinsel_amd
    g = np.load(os.path.join(oa.GOLDEN, name + ".golden.npz"))
    scene = tinsel_amd.Scene.load_pack(os.path.join(oa.GOLDEN, name + ".pack"))
    cam = abi.Camera.from_buffer_copy(g["camera"].tobytes())
    opt = abi.Options.from_buffer_copy(g["options"].tobytes())
    return scene, cam, opt, g


def _port():
    if not oa.have_port():
        subprocess.run(["make", "-C", os.path.join(oa.ROOT, "oracle"), "port"], check=True)
    P = oa.PortOracle()
    P.lib.port_set_russian_roulette.restype = None
    return P


@pytest.mark.parametrize("name,depth,start", [("glass", 12, 3), ("features", 6, 2), ("cornell", 4, 1), ("features_probe", 8, 4)])
def test_roulette_matches_the_oracle_rule(name, depth, start):
    import tinsel_amd
    P = _port()
    scene, cam, opt, g = _load(name)
    opt.max_depth = depth
    h = P.load_pack(os.path.join(oa.GOLDEN, name + ".pack"))
    try:
        P.lib.port_set_russian_roulette(start)
        ref, rad, _ = P.render_seeded(h, cam, opt, 0, 3, want_accum=True, want_radiance=True)
    finally:
        P.lib.port_set_russian_roulette(0)
    plain, _, _ = P.render_seeded(h, cam, opt, 0, 3)
    P.free(h)
    assert not np.array_equal(ref, plain)               # the rule does something
    for pipeline in (abi.PIPELINE_WAVEFRONT, abi.PIPELINE_WAVEFRONT_SPLIT, abi.PIPELINE_MEGAKERNEL):
        r = tinsel_amd.create_gpu_renderer(scene)
        r.set_pipeline(pipeline)
        r.set_russian_roulette(start)
        r.init(opt.width, opt.height)
        out = r.render(cam, opt, passes=3)
        got = r.batch_radiance(3, opt.height, opt.width)
        st = r.stats()
        # switched off again: the reference's image, bit for bit
        r.set_russian_roulette(0)
        r.init(opt.width, opt.height)
        r.set_pass_index(0)
        off = r.render(cam, opt, passes=3)
        r.close()
        assert np.array_equal(got, rad), "pipeline %d" % pipeline
        assert np.array_equal(out, ref)
        assert np.array_equal(off, plain)


def test_roulette_is_unbiased_and_cheaper():
    import tinsel_amd
    scene, cam, opt, g = _load("glass")
    opt.width, opt.height, opt.max_depth = 96, 96, 12
    res = {}
    for start in (0, 3):
        r = tinsel_amd.create_gpu_renderer(scene)
        r.set_russian_roulette(start)
        r.init(opt.width, opt.height)
        out = r.render(cam, opt, passes=1024)
        res[start] = (out[..., :3].sum(axis=(0, 1))/out[..., 3].sum(), r.stats()["rays"])
        r.close()
    (m0, rays0), (m1, rays1) = res[0], res[3]
    print("mean radiance without / with roulette: %s / %s; rays %d / %d (%.2fx fewer)" % (m0, m1, rays0, rays1, rays0/rays1))
    assert np.all(np.abs(m1/m0 - 1.0) < 0.01)
    assert rays1 < 0.8*rays0
