"""The PAIRED wavefront pipeline (tinsel_amd/csrc/tn_paired.h, TINSEL_PIPELINE_WAVEFRONT_PAIRED): one k_walk launch per bounce for the shadow
rays of the last bounce AND the extension rays of this one, one streaming kernel (k_step) for everything else -- the light samples' BSDF terms
evaluated where the samples are drawn and carried across the trace, a path's last samples resolved one step after its loop ended.  It must give,
bit for bit, what every other pipeline gives: the reference's PathTrace (render.cpp:230-388) and AddSample on the same seeds."""
import os

import numpy as np
import pytest

from tinsel_amd import abi
from tests import oracle_api as oa
from tests.test_gpu_parity import SCENES, _load

pytestmark = pytest.mark.gpu


def _render_paired(scene, cam, opt, passes, first_pass=0, tuning=None, batch=None, roulette=0):
    from tinsel_amd import create_gpu_renderer
    r = create_gpu_renderer(scene, 0, tuning)
    walked = r.walked_prims
    r.set_pipeline(abi.PIPELINE_WAVEFRONT_PAIRED)
    if batch:
        r.set_batch_paths(batch)
    if roulette:
        r.set_russian_roulette(roulette)
    r.init(opt.width, opt.height)
    r.set_pass_index(first_pass)
    r.enable_kernel_timing(True)
    out = r.render(cam, opt, passes=passes)
    rad = r.batch_radiance(passes, opt.height, opt.width)
    kernels = r.kernel_times()
    st = r.stats()
    r.close()
    return out, rad, walked, kernels, st


WALK_ALL = dict(walk_min_tris=0, small_mesh_bytes=0)        # every mesh of every fixture through k_walk's mixed mode
# quads (two-triangle meshes: 216 B) stay in the arena and are tested in the scan, every other mesh is walked: the lean kernels' second
# level (SceneT<.., 2, ..>: k_step<1,2,1>, k_extend<0,1,2,1>, k_shadow<0,1,2,1>) -- what glass.tin gets by default
QUADS_IN_SCAN = dict(walk_min_tris=0, small_mesh_bytes=256)


@pytest.mark.parametrize("tune", [None, WALK_ALL, dict(walk_min_tris=0, small_mesh_bytes=0, walk_single=0), dict(lds_scene=0), QUADS_IN_SCAN, dict(quads_in_scan=0)],
                         ids=["default", "walk-all", "walk-all-rays-kernel", "arena-in-hbm", "quads-in-scan", "quads-general-kernels"])
@pytest.mark.parametrize("name", SCENES)
def test_paired_pipeline_matches_the_reference(name, tune):
    scene, cam, opt, g = _load(name)
    passes = int(g["passes"])
    out, rad, walked, kernels, st = _render_paired(scene, cam, opt, passes, tuning=abi.Tuning(**tune) if tune else None)
    assert np.array_equal(rad, g["radiance"]), "%s: %d paths differ (%d walked primitives, kernels %s)" % (
        name, int((rad != g["radiance"]).any(axis=-1).sum()), walked, sorted(kernels))
    assert np.array_equal(out, g["accum"])
    assert st["samples"] == passes*opt.width*opt.height
    if name != "many_spheres":          # (203 primitives: beyond the flat scan, the paired pipeline hands over to the split one)
        assert "k_step" in kernels and "k_shade" not in kernels and "k_extend" not in kernels, sorted(kernels)


def test_quads_in_the_scan_beside_walked_meshes_split_pipeline():
    """the same second level of the lean kernels in the SPLIT pipeline (k_extend / k_shadow), every fixture; AUTO keeps glass.tin's shape
    (sphere and cube walked, the lamp a quad in the arena) on the split pipeline, with or without quads_in_scan"""
    from tinsel_amd import create_gpu_renderer
    for name in SCENES:
        scene, cam, opt, g = _load(name)
        r = create_gpu_renderer(scene, 0, abi.Tuning(**QUADS_IN_SCAN))
        r.set_pipeline(abi.PIPELINE_WAVEFRONT_SPLIT)
        r.init(opt.width, opt.height)
        out = r.render(cam, opt, passes=int(g["passes"]))
        r.close()
        assert np.array_equal(out, g["accum"]), name
    scene, cam, opt, g = _load("glass")
    for tune, want in ((None, "k_shade"), (abi.Tuning(quads_in_scan=0), "k_shade")):
        r = create_gpu_renderer(scene, 0, tune)
        r.init(opt.width, opt.height)
        r.enable_kernel_timing(True)
        out = r.render(cam, opt, passes=int(g["passes"]))
        kernels = r.kernel_times()
        r.close()
        assert np.array_equal(out, g["accum"]) and want in kernels, sorted(kernels)


def test_paired_runs_one_walk_per_bounce():
    """What the re-cut is for: glass at depth 12 -- 13 steps (12 bounces + the step that resolves the last bounce's light samples) and 13
    walks, where the split pipeline launches 24 walks and 36 streaming kernels."""
    scene, cam, opt, g = _load("glass")
    opt.max_depth = 12
    out, rad, walked, kernels, st = _render_paired(scene, cam, opt, 2)
    assert walked == 2 and kernels["k_step"][0] == 13 and kernels["k_walk"][0] == 13, kernels
    from tests.test_gpu_parity import _render
    want, st2 = _render(scene, cam, opt, 2, abi.PIPELINE_WAVEFRONT_SPLIT, want_radiance=True)
    assert np.array_equal(out, want) and np.array_equal(rad, st2["radiance"])
    assert st["rays"] == st2["rays"] and st["shadow_rays"] == st2["shadow_rays"]


@pytest.mark.parametrize("name", ["features", "glass", "veach", "ajax_standin_96"])
def test_paired_small_batches_and_roulette(name):
    """several batches per call; opt-in Russian roulette (one more draw from the path's stream behind the BSDF sample) against the split pipeline"""
    scene, cam, opt, g = _load(name)
    passes = int(g["passes"])
    out, rad, _, _, _ = _render_paired(scene, cam, opt, passes, batch=65536)
    assert np.array_equal(out, g["accum"])
    from tinsel_amd import create_gpu_renderer
    r = create_gpu_renderer(scene)
    r.set_pipeline(abi.PIPELINE_WAVEFRONT_SPLIT)
    r.set_russian_roulette(2)
    r.init(opt.width, opt.height)
    want = r.render(cam, opt, passes=passes)
    r.close()
    got, _, _, _, _ = _render_paired(scene, cam, opt, passes, roulette=2)
    assert np.array_equal(got, want)


def test_fuzz_corpus_through_the_paired_pipeline():
    import tinsel_amd
    corpus = np.load(os.path.join(oa.GOLDEN, "fuzz.golden.npz"))
    bad = []
    for k in range(int(corpus["count"])):
        scene = tinsel_amd.Scene(corpus["pack_%02d" % k].tobytes())
        for tune in (None, abi.Tuning(**WALK_ALL)):
            out, rad, _, _, _ = _render_paired(scene, scene.camera, scene.options, 2, int(corpus["first_pass_%02d" % k]), tuning=tune)
            if not np.array_equal(rad, corpus["radiance_%02d" % k]) or not np.array_equal(out, corpus["accum_%02d" % k]):
                bad.append((k, tune is not None))
    assert not bad, "scenes that differ through the paired pipeline: %s" % bad


@pytest.mark.parametrize("name", ["glass", "features", "motionblur"])
def test_every_rank_of_a_shard_through_the_paired_pipeline(name):
    """pixel-tile shards (rank-local slots): the sum over ranks of the accumulators is the unsharded image's paths, rank by rank bit-identical to
    the split pipeline's shard"""
    from tinsel_amd import create_gpu_renderer
    scene, cam, opt, g = _load(name)
    passes = int(g["passes"])
    for rank in range(3):
        imgs = []
        for pipe in (abi.PIPELINE_WAVEFRONT_PAIRED, abi.PIPELINE_WAVEFRONT_SPLIT):
            r = create_gpu_renderer(scene)
            r.set_pipeline(pipe)
            r.set_shard(rank, 3, 16)
            r.init(opt.width, opt.height)
            imgs.append(r.render(cam, opt, passes=passes))
            r.close()
        assert np.array_equal(imgs[0], imgs[1]), "rank %d" % rank
