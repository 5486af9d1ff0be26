"""The split pipeline's dense path state (regions packed at both ends, ballot appends, no queues):
whatever the frame, the batch and the number of regions, it must give the bits of the megakernel arm, which shares none of
that machinery (one lane walks one whole path) and is itself pinned to the reference in test_gpu_parity.py."""
import numpy as np
import pytest

from tinsel_amd import abi
from tests.test_gpu_parity import _load

pytestmark = pytest.mark.gpu


def _render(scene, cam, opt, passes, pipeline, batch=None):
    from tinsel_amd import create_gpu_renderer
    r = create_gpu_renderer(scene)
    r.set_pipeline(pipeline)
    if batch:
        r.set_batch_paths(batch)
    r.init(opt.width, opt.height)
    out = r.render(cam, opt, passes=passes)
    counts = r.queue_counts() if pipeline != abi.PIPELINE_MEGAKERNEL else None      # the megakernel arm has no bounces to count
    st = r.stats()
    r.close()
    return out, counts, st


# frames of one pixel, of less than a wave, of a few waves with a ragged tail, and one that fills several regions per wave
# of the grid; batches of one pass and of all passes (regions of 64 positions up to thousands)
@pytest.mark.parametrize("size", [(1, 1), (7, 5), (65, 3), (257, 129)], ids=lambda s: "%dx%d" % s)
@pytest.mark.parametrize("name", ["glass", "features", "many_spheres", "ajax_standin_96"])
def test_odd_frames_and_batches(name, size):
    scene, cam, opt, g = _load(name)
    o = opt.copy()
    o.width, o.height = size
    passes = 5
    ref, _, _ = _render(scene, cam, o, passes, abi.PIPELINE_MEGAKERNEL)
    for batch in (None, 1024):
        out, (live, shadow), st = _render(scene, cam, o, passes, abi.PIPELINE_WAVEFRONT_SPLIT, batch=batch)
        assert st["samples"] == passes*size[0]*size[1]
        assert np.array_equal(out, ref), "split pipeline differs from the megakernel arm (batch %s)" % batch
        # the last batch's bookkeeping: every generated path is alive at bounce 0, nothing comes back to life, and only
        # paths that hit something have shadow rays
        assert live[0] > 0 and live[0] <= passes*size[0]*size[1]
        assert all(a >= b for a, b in zip(live, live[1:]))
        assert all(s <= l for s, l in zip(shadow, live))


def test_queue_counts_of_both_pipelines():
    scene, cam, opt, g = _load("cornell")
    passes = 3
    for pipeline in (abi.PIPELINE_WAVEFRONT, abi.PIPELINE_WAVEFRONT_SPLIT):
        _, (live, shadow), st = _render(scene, cam, opt, passes, pipeline)
        assert len(live) == opt.max_depth
        assert live[0] == passes*opt.width*opt.height
        assert all(a >= b for a, b in zip(live, live[1:])) and live[-1] > 0
    # the split pipeline also counts the paths with shadow rays: the paths that hit something
    assert 0 < shadow[0] <= live[0]
