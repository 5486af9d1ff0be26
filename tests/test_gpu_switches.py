"""Every field of tinsel_hip_tuning (include/tinsel_hip.h; DESIGN.md Appendix A: the choices between code paths that scenes can reach) leaves
results bit-identical.  One child process per setting (tests/switch_probe.py; a process of its own keeps a crash or a leak of one setting from
the others) renders cornell, veach, glass, features, the mesh stand-in and many_spheres through the fused and the split pipeline with the
tuning handed to tinsel_hip_create_tuned / tinsel_hip_set_tuning -- the library reads nothing from the environment -- and compares radiance
and framebuffer with the golden files."""
import json
import os
import subprocess
import sys

import pytest

from tinsel_amd import abi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TAIL = lambda share, divide: {"tail_split": 1, "tail_share": share, "tail_divide": divide}     # noqa: E731
SETTINGS = [
    {},                                                     # the defaults, through the same child
    {"batch_paths": 65536},                                 # several batches per call
    {"grid_mult": 2},
    # k_bounce: its workgroup's regions as one stream or not, the shading pools (the host picks per scene and batch)
    {"bounce_share": 0}, {"bounce_share": 1},
    {"repack": 0}, {"repack": 1}, {"repack": 1, "bounce_share": 1},
    {"repack": 0, "batch_paths": 65536},
    {"shade_sorted": 1}, {"shade_sorted": 0},
    {"scene_walk": 0}, {"swalk_lds": 0},
    {"lds_scene": 0}, {"arena_lds_limit": 1024},
    {"small_mesh_bytes": 0}, {"inline_max_tris": 100000}, {"flat_scan": 0},
    {"walk": 0}, {"walk_min_tris": 1}, {"walk_min_tris": 1, "small_mesh_bytes": 0},
    {"walk_lds_stack": 0}, {"walk_lds_stack": 2}, {"walk_block": 256}, {"walk_single": 0},
    {"walk_lds_stack": 2, "walk_min_tris": 1, "small_mesh_bytes": 0},
    {"walk_refill_min": 1, "walk_leaf_min": 1}, {"walk_refill_min": 64, "walk_leaf_min": 64}, {"walk_grid_mult": 3}, {"quads_in_scan": 0},
    {"tail_split": 0}, TAIL(0.4, 8), TAIL(0.05, 2), TAIL(0.125, 4),
    # a batch's passes as two overlapped chunks on two streams (render_impl): every fixture, both pipelines; with several batches per call; off
    {"overlap": 1}, {"overlap": 1, "batch_paths": 65536}, {"overlap": 0},
    {"overlap": 1, "walk_min_tris": 1, "small_mesh_bytes": 0},
    # the accumulate kernel for filter widths up to 1: 256-thread workgroups, 512 (second half stages), staging and gathering overlapped
    {"accumulate": abi.ACCUMULATE_TILED}, {"accumulate": abi.ACCUMULATE_WIDE}, {"accumulate": abi.ACCUMULATE_PIPED},
    {"accumulate": abi.ACCUMULATE_PIPED, "batch_paths": 65536},
]


def _id(s):
    return ",".join("%s=%s" % (k, v) for k, v in s.items()) or "defaults"


@pytest.mark.parametrize("setting", SETTINGS, ids=[_id(s) for s in SETTINGS])
def test_switch_changes_no_bit(setting):
    # (a clean environment as far as the library's former switches go: nothing in it may matter)
    env = {k: v for k, v in os.environ.items() if not (k.startswith("TINSEL_HIP_") and k != "TINSEL_HIP_LIB")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "switch_probe.py"), "--tuning", json.dumps(setting)],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ": " in ln and "/" in ln.split(":")[0]]
    assert p.returncode == 0 and len(lines) == 12 and all(ln.endswith(": ok") for ln in lines), p.stdout[-3000:] + p.stderr[-2000:]


def test_the_environment_steers_nothing():
    """The former TINSEL_HIP_* switches, all set to their most disruptive values in the ENVIRONMENT: the library must not notice (the region
    plan, the pipeline, the walked primitives and the image are those of a clean environment)."""
    noisy = dict(os.environ, TINSEL_HIP_BATCH_PATHS="65536", TINSEL_HIP_GRID_MULT="2", TINSEL_HIP_NO_WALK="1", TINSEL_HIP_NO_FLAT_SCAN="1",
                 TINSEL_HIP_SMALL_MESH_BYTES="0", TINSEL_HIP_NO_LDS_SCENE="1", TINSEL_HIP_ACCUMULATE="piped", TINSEL_HIP_WALK_MIN_TRIS="0",
                 TINSEL_HIP_SHADE_SORTED="1", TINSEL_HIP_OVERLAP="1", TINSEL_HIP_REPACK="1", TINSEL_HIP_TAIL_SPLIT="0.4,8")
    clean = {k: v for k, v in os.environ.items() if not (k.startswith("TINSEL_HIP_") and k != "TINSEL_HIP_LIB")}
    outs = []
    for env in (clean, noisy):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "switch_probe.py"), "--describe", "glass,cornell"],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        outs.append([ln for ln in p.stdout.splitlines() if ln.startswith("describe ")])
    assert outs[0] and outs[0] == outs[1], outs


def test_set_tuning_between_renders():
    """tinsel_hip_set_tuning on a LIVE renderer: per-render fields change between two renders of one renderer, the create-time fields stay
    as created, the image never moves."""
    import numpy as np
    from tests.test_gpu_parity import _load
    from tinsel_amd import create_gpu_renderer
    scene, cam, opt, g = _load("glass")
    passes = int(g["passes"])
    r = create_gpu_renderer(scene, 0, abi.Tuning(walk_min_tris=1))
    r.set_pipeline(abi.PIPELINE_WAVEFRONT_SPLIT)
    for fields in ({}, {"grid_mult": 4, "walk_lds_stack": 2}, {"shade_sorted": 1, "accumulate": abi.ACCUMULATE_WIDE, "batch_paths": 65536},
                   {"overlap": 1, "walk_single": 0, "small_mesh_bytes": 0}):
        r.set_tuning(abi.Tuning(**fields))
        t = r.get_tuning()
        assert t.walk_min_tris == 1 and t.small_mesh_bytes == -1, "create-time fields must survive set_tuning"
        for k, v in fields.items():
            if k not in abi.Tuning.CREATE_FIELDS:
                assert getattr(t, k) == v
        r.init(opt.width, opt.height)
        r.set_pass_index(0)
        out = r.render(cam, opt, passes=passes)
        assert np.array_equal(out, g["accum"]), fields
    with pytest.raises(Exception):
        r.set_tuning(abi.Tuning(walk_block=100))
    r.close()
