"""Every switch of the library (the ones that choose between code paths scenes can reach; round 5 removed the tuning
knobs and the arms that lost their A/B) leaves results bit-identical: one child process per setting (most switches
are read once per process) renders cornell, veach, glass, features, the mesh stand-in and many_spheres through the fused and the
split pipeline and compares radiance and framebuffer with the golden files (tests/switch_probe.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SETTINGS = [
    {},                                                     # the defaults, through the same child
    {"TINSEL_HIP_BATCH_PATHS": "65536"},                    # several batches per call
    {"TINSEL_HIP_GRID_MULT": "2"},
    # k_bounce: its workgroup's regions as one stream or not, the shading pools (the host picks per scene and batch)
    {"TINSEL_HIP_BOUNCE_SHARE": "0"}, {"TINSEL_HIP_BOUNCE_SHARE": "1"},
    {"TINSEL_HIP_REPACK": "0"}, {"TINSEL_HIP_REPACK": "1"}, {"TINSEL_HIP_REPACK": "1", "TINSEL_HIP_BOUNCE_SHARE": "1"},
    {"TINSEL_HIP_REPACK": "0", "TINSEL_HIP_BATCH_PATHS": "65536"},
    {"TINSEL_HIP_SHADE_SORTED": "1"}, {"TINSEL_HIP_SHADE_SORTED": "0"},
    {"TINSEL_HIP_NO_SCENE_WALK": "1"}, {"TINSEL_HIP_SWALK_NO_LDS": "1"},
    {"TINSEL_HIP_NO_LDS_SCENE": "1"}, {"TINSEL_HIP_ARENA_LDS_LIMIT": "1024"},
    {"TINSEL_HIP_SMALL_MESH_BYTES": "0"}, {"TINSEL_HIP_INLINE_MAX_TRIS": "100000"}, {"TINSEL_HIP_NO_FLAT_SCAN": "1"},
    {"TINSEL_HIP_NO_WALK": "1"}, {"TINSEL_HIP_WALK_MIN_TRIS": "1"}, {"TINSEL_HIP_WALK_MIN_TRIS": "1", "TINSEL_HIP_SMALL_MESH_BYTES": "0"},
    {"TINSEL_HIP_WALK_LDS_STACK": "0"}, {"TINSEL_HIP_WALK_LDS_STACK": "2"}, {"TINSEL_HIP_WALK_BLOCK": "256"}, {"TINSEL_HIP_WALK_SINGLE": "0"},
    {"TINSEL_HIP_WALK_LDS_STACK": "2", "TINSEL_HIP_WALK_MIN_TRIS": "1", "TINSEL_HIP_SMALL_MESH_BYTES": "0"},
    {"TINSEL_HIP_TAIL_SPLIT": "0"}, {"TINSEL_HIP_TAIL_SPLIT": "0.4,8"}, {"TINSEL_HIP_TAIL_SPLIT": "0.05,2"}, {"TINSEL_HIP_TAIL_SPLIT": "0.125,4"},
    # a batch's passes as two overlapped chunks on two streams (render_impl): every fixture, both pipelines; with several batches per call; off
    {"TINSEL_HIP_OVERLAP": "1"}, {"TINSEL_HIP_OVERLAP": "1", "TINSEL_HIP_BATCH_PATHS": "65536"}, {"TINSEL_HIP_OVERLAP": "0"},
    {"TINSEL_HIP_OVERLAP": "1", "TINSEL_HIP_WALK_MIN_TRIS": "1", "TINSEL_HIP_SMALL_MESH_BYTES": "0"},
    # the accumulate kernel for filter widths up to 1: 256-thread workgroups, 512 (second half stages), staging and gathering overlapped
    {"TINSEL_HIP_ACCUMULATE": "tiled"}, {"TINSEL_HIP_ACCUMULATE": "wide"}, {"TINSEL_HIP_ACCUMULATE": "piped"},
    {"TINSEL_HIP_ACCUMULATE": "piped", "TINSEL_HIP_BATCH_PATHS": "65536"},
]


@pytest.mark.parametrize("setting", SETTINGS, ids=[",".join("%s=%s" % (k.replace("TINSEL_HIP_", ""), v) for k, v in s.items()) or "defaults" for s in SETTINGS])
def test_switch_changes_no_bit(setting):
    env = dict(os.environ, **setting)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "switch_probe.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ": " in ln and "/" in ln.split(":")[0]]
    assert p.returncode == 0 and len(lines) == 12 and all(ln.endswith(": ok") for ln in lines), p.stdout[-3000:] + p.stderr[-2000:]
