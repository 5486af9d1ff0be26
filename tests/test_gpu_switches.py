"""Every A/B switch of the library leaves results bit-identical: one child process per setting (most switches
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
    {"TINSEL_HIP_REGION_LEN": "128"}, {"TINSEL_HIP_REGION_LEN": "4096"}, {"TINSEL_HIP_NO_REGION_ORDER": "1"},
    {"TINSEL_HIP_GRID_MULT": "2"}, {"TINSEL_HIP_GRID_MULT_TRACE": "1"}, {"TINSEL_HIP_GRID_MIN": "1"}, {"TINSEL_HIP_GRID_ROUND": "0"},
    {"TINSEL_HIP_BOUNCE_LAUNCHES": "per"}, {"TINSEL_HIP_BOUNCE_LAUNCHES": "per", "TINSEL_HIP_NO_REGION_ORDER": "1"},
    {"TINSEL_HIP_BOUNCE_GROUP_STEP": "0"}, {"TINSEL_HIP_BOUNCE_SHARE": "0"}, {"TINSEL_HIP_BOUNCE_SHARE": "1"}, {"TINSEL_HIP_BOUNCE_SHARE_LEN": "0"},
    {"TINSEL_HIP_REPACK": "0"}, {"TINSEL_HIP_REPACK": "1"}, {"TINSEL_HIP_REPACK": "1", "TINSEL_HIP_BOUNCE_SHARE": "1"},
    {"TINSEL_HIP_SHADE_SORTED": "1"}, {"TINSEL_HIP_SHADE_SORTED": "0"}, {"TINSEL_HIP_LIGHTS_IN_EXTEND": "0"},
    {"TINSEL_HIP_NO_SCENE_WALK": "1"}, {"TINSEL_HIP_SWALK_NO_LDS": "1"}, {"TINSEL_HIP_SWALK_REFILL": "8", "TINSEL_HIP_SWALK_LEAFMIN": "1"},
    {"TINSEL_HIP_SWALK_GRID_MULT": "4", "TINSEL_HIP_SWALK_LIST_STEP": "1"},
    {"TINSEL_HIP_NO_LDS_SCENE": "1"}, {"TINSEL_HIP_NO_LDS_TEMPLATE": "1"}, {"TINSEL_HIP_ARENA_LDS_LIMIT": "1024"},
    {"TINSEL_HIP_SMALL_MESH_BYTES": "0"}, {"TINSEL_HIP_INLINE_MAX_TRIS": "100000"}, {"TINSEL_HIP_NO_FLAT_SCAN": "1"},
    {"TINSEL_HIP_NO_PLANE_TABLE": "1"}, {"TINSEL_HIP_NO_BIN": "1"}, {"TINSEL_HIP_NO_SORT_QUEUES": "1"}, {"TINSEL_HIP_NO_DEFER_MESHES": "1"}, {"TINSEL_HIP_NO_TWO_LEAVES": "1"},
    {"TINSEL_HIP_NO_SHADE_ARENA": "1"}, {"TINSEL_HIP_NO_LEAN_SCAN": "1"},
    {"TINSEL_HIP_NO_WALK": "1"}, {"TINSEL_HIP_WALK_MIN_TRIS": "1"}, {"TINSEL_HIP_WALK_MIN_TRIS": "1", "TINSEL_HIP_SMALL_MESH_BYTES": "0"},
    {"TINSEL_HIP_WALK_LDS_STACK": "0"}, {"TINSEL_HIP_WALK_LDS_STACK": "2"}, {"TINSEL_HIP_WALK_GRID_MULT": "3"},
    {"TINSEL_HIP_WALK_REFILL": "1", "TINSEL_HIP_WALK_LEAFMIN": "1"}, {"TINSEL_HIP_WALK_REFILL": "64", "TINSEL_HIP_WALK_LEAFMIN": "64"},
    {"TINSEL_HIP_WALK_TOP": "0"}, {"TINSEL_HIP_WALK_BLOCK": "256"}, {"TINSEL_HIP_WALK_LIST_STEP": "1"},
    {"TINSEL_HIP_WALK_PAIRS": "1"}, {"TINSEL_HIP_WALK_PAIRS": "1", "TINSEL_HIP_WALK_SINGLE": "0"}, {"TINSEL_HIP_WALK_SINGLE": "0"},
    {"TINSEL_HIP_WALK_PAIRS": "1", "TINSEL_HIP_WALK_MIN_TRIS": "1", "TINSEL_HIP_SMALL_MESH_BYTES": "0"},
    {"TINSEL_HIP_WALK_PAIRS": "1", "TINSEL_HIP_WALK_LDS_STACK": "2", "TINSEL_HIP_WALK_MIN_TRIS": "1", "TINSEL_HIP_WALK_LEAFMIN": "1"}, {"TINSEL_HIP_WALK_BLOCK": "256", "TINSEL_HIP_WALK_PAIRS": "1"},
    {"TINSEL_HIP_TAIL_SPLIT": "0"}, {"TINSEL_HIP_TAIL_SPLIT_SPLIT": "1"}, {"TINSEL_HIP_TAIL_SPLIT_SPLIT": "1", "TINSEL_HIP_TAIL_SPLIT": "0.3,2"}, {"TINSEL_HIP_TAIL_SPLIT": "0.4,8"}, {"TINSEL_HIP_TAIL_SPLIT": "0.05,2"}, {"TINSEL_HIP_TAIL_SPLIT": "0.125,4"},
    {"TINSEL_HIP_ACC_NO_SPAN": "1"}, {"TINSEL_HIP_ACC_WIDE": "0"}, {"TINSEL_HIP_ACC_WIDE": "1"},
    # a batch's passes as two overlapped chunks on two streams (render_impl): every fixture, both pipelines; with several batches per call; off
    {"TINSEL_HIP_OVERLAP": "1"}, {"TINSEL_HIP_OVERLAP": "1", "TINSEL_HIP_BATCH_PATHS": "65536"}, {"TINSEL_HIP_OVERLAP": "0"},
    {"TINSEL_HIP_OVERLAP": "1", "TINSEL_HIP_WALK_MIN_TRIS": "1", "TINSEL_HIP_SMALL_MESH_BYTES": "0"}, {"TINSEL_HIP_OVERLAP": "1", "TINSEL_HIP_BOUNCE_LAUNCHES": "per"},
    # k_shade traces the shadow rays itself (no k_shadow launch): scenes as they are, and with every mesh walked by k_walk (the lean variant)
    {"TINSEL_HIP_SHADOW_IN_SHADE": "1"}, {"TINSEL_HIP_SHADOW_IN_SHADE": "0"},
    {"TINSEL_HIP_SHADOW_IN_SHADE": "1", "TINSEL_HIP_WALK_MIN_TRIS": "1", "TINSEL_HIP_SMALL_MESH_BYTES": "0"},
    {"TINSEL_HIP_SHADOW_IN_SHADE": "1", "TINSEL_HIP_WALK_LDS_STACK": "2", "TINSEL_HIP_WALK_MIN_TRIS": "1"}, {"TINSEL_HIP_SHADOW_IN_SHADE": "1", "TINSEL_HIP_OVERLAP": "1"},
]


@pytest.mark.parametrize("setting", SETTINGS, ids=[",".join("%s=%s" % (k.replace("TINSEL_HIP_", ""), v) for k, v in s.items()) or "defaults" for s in SETTINGS])
def test_switch_changes_no_bit(setting):
    env = dict(os.environ, **setting)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "switch_probe.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ": " in ln and "/" in ln.split(":")[0]]
    assert p.returncode == 0 and len(lines) == 12 and all(ln.endswith(": ok") for ln in lines), p.stdout[-3000:] + p.stderr[-2000:]
