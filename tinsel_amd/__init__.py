"""tinsel_amd -- MI355X (gfx950) streaming path tracer behind Tinsel's Renderer API.

Only what the hot path needs: `csrc/` (hand-written HIP kernels + the C-ABI of
include/tinsel_hip.h), `renderer.py` (host mirror of render.h:66-79), `abi.py`
(ctypes PODs), `build.py` (hipcc driver), `distributed.py` (pixel-tile shard + reduce).
"""
from . import abi  # noqa: F401
from .renderer import HipRenderer, HipRendererGroup, Scene, TinselHipError, create_gpu_renderer, load_library, plan_regions, resolve, selftest_arith, selftest_scan, selftest_sort, ubench, use_tuning  # noqa: F401
