"""Host-side mirror of the reference's renderer interface over the C-ABI.

Reference interface (src/render.h:66-79):

    struct Renderer { virtual void Init(int w, int h); virtual void Render(const Camera&, const Options&, Color* out); };
    Renderer* CreateGpuRenderer(const Scene* s);

Here: `create_gpu_renderer(scene) -> HipRenderer` with `.init(w, h)` and
`.render(camera, options, output=None, passes=1)`; same argument meaning, same
framebuffer semantics (output == running sum of (rgb*w, w) since init), same
"Init before Render, options.width/height must match Init" contract.  Errors
raise `TinselHipError` carrying tinsel_hip_last_error().

The compute lives entirely in libtinsel_hip.so (hand-written HIP for gfx950).
There is no CPU fallback: without the library or without a GPU these calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TINSEL_HIP_LIB", os.path.join(_HERE, "libtinsel_hip.so"))     # env: A/B builds only

_lib = None
# The tuning renderers are created with when the caller passes none: None = the library's defaults (tinsel_hip_create).  A plain module
# attribute of THIS binding (tests set it through `use_tuning` to push whole suites through another code path) -- the library itself reads
# no environment variable and has no process-wide state.
DEFAULT_TUNING = None


class TinselHipError(RuntimeError):
    pass


def load_library():
    """dlopen libtinsel_hip.so (once).  torch, when importable, is imported first so that both
    share one HIP runtime (same SONAME libamdhip64.so.7) and torch device pointers are valid here."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TinselHipError(
            "%s is missing: build it with `python -m tinsel_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is plumbing, not a requirement of the ABI
        pass
    L = C.CDLL(LIB_PATH)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    L.tinsel_hip_create.restype = vp
    L.tinsel_hip_create.argtypes = [C.POINTER(abi.SceneDesc), ci]
    if hasattr(L, "tinsel_hip_create_tuned"):         # absent from libraries built before round 6 (TINSEL_HIP_LIB: an A/B against an older build)
        L.tinsel_hip_tuning_init.restype = None
        L.tinsel_hip_tuning_init.argtypes = [C.POINTER(abi.Tuning)]
        L.tinsel_hip_create_tuned.restype = vp
        L.tinsel_hip_create_tuned.argtypes = [C.POINTER(abi.SceneDesc), ci, C.POINTER(abi.Tuning)]
        L.tinsel_hip_set_tuning.argtypes = [vp, C.POINTER(abi.Tuning)]
        L.tinsel_hip_get_tuning.argtypes = [vp, C.POINTER(abi.Tuning)]
        L.tinsel_hip_group_create_tuned.restype = vp
        L.tinsel_hip_group_create_tuned.argtypes = [C.POINTER(abi.SceneDesc), ci, ci, C.POINTER(abi.Tuning)]
        L.tinsel_hip_comm_unique_id.argtypes = [vp, ci]
        L.tinsel_hip_comm_init.argtypes = [vp, vp, ci, ci]
        L.tinsel_hip_comm_size.argtypes = [vp]
        L.tinsel_hip_comm_reduce_accum.argtypes = [vp, vp, ci, vp]
    L.tinsel_hip_destroy.restype = None
    L.tinsel_hip_destroy.argtypes = [vp]
    L.tinsel_hip_init.argtypes = [vp, ci, ci]
    L.tinsel_hip_init_external.argtypes = [vp, ci, ci, vp]
    L.tinsel_hip_render.argtypes = [vp, C.POINTER(abi.Camera), C.POINTER(abi.Options), vp, ci]
    L.tinsel_hip_render_async.argtypes = [vp, C.POINTER(abi.Camera), C.POINTER(abi.Options), ci, vp]
    L.tinsel_hip_accum_device_ptr.restype = vp
    L.tinsel_hip_accum_device_ptr.argtypes = [vp]
    L.tinsel_hip_read_accum.argtypes = [vp, vp]
    L.tinsel_hip_set_shard.argtypes = [vp, ci, ci, ci]
    L.tinsel_hip_set_pipeline.argtypes = [vp, ci]
    L.tinsel_hip_set_pass_index.argtypes = [vp, C.c_uint32]
    L.tinsel_hip_get_pass_index.restype = C.c_uint32
    L.tinsel_hip_get_pass_index.argtypes = [vp]
    L.tinsel_hip_stats.restype = None
    L.tinsel_hip_stats.argtypes = [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.POINTER(C.c_double)]
    L.tinsel_hip_reset_stats.restype = None
    L.tinsel_hip_reset_stats.argtypes = [vp]
    L.tinsel_hip_stats_detail.argtypes = [vp, vp]
    L.tinsel_hip_set_detail_counters.argtypes = [vp, ci]
    L.tinsel_hip_kernel_times.argtypes = [vp, C.POINTER(abi.KernelTime), ci]
    L.tinsel_hip_enable_kernel_timing.argtypes = [vp, ci]
    L.tinsel_hip_set_batch_paths.argtypes = [vp, C.c_ulonglong]
    L.tinsel_hip_read_batch_radiance.restype = C.c_longlong
    L.tinsel_hip_read_batch_radiance.argtypes = [vp, vp, C.c_ulonglong]
    L.tinsel_hip_leaf.argtypes = [vp, ci, ci, ci, vp, ci, vp, vp, ci, C.POINTER(abi.Camera), ci, ci]
    L.tinsel_hip_write_accum.argtypes = [vp, vp, C.c_uint32]
    L.tinsel_hip_reserve.argtypes = [vp, ci, ci]
    L.tinsel_hip_set_russian_roulette.argtypes = [vp, ci]
    L.tinsel_hip_set_mesh_bvh.argtypes = [vp, ci, C.POINTER(C.c_double)]
    L.tinsel_hip_present.argtypes = [vp, C.POINTER(abi.Options), ci, C.c_float, vp]
    L.tinsel_hip_present_async.argtypes = [vp, C.POINTER(abi.Options), ci, C.c_float, vp]
    L.tinsel_hip_present_device_ptr.restype = vp
    L.tinsel_hip_present_device_ptr.argtypes = [vp]
    L.tinsel_image_quantize_rgb8.argtypes = [vp, ci, ci, vp]
    L.tinsel_hip_stack_entries.argtypes = [vp]
    L.tinsel_hip_nee_per_path.argtypes = [vp]
    L.tinsel_hip_walked_prims.argtypes = [vp]
    if hasattr(L, "tinsel_hip_queue_counts"):        # absent from older builds loaded through TINSEL_HIP_LIB for an A/B
        L.tinsel_hip_queue_counts.argtypes = [vp, C.POINTER(C.c_uint32), C.c_int]
    L.tinsel_hip_set_lookahead.argtypes = [vp, ci]
    L.tinsel_hip_refit_mesh.argtypes = [vp, ci, vp, ci, vp]
    if hasattr(L, "tinsel_hip_rebuild_scene"):         # absent from older builds loaded through TINSEL_HIP_LIB for an A/B
        L.tinsel_hip_set_primitive_transform.argtypes = [vp, ci, C.POINTER(abi.Transform), C.POINTER(abi.Transform)]
        L.tinsel_hip_rebuild_scene.argtypes = [vp, ci, vp, ci, C.POINTER(C.c_double)]
    L.tinsel_hip_set_probe_sampling.argtypes = [vp, ci]
    L.tinsel_hip_set_arithmetic.argtypes = [vp, ci]
    L.tinsel_hip_get_arithmetic.argtypes = [vp]
    L.tinsel_hip_group_create.restype = vp
    L.tinsel_hip_group_create.argtypes = [C.POINTER(abi.SceneDesc), ci, ci]
    L.tinsel_hip_group_destroy.restype = None
    L.tinsel_hip_group_destroy.argtypes = [vp]
    L.tinsel_hip_group_init.argtypes = [vp, ci, ci]
    L.tinsel_hip_group_render.argtypes = [vp, C.POINTER(abi.Camera), C.POINTER(abi.Options), vp, ci]
    L.tinsel_hip_group_present.argtypes = [vp, C.POINTER(abi.Options), ci, C.c_float, vp]
    L.tinsel_hip_group_size.argtypes = [vp]
    L.tinsel_hip_group_member.restype = vp
    L.tinsel_hip_group_member.argtypes = [vp, ci]
    L.tinsel_hip_group_set_lookahead.argtypes = [vp, ci]
    L.tinsel_hip_ubench.argtypes = [ci, ci, C.c_ulonglong, ci, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.tinsel_hip_selftest_arith.argtypes = [ci, ci, ci, C.POINTER(C.c_ulonglong), C.POINTER(C.c_uint)]
    if hasattr(L, "tinsel_hip_selftest_sort"):          # (absent from libraries built before round 5: TINSEL_HIP_LIB)
        L.tinsel_hip_selftest_sort.argtypes = [ci, vp, C.c_ulonglong, ci, ci]
        L.tinsel_hip_selftest_scan.argtypes = [ci, vp, vp, C.c_ulonglong]
    L.tinsel_hip_plan_regions.argtypes = [C.c_ulonglong, ci, ci, ci, C.POINTER(C.c_uint)]
    L.tinsel_hip_last_error.restype = C.c_char_p
    L.tinsel_pack_open.argtypes = [vp, C.c_size_t, C.POINTER(abi.SceneDesc), C.POINTER(abi.Camera), C.POINTER(abi.Options)]
    _lib = L
    return L


EXPORTED_SYMBOLS = [
    "tinsel_hip_create", "tinsel_hip_destroy", "tinsel_hip_init", "tinsel_hip_init_external", "tinsel_hip_render",
    "tinsel_hip_render_async", "tinsel_hip_accum_device_ptr", "tinsel_hip_read_accum", "tinsel_hip_set_shard",
    "tinsel_hip_set_pipeline", "tinsel_hip_set_pass_index", "tinsel_hip_get_pass_index", "tinsel_hip_stats",
    "tinsel_hip_reset_stats", "tinsel_hip_stats_detail", "tinsel_hip_set_detail_counters", "tinsel_hip_kernel_times", "tinsel_hip_kernel_time_bytes",
    "tinsel_hip_enable_kernel_timing", "tinsel_hip_set_batch_paths", "tinsel_hip_stack_entries",
    "tinsel_hip_nee_per_path", "tinsel_hip_last_error", "tinsel_pack_open", "tinsel_hip_read_batch_radiance", "tinsel_hip_leaf",
    "tinsel_hip_write_accum", "tinsel_hip_reserve", "tinsel_hip_set_russian_roulette", "tinsel_hip_set_mesh_bvh", "tinsel_hip_present", "tinsel_hip_present_async", "tinsel_hip_present_device_ptr", "tinsel_image_quantize_rgb8",
    "tinsel_hip_walked_prims", "tinsel_hip_queue_counts", "tinsel_hip_set_lookahead", "tinsel_hip_set_arithmetic", "tinsel_hip_get_arithmetic", "tinsel_hip_refit_mesh", "tinsel_hip_set_probe_sampling",
    "tinsel_hip_set_primitive_transform", "tinsel_hip_rebuild_scene",
    "tinsel_hip_group_create", "tinsel_hip_group_destroy", "tinsel_hip_group_init", "tinsel_hip_group_render", "tinsel_hip_group_present",
    "tinsel_hip_group_size", "tinsel_hip_group_member", "tinsel_hip_group_set_lookahead", "tinsel_hip_ubench",
    "tinsel_hip_selftest_arith", "tinsel_hip_selftest_sort", "tinsel_hip_selftest_scan", "tinsel_hip_plan_regions",
    "tinsel_hip_tuning_init", "tinsel_hip_create_tuned", "tinsel_hip_set_tuning", "tinsel_hip_get_tuning", "tinsel_hip_group_create_tuned",
    "tinsel_hip_comm_unique_id", "tinsel_hip_comm_init", "tinsel_hip_comm_size", "tinsel_hip_comm_reduce_accum",
]


def _check(rc, what):
    if rc != 0:
        msg = load_library().tinsel_hip_last_error()
        raise TinselHipError("%s failed: %s" % (what, msg.decode() if msg else "?"))


class Scene:
    """A scene as the C-ABI sees it: a `SceneDesc` plus the camera/options the scene file carried.

    Scenes come from scene packs (DESIGN.md "Scene pack"): one relocatable blob written by the
    reference's own loader + Scene::Build (tests/golden/make_golden.py).
    """

    def __init__(self, blob: bytes):
        L = load_library()
        self._buf = C.create_string_buffer(blob, len(blob))     # owned, writable: pack_open relocates in place
        self.desc = abi.SceneDesc()
        self.camera = abi.Camera()
        self.options = abi.Options()
        _check(L.tinsel_pack_open(self._buf, len(blob), C.byref(self.desc), C.byref(self.camera), C.byref(self.options)),
               "tinsel_pack_open")

    @classmethod
    def load_pack(cls, path):
        with open(path, "rb") as fh:
            return cls(fh.read())

    @property
    def num_primitives(self):
        return self.desc.num_primitives


def _bytes_at(ptr, n):
    return C.string_at(ptr, n) if (ptr and n > 0) else b""


_libc = C.CDLL(None)
_libc.memcmp.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]


def _same_bytes(pa, pb, n):
    """memcmp of two host arrays in place (a frame's mesh arrays are tens of MB: no copies)"""
    if n <= 0 or pa == pb:
        return True
    return bool(pa) and bool(pb) and _libc.memcmp(pa, pb, n) == 0


def _same_nodes(pa, pb, n):
    """two reference-format trees node by node.  Not just a byte compare: a leaf's rightIndex bits are whatever the reference builder's heap
    held (bvh.h:236-237 sets leftIndex and `leaf` only) and may differ from one load of the same file to the next."""
    if n <= 0 or _same_bytes(pa, pb, n*32):
        return True
    a = np.ctypeslib.as_array((C.c_uint32*(n*8)).from_address(pa)).reshape(n, 8)
    b = np.ctypeslib.as_array((C.c_uint32*(n*8)).from_address(pb)).reshape(n, 8)
    if not np.array_equal(a[:, :7], b[:, :7]):                      # bounds (6 floats as bits) + leftIndex
        return False
    leaf_a, leaf_b = a[:, 7] >> 31, b[:, 7] >> 31
    if not np.array_equal(leaf_a, leaf_b):
        return False
    inner = leaf_a == 0
    return bool(np.array_equal(a[inner, 7], b[inner, 7]))


def scene_delta(old: Scene, new: Scene):
    """What turns a renderer created from `old` into one for `new` WITHOUT re-uploading anything: ([(primitive, start, end)], the new scene
    BVH's nodes) when `new` is `old` with other primitive transforms -- the frames of a rigid animation, what the reference's batch mode
    re-loads and re-creates per frame (main.cpp:314-327) -- else None (geometry, materials, lights, sky or probe differ: re-create)."""
    a, b = old.desc, new.desc
    if a.num_primitives != b.num_primitives or bytes(a.sky_horizon) != bytes(b.sky_horizon) or bytes(a.sky_zenith) != bytes(b.sky_zenith):
        return None
    if (a.probe_valid, a.probe_width, a.probe_height) != (b.probe_valid, b.probe_width, b.probe_height):
        return None
    if a.probe_valid and not _same_bytes(a.probe_data, b.probe_data, a.probe_width*a.probe_height*16):
        return None
    P = a.num_primitives
    pa, pb = (abi.Primitive*P).from_address(a.primitives), (abi.Primitive*P).from_address(b.primitives)
    moves = []
    for i in range(P):
        x, y = pa[i], pb[i]
        mx, my = bytes(x.material), bytes(y.material)
        # (the material's parameters: bytes [0, 84) and [112, 128); between them padding and the bump-map texture -- a host pointer and sizes
        # the integrator never reads: bump mapping is dead code in the reference)
        if x.type != y.type or x.light_samples != y.light_samples or mx[:84] != my[:84] or mx[112:] != my[112:]:
            return None
        if x.type == abi.GEOM_SPHERE and x.geo.sphere.radius != y.geo.sphere.radius:
            return None
        if x.type == abi.GEOM_PLANE and bytes(x.geo.plane) != bytes(y.geo.plane):
            return None
        if x.type == abi.GEOM_MESH:
            g, h = x.geo.mesh, y.geo.mesh
            if (g.num_vertices, g.num_indices, g.num_nodes, g.area) != (h.num_vertices, h.num_indices, h.num_nodes, h.area):
                return None
            for f, n in (("positions", g.num_vertices*12), ("normals", g.num_vertices*12), ("indices", g.num_indices*4), ("cdf", (g.num_indices//3)*4)):
                if not _same_bytes(getattr(g, f), getattr(h, f), n):
                    return None
            if not _same_nodes(g.nodes, h.nodes, g.num_nodes):
                return None
        if bytes(x.start_transform) != bytes(y.start_transform) or bytes(x.end_transform) != bytes(y.end_transform):
            moves.append((i, abi.Transform.from_buffer_copy(bytes(y.start_transform)), abi.Transform.from_buffer_copy(bytes(y.end_transform))))
    nodes = (abi.BVHNode*b.num_bvh_nodes).from_buffer_copy(_bytes_at(b.bvh_nodes, b.num_bvh_nodes*C.sizeof(abi.BVHNode)))
    return moves, nodes


class HipRenderer:
    """`Renderer` (render.h:66-73) implemented by the gfx950 streaming path tracer."""

    def __init__(self, scene: Scene, device: int = 0, tuning: "abi.Tuning | None" = None):
        L = load_library()
        self._L = L
        tuning = tuning if tuning is not None else DEFAULT_TUNING
        if tuning is None:
            self._h = L.tinsel_hip_create(C.byref(scene.desc), device)
        else:
            self._h = L.tinsel_hip_create_tuned(C.byref(scene.desc), device, C.byref(tuning))
        if not self._h:
            msg = L.tinsel_hip_last_error()
            raise TinselHipError("tinsel_hip_create failed: %s" % (msg.decode() if msg else "?"))
        self.device = device
        self.width = self.height = 0
        self._accum_tensor = None

    # -- Renderer interface ------------------------------------------------
    def init(self, width, height, accum_tensor=None):
        """Renderer::Init: (re)allocate + zero the accumulator.  `accum_tensor`: optional torch
        float32 device tensor [H,W,4] to use as the accumulator (for an RCCL reduce)."""
        if accum_tensor is not None:
            assert tuple(accum_tensor.shape) == (height, width, 4) and accum_tensor.is_contiguous()
            self._accum_tensor = accum_tensor
            _check(self._L.tinsel_hip_init_external(self._h, width, height, accum_tensor.data_ptr()), "tinsel_hip_init_external")
        else:
            self._accum_tensor = None
            _check(self._L.tinsel_hip_init(self._h, width, height), "tinsel_hip_init")
        self.width, self.height = width, height

    def render(self, camera, options, output=None, passes=1, readback=True):
        """Renderer::Render: add `passes` samples per pixel; returns the running sum [H,W,4] (rgb*w, w)."""
        if output is None and readback:
            output = np.empty((options.height, options.width, 4), np.float32)
        ptr = output.ctypes.data_as(C.c_void_p) if (readback and output is not None) else None
        _check(self._L.tinsel_hip_render(self._h, C.byref(camera), C.byref(options), ptr, passes), "tinsel_hip_render")
        return output

    Init = init
    Render = render

    # -- beyond the reference interface -------------------------------------
    def render_async(self, camera, options, passes=1, stream=None):
        _check(self._L.tinsel_hip_render_async(self._h, C.byref(camera), C.byref(options), passes, stream), "tinsel_hip_render_async")

    def read_accum(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        _check(self._L.tinsel_hip_read_accum(self._h, out.ctypes.data_as(C.c_void_p)), "tinsel_hip_read_accum")
        return out

    def set_mesh_bvh(self, mode):
        """abi.BVH_REFERENCE (the reference's host-built trees, parity path) or abi.BVH_LBVH (rebuild large meshes on
        the device); returns the device build time in ms."""
        ms = C.c_double(0.0)
        _check(self._L.tinsel_hip_set_mesh_bvh(self._h, int(mode), C.byref(ms)), "tinsel_hip_set_mesh_bvh")
        return ms.value

    def write_accum(self, accum, next_pass_index):
        """Restores a saved accumulator [H,W,4] and the index of the next pass (resume a progressive render)."""
        accum = np.ascontiguousarray(accum, np.float32)
        assert accum.shape == (self.height, self.width, 4)
        _check(self._L.tinsel_hip_write_accum(self._h, accum.ctypes.data_as(C.c_void_p), int(next_pass_index)), "tinsel_hip_write_accum")

    def present(self, options, nlm_width=0, nlm_falloff=200.0, readback=True):
        """The display stage of the reference's frame loop (main.cpp:258-282) on the device accumulator:
        normalise by the filter weight, ToneMap, LinearToSrgb, optional NonLocalMeansFilter.
        Returns the float image [H,W,4] main.cpp presents / hands to WritePng."""
        out = np.empty((self.height, self.width, 4), np.float32) if readback else None
        ptr = out.ctypes.data_as(C.c_void_p) if readback else None
        _check(self._L.tinsel_hip_present(self._h, C.byref(options), int(nlm_width), float(nlm_falloff), ptr), "tinsel_hip_present")
        return out

    @property
    def accum_tensor(self):
        return self._accum_tensor

    def accum_device_ptr(self):
        return self._L.tinsel_hip_accum_device_ptr(self._h)

    def set_shard(self, rank, world, tile=32):
        _check(self._L.tinsel_hip_set_shard(self._h, rank, world, tile), "tinsel_hip_set_shard")

    def set_pipeline(self, pipeline):
        _check(self._L.tinsel_hip_set_pipeline(self._h, pipeline), "tinsel_hip_set_pipeline")

    # -- the process-per-GPU arm of the one collective (tinsel_hip_comm_*): the library's own ncclReduce, the id carried by the host language
    @staticmethod
    def comm_unique_id():
        """bytes of an RCCL unique id (rank 0 makes it, the host language hands it to the other ranks)"""
        buf = (C.c_ubyte*abi.COMM_ID_BYTES)()
        _check(load_library().tinsel_hip_comm_unique_id(buf, abi.COMM_ID_BYTES), "tinsel_hip_comm_unique_id")
        return bytes(buf)

    def comm_init(self, id_bytes, rank, world):
        """collective: every rank calls it with rank 0's id"""
        buf = (C.c_ubyte*abi.COMM_ID_BYTES).from_buffer_copy(bytes(id_bytes))
        _check(self._L.tinsel_hip_comm_init(self._h, buf, int(rank), int(world)), "tinsel_hip_comm_init")

    def comm_size(self):
        """ranks of this renderer's RCCL communicator as RCCL counts them (ncclCommCount); 0: none"""
        return int(self._L.tinsel_hip_comm_size(self._h)) if hasattr(self._L, "tinsel_hip_comm_size") else 0

    def comm_reduce_accum(self, out_ptr, root=0, stream=None):
        """sum of every rank's accumulator into device memory `out_ptr` (W*H*4 floats) on `root`; waits for `stream`"""
        _check(self._L.tinsel_hip_comm_reduce_accum(self._h, out_ptr, int(root), stream), "tinsel_hip_comm_reduce_accum")

    def set_tuning(self, tuning=None, **fields):
        """tinsel_hip_set_tuning: the per-render fields of an abi.Tuning (or of the tuning in force with `fields` replaced); the create-time
        fields (abi.Tuning.CREATE_FIELDS) belong to create_gpu_renderer(scene, device, tuning)."""
        t = (tuning if tuning is not None else self.get_tuning()).replace(**fields)
        _check(self._L.tinsel_hip_set_tuning(self._h, C.byref(t)), "tinsel_hip_set_tuning")

    def get_tuning(self):
        t = abi.Tuning()
        _check(self._L.tinsel_hip_get_tuning(self._h, C.byref(t)), "tinsel_hip_get_tuning")
        return t

    def set_arithmetic(self, mode):
        """abi.ARITH_EXACT (default: bit-identical to the CPU reference) or abi.ARITH_FAST (FMA / rcp / hardware transcendentals:
        the reference's own -ffast-math trade, inside the 1e-3 L2 bar but not bit-reproducible)."""
        _check(self._L.tinsel_hip_set_arithmetic(self._h, int(mode)), "tinsel_hip_set_arithmetic")

    def refit_mesh(self, primitive, positions, normals=None):
        """New vertex positions [V,3] (and optionally normals) for `primitive`'s mesh, same topology: boxes refitted on the device."""
        positions = np.ascontiguousarray(positions, np.float32)
        nptr = None
        if normals is not None:
            normals = np.ascontiguousarray(normals, np.float32)
            nptr = normals.ctypes.data_as(C.c_void_p)
        _check(self._L.tinsel_hip_refit_mesh(self._h, int(primitive), positions.ctypes.data_as(C.c_void_p), int(positions.shape[0]), nptr),
               "tinsel_hip_refit_mesh")

    def set_primitive_transform(self, index, start, end=None):
        """Primitive `index` moves: new start / end transforms (abi.Transform; end defaults to start).  Follow with rebuild_scene()."""
        end = start if end is None else end
        _check(self._L.tinsel_hip_set_primitive_transform(self._h, int(index), C.byref(start), C.byref(end)), "tinsel_hip_set_primitive_transform")

    def rebuild_scene(self, nodes=None):
        """The scene-level BVH from the primitives as they are now: from `nodes` (a ctypes array of abi.BVHNode, e.g. the reference's own
        Scene::Build -- bit-identical to a fresh renderer) or, without, built on the device.  Returns the device build time in ms."""
        ms = C.c_double(0.0)
        if nodes is None:
            rc = self._L.tinsel_hip_rebuild_scene(self._h, abi.SCENE_BVH_DEVICE, None, 0, C.byref(ms))
        else:
            rc = self._L.tinsel_hip_rebuild_scene(self._h, abi.SCENE_BVH_NODES, C.cast(nodes, C.c_void_p), len(nodes), C.byref(ms))
        _check(rc, "tinsel_hip_rebuild_scene")
        return ms.value

    def update_scene(self, old: Scene, new: Scene):
        """The next frame of an animation on THIS renderer: when `new` is `old` (the scene this renderer holds) with other primitive transforms,
        the moved primitives' records are rewritten in place and the scene level is rebuilt from `new`'s own nodes (the reference's
        Scene::Build for that frame) -- bit-identical to a renderer created from `new`, nothing re-uploaded.  False (and nothing changed)
        when the frames differ in more than that: the caller re-creates, as the reference always does (main.cpp:318-327)."""
        d = scene_delta(old, new)
        if d is None:
            return False
        moves, nodes = d
        for i, s, e in moves:
            self.set_primitive_transform(i, s, e)
        self.rebuild_scene(nodes)
        return True

    def set_probe_sampling(self, mode):
        """abi.PROBE_CDF (the reference's binary searches: sample-identical) or abi.PROBE_ALIAS (O(1) alias table: same distribution)."""
        _check(self._L.tinsel_hip_set_probe_sampling(self._h, int(mode)), "tinsel_hip_set_probe_sampling")

    def set_lookahead(self, on):
        """Trace the next call's passes while this call's image is copied out (tinsel_hip_set_lookahead); images unchanged.
        abi.LOOKAHEAD_PIN_OUTPUT also page-locks the output array in place (it must then outlive the renderer / the next init)."""
        _check(self._L.tinsel_hip_set_lookahead(self._h, int(on)), "tinsel_hip_set_lookahead")

    def set_russian_roulette(self, start_bounce):
        """0 (default) = none, as the reference; b > 0 = roulette from bounce b on (unbiased, not sample-identical)."""
        _check(self._L.tinsel_hip_set_russian_roulette(self._h, int(start_bounce)), "tinsel_hip_set_russian_roulette")

    def reserve(self, passes, max_depth):
        """Pre-allocates the path buffers for renders of up to `passes` passes at `max_depth`."""
        _check(self._L.tinsel_hip_reserve(self._h, int(passes), int(max_depth)), "tinsel_hip_reserve")

    def get_pass_index(self):
        """Index of the next pass (its seed is the (index+1)-th output of Random(1).Rand())."""
        return int(self._L.tinsel_hip_get_pass_index(self._h))

    def set_pass_index(self, i):
        _check(self._L.tinsel_hip_set_pass_index(self._h, i), "tinsel_hip_set_pass_index")

    def set_batch_paths(self, n):
        _check(self._L.tinsel_hip_set_batch_paths(self._h, n), "tinsel_hip_set_batch_paths")

    def set_detail_counters(self, on):
        _check(self._L.tinsel_hip_set_detail_counters(self._h, int(on)), "tinsel_hip_set_detail_counters")

    def enable_kernel_timing(self, on):
        _check(self._L.tinsel_hip_enable_kernel_timing(self._h, int(on)), "tinsel_hip_enable_kernel_timing")

    def kernel_times(self):
        """{kernel: (launches, sum of their durations in ms, union of their intervals in ms)}; the union is smaller than the sum where a
        call's chunks overlap on two streams.  The record stride is what the LIBRARY says (tinsel_hip_kernel_time_bytes: another build may
        be loaded through TINSEL_HIP_LIB for an A/B); a library too old to say (before round 5: 36- or 40-byte records, nothing to tell them
        apart but the payload) is refused rather than guessed at."""
        if not hasattr(self._L, "tinsel_hip_kernel_time_bytes"):
            raise TinselHipError("%s predates tinsel_hip_kernel_time_bytes: its timing records cannot be read reliably" % LIB_PATH)
        size = self._L.tinsel_hip_kernel_time_bytes()
        if size not in (C.sizeof(abi.KernelTime), C.sizeof(abi.KernelTimeV1)):
            raise TinselHipError("tinsel_hip_kernel_time_bytes() = %d: a record layout this binding does not know" % size)
        raw = (C.c_ubyte * (16 * C.sizeof(abi.KernelTime)))()
        n = self._L.tinsel_hip_kernel_times(self._h, C.cast(raw, C.POINTER(abi.KernelTime)), 16)
        if n < 0:
            _check(n, "tinsel_hip_kernel_times")
        data = bytes(raw)
        T = abi.KernelTime if size == C.sizeof(abi.KernelTime) else abi.KernelTimeV1
        out = {}
        for i in range(n):
            rec = T.from_buffer_copy(data[i*size:(i + 1)*size])
            out[rec.name.decode()] = (rec.launches, rec.total_ms, rec.busy_ms if T is abi.KernelTime else rec.total_ms)
        return out

    def stats(self):
        names = ["rays", "samples", "internal_visits", "tri_tests", "prim_tests", "shadow_rays", "_6", "_7"]
        out = np.zeros(8, np.uint64)
        _check(self._L.tinsel_hip_stats_detail(self._h, out.ctypes.data_as(C.c_void_p)), "tinsel_hip_stats_detail")
        return dict(zip(names, (int(v) for v in out)))

    def batch_radiance(self, passes, height, width):
        """Per-path radiance [passes,H,W,3] of the most recent batch (test hook)."""
        n = passes*height*width
        out = np.empty((n, 4), np.float32)
        got = self._L.tinsel_hip_read_batch_radiance(self._h, out.ctypes.data_as(C.c_void_p), n)
        if got != n:
            raise TinselHipError("batch holds %d paths, wanted %d" % (got, n))
        return out[:, :3].reshape(passes, height, width, 3).copy()

    def leaf(self, op, index, n, out_stride, rows=None, seeds=None, camera=None, width=0, height=0):
        """Test hook: device leaf function `op` on `n` rows (see include/tinsel_hip.h: tinsel_hip_leaf)."""
        out = np.zeros((n, out_stride), np.float32)
        rp, rs = None, 0
        if rows is not None:
            rows = np.ascontiguousarray(rows, np.float32)
            rp, rs = rows.ctypes.data_as(C.c_void_p), rows.shape[1]
        sp = None
        if seeds is not None:
            seeds = np.ascontiguousarray(seeds, np.uint32)
            sp = seeds.ctypes.data_as(C.c_void_p)
        _check(self._L.tinsel_hip_leaf(self._h, op, index, n, rp, rs, sp, out.ctypes.data_as(C.c_void_p), out_stride,
                                       C.byref(camera) if camera is not None else None, width, height), "tinsel_hip_leaf")
        return out

    def reset_stats(self):
        self._L.tinsel_hip_reset_stats(self._h)

    @property
    def stack_entries(self):
        return self._L.tinsel_hip_stack_entries(self._h)

    @property
    def nee_per_path(self):
        return self._L.tinsel_hip_nee_per_path(self._h)

    @property
    def walked_prims(self):
        """Primitives whose mesh BVH the dedicated k_walk kernel traverses (0: all meshes are walked inline)."""
        return self._L.tinsel_hip_walked_prims(self._h)

    def queue_counts(self, max_bounces=64):
        """(paths alive at the start of each bounce, paths with shadow rays at each bounce) of the last batch."""
        out = (C.c_uint32*(2*max_bounces))()
        n = self._L.tinsel_hip_queue_counts(self._h, out, max_bounces)
        _check(min(n, 0), "tinsel_hip_queue_counts")
        return list(out[:n]), list(out[max_bounces:max_bounces + n])

    def close(self):
        if self._h:
            self._L.tinsel_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipRendererGroup:
    """`Renderer` (render.h:66-73) over several GPUs of one node: the C-ABI's tinsel_hip_group (one host thread per
    device inside the library, pixel-tile shards, ONE RCCL reduce of the accumulator per read-back)."""

    def __init__(self, scene: Scene, num_gpus: int = 0, tile: int = 64, tuning: "abi.Tuning | None" = None):
        L = load_library()
        self._L = L
        tuning = tuning if tuning is not None else DEFAULT_TUNING
        if tuning is None:
            self._h = L.tinsel_hip_group_create(C.byref(scene.desc), num_gpus, tile)
        else:
            self._h = L.tinsel_hip_group_create_tuned(C.byref(scene.desc), num_gpus, tile, C.byref(tuning))
        if not self._h:
            msg = L.tinsel_hip_last_error()
            raise TinselHipError("tinsel_hip_group_create failed: %s" % (msg.decode() if msg else "?"))
        self.width = self.height = 0

    @property
    def num_gpus(self):
        return self._L.tinsel_hip_group_size(self._h)

    def init(self, width, height):
        _check(self._L.tinsel_hip_group_init(self._h, width, height), "tinsel_hip_group_init")
        self.width, self.height = width, height

    def render(self, camera, options, output=None, passes=1, readback=True):
        if output is None and readback:
            output = np.empty((options.height, options.width, 4), np.float32)
        ptr = output.ctypes.data_as(C.c_void_p) if (readback and output is not None) else None
        _check(self._L.tinsel_hip_group_render(self._h, C.byref(camera), C.byref(options), ptr, passes), "tinsel_hip_group_render")
        return output

    Init = init
    Render = render

    def set_lookahead(self, mode):
        """Trace and reduce the next calls while this call's image is copied out (tinsel_hip_group_set_lookahead); images unchanged.
        mode: abi.LOOKAHEAD_OFF / ON / PIN_OUTPUT (or a bool)."""
        _check(self._L.tinsel_hip_group_set_lookahead(self._h, int(mode)), "tinsel_hip_group_set_lookahead")

    def present(self, options, nlm_width=0, nlm_falloff=200.0):
        out = np.empty((self.height, self.width, 4), np.float32)
        _check(self._L.tinsel_hip_group_present(self._h, C.byref(options), int(nlm_width), float(nlm_falloff), out.ctypes.data_as(C.c_void_p)),
               "tinsel_hip_group_present")
        return out

    def member_stats(self, rank):
        """rays / samples counters of member `rank` (tinsel_hip_stats_detail on its renderer)."""
        m = self._L.tinsel_hip_group_member(self._h, rank)
        out = np.zeros(8, np.uint64)
        _check(self._L.tinsel_hip_stats_detail(m, out.ctypes.data_as(C.c_void_p)), "tinsel_hip_stats_detail")
        return {"rays": int(out[0]), "samples": int(out[1]), "shadow_rays": int(out[5])}

    def close(self):
        if self._h:
            self._L.tinsel_hip_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


UBENCH_COPY, UBENCH_GATHER_BEYOND_CACHE, UBENCH_GATHER_TREE, UBENCH_GATHER_L2 = 0, 1, 2, 3


def ubench(kind, nbytes, steps=64, device=0):
    """tinsel_hip_ubench: (milliseconds, units) of one timed launch -- units = bytes moved (copy) or 64-B records visited."""
    L = load_library()
    ms, units = C.c_double(0.0), C.c_double(0.0)
    _check(L.tinsel_hip_ubench(int(device), int(kind), int(nbytes), int(steps), C.byref(ms), C.byref(units)), "tinsel_hip_ubench")
    return ms.value, units.value


def plan_regions(slots, num_cus=256, nee_per_path=1, fused=True):
    """tinsel_hip_plan_regions: dict of how a batch of `slots` path slots would be cut (host arithmetic only, no GPU needed)."""
    L = load_library()
    out = (C.c_uint*6)()
    _check(L.tinsel_hip_plan_regions(int(slots), int(num_cus), int(nee_per_path), 1 if fused else 0, out), "tinsel_hip_plan_regions")
    return dict(zip(("num_regions", "region_len", "big_regions", "short_len", "grid", "max_regions"), (int(v) for v in out)))


def selftest_sort(keys, begin_bit=0, end_bit=64, device=0):
    """tinsel_hip_selftest_sort: the library's stable LSD radix sort of uint64 keys by their bits [begin_bit, end_bit); returns the sorted copy"""
    L = load_library()
    out = np.ascontiguousarray(keys, np.uint64).copy()
    _check(L.tinsel_hip_selftest_sort(int(device), out.ctypes.data_as(C.c_void_p), out.size, int(begin_bit), int(end_bit)), "tinsel_hip_selftest_sort")
    return out


def selftest_scan(values, device=0):
    """tinsel_hip_selftest_scan: the library's exclusive prefix sum of int32 values"""
    L = load_library()
    a = np.ascontiguousarray(values, np.int32)
    out = np.empty_like(a)
    _check(L.tinsel_hip_selftest_scan(int(device), a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), a.size), "tinsel_hip_selftest_scan")
    return out


def selftest_arith(op, variant=-1, device=0):
    """tinsel_hip_selftest_arith: (counts[4 + 256 by exponent field], first_bad) of the exhaustive 2^32-input comparison of a reciprocal (op 0) /
    square-root (op 1) sequence with the compiler's IEEE expansion; variant -1 = the one the library is built with."""
    L = load_library()
    counts, first = (C.c_ulonglong*260)(), C.c_uint(0)
    _check(L.tinsel_hip_selftest_arith(int(device), int(op), int(variant), counts, C.byref(first)), "tinsel_hip_selftest_arith")
    return [int(c) for c in counts], int(first.value)


def create_gpu_renderer(scene: Scene, device: int = 0, tuning: "abi.Tuning | None" = None) -> HipRenderer:
    """`Renderer* CreateGpuRenderer(const Scene*)` (render.h:79); `tuning`: an abi.Tuning (tinsel_hip_create_tuned), None = the defaults."""
    return HipRenderer(scene, device, tuning)


class use_tuning:
    """`with use_tuning(walk_min_tris=0, small_mesh_bytes=0): ...` -- renderers created inside the block without an explicit tuning get
    this one (DEFAULT_TUNING above).  Also usable as `use_tuning(...).__enter__()` / pytest's monkeypatch.setattr(renderer, "DEFAULT_TUNING", ..)."""

    def __init__(self, tuning=None, **fields):
        self.tuning = (tuning if tuning is not None else abi.Tuning()).replace(**fields)

    def __enter__(self):
        global DEFAULT_TUNING
        self._was = DEFAULT_TUNING
        DEFAULT_TUNING = self.tuning
        return self.tuning

    def __exit__(self, *exc):
        global DEFAULT_TUNING
        DEFAULT_TUNING = self._was
        return False


def resolve(accum):
    """Consumer-side normalisation of main.cpp:262-268: rgb / w (0 where w == 0)."""
    w = accum[..., 3:4]
    return np.where(w > 0, accum[..., :3] / np.where(w > 0, w, 1.0), 0.0)
