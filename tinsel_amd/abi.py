"""ctypes mirrors of include/tinsel_hip.h (which mirrors the reference PODs).

Reference layouts: Vec3 maths.h:214, Transform maths.h:575, BVHNode bvh.h:9,
Camera scene.h:11, Material scene.h:45, MeshGeometry scene.h:119,
Primitive scene.h:138, Filter render.h:13, Options render.h:50.
"""
import ctypes as C


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def __iter__(self):
        return iter((self.x, self.y, self.z))


class Vec4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]

    def __iter__(self):
        return iter((self.x, self.y, self.z, self.w))


class Transform(C.Structure):
    _fields_ = [("p", Vec3), ("r", Vec4), ("s", C.c_float)]


class BVHNode(C.Structure):
    _fields_ = [("lower", Vec3), ("upper", Vec3), ("left_index", C.c_uint32), ("right_index_leaf", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [("position", Vec3), ("rotation", Vec4), ("fov", C.c_float),
                ("shutter_start", C.c_float), ("shutter_end", C.c_float)]


class Texture(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32),
                ("depth", C.c_int32), ("_pad", C.c_int32)]


class Material(C.Structure):
    _fields_ = [("emission", Vec3), ("color", Vec3), ("absorption", Vec3),
                ("eta", C.c_float), ("metallic", C.c_float), ("subsurface", C.c_float),
                ("specular", C.c_float), ("roughness", C.c_float), ("specular_tint", C.c_float),
                ("anisotropic", C.c_float), ("sheen", C.c_float), ("sheen_tint", C.c_float),
                ("clearcoat", C.c_float), ("clearcoat_gloss", C.c_float), ("transmission", C.c_float),
                ("_pad0", C.c_int32), ("bump_map", Texture), ("bump", C.c_float), ("bump_tile", Vec3)]


class MeshGeometry(C.Structure):
    _fields_ = [("positions", C.c_void_p), ("normals", C.c_void_p), ("indices", C.c_void_p),
                ("nodes", C.c_void_p), ("cdf", C.c_void_p),
                ("num_vertices", C.c_int32), ("num_indices", C.c_int32), ("num_nodes", C.c_int32),
                ("area", C.c_float), ("id", C.c_uint64)]


class _Sphere(C.Structure):
    _fields_ = [("radius", C.c_float)]


class _Plane(C.Structure):
    _fields_ = [("plane", C.c_float * 4)]


class _Geo(C.Union):
    _fields_ = [("sphere", _Sphere), ("plane", _Plane), ("mesh", MeshGeometry)]


class Primitive(C.Structure):
    _fields_ = [("start_transform", Transform), ("end_transform", Transform),
                ("type", C.c_int32), ("_pad0", C.c_int32), ("geo", _Geo),
                ("material", Material), ("light_samples", C.c_int32), ("_pad1", C.c_int32)]


class Filter(C.Structure):
    _fields_ = [("type", C.c_int32), ("width", C.c_float), ("falloff", C.c_float), ("offset", C.c_float)]


class Options(C.Structure):
    _fields_ = [("mode", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("filter", Filter),
                ("exposure", C.c_float), ("limit", C.c_float), ("clamp", C.c_float),
                ("max_depth", C.c_int32), ("max_samples", C.c_int32)]

    def copy(self):
        o = Options()
        C.memmove(C.byref(o), C.byref(self), C.sizeof(Options))
        return o


class SceneDesc(C.Structure):
    _fields_ = [("primitives", C.c_void_p), ("num_primitives", C.c_int32), ("num_bvh_nodes", C.c_int32),
                ("bvh_nodes", C.c_void_p), ("sky_horizon", Vec3), ("sky_zenith", Vec3),
                ("probe_valid", C.c_int32), ("probe_width", C.c_int32), ("probe_height", C.c_int32),
                ("_pad", C.c_int32), ("probe_data", C.c_void_p), ("probe_pdf_x", C.c_void_p),
                ("probe_cdf_x", C.c_void_p), ("probe_pdf_y", C.c_void_p), ("probe_cdf_y", C.c_void_p)]


class PackHeader(C.Structure):
    _fields_ = [("magic", C.c_char * 8), ("version", C.c_uint32), ("num_primitives", C.c_uint32),
                ("num_bvh_nodes", C.c_uint32), ("num_meshes", C.c_uint32), ("total_bytes", C.c_uint64),
                ("off_primitives", C.c_uint64), ("off_bvh_nodes", C.c_uint64), ("off_probe_data", C.c_uint64),
                ("off_probe_pdf_x", C.c_uint64), ("off_probe_cdf_x", C.c_uint64),
                ("off_probe_pdf_y", C.c_uint64), ("off_probe_cdf_y", C.c_uint64),
                ("probe_width", C.c_int32), ("probe_height", C.c_int32),
                ("sky_horizon", Vec3), ("sky_zenith", Vec3), ("camera", Camera), ("options", Options),
                ("_reserved", C.c_uint8 * 48)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_uint32), ("total_ms", C.c_float), ("busy_ms", C.c_float)]


class KernelTimeV1(C.Structure):
    """tinsel_kernel_time as libraries built before round 4 wrote it (no busy_ms): renderer.HipRenderer.kernel_times"""
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_uint32), ("total_ms", C.c_float)]


class Tuning(C.Structure):
    """tinsel_hip_tuning (include/tinsel_hip.h): every choice between two code paths of the library as one plain struct.  Tuning() holds
    the defaults ("the library decides"); Tuning(walk_min_tris=0, small_mesh_bytes=0) overrides fields.  Nothing is read from the
    environment."""
    _fields_ = [("struct_bytes", C.c_uint32),
                ("flat_scan", C.c_int32), ("lds_scene", C.c_int32), ("walk", C.c_int32), ("inline_max_tris", C.c_int32), ("walk_min_tris", C.c_int32),
                ("small_mesh_bytes", C.c_int64), ("arena_lds_limit", C.c_int64),
                ("batch_paths", C.c_int64), ("grid_mult", C.c_int32), ("bounce_share", C.c_int32), ("repack", C.c_int32),
                ("tail_split", C.c_int32), ("tail_share", C.c_float), ("tail_divide", C.c_int32),
                ("shade_sorted", C.c_int32), ("overlap", C.c_int32), ("scene_walk", C.c_int32), ("swalk_lds", C.c_int32), ("accumulate", C.c_int32),
                ("walk_block", C.c_int32), ("walk_single", C.c_int32), ("walk_lds_stack", C.c_int32), ("walk_refill_min", C.c_int32), ("walk_leaf_min", C.c_int32),
                ("walk_grid_mult", C.c_int32), ("quads_in_scan", C.c_int32)]
    CREATE_FIELDS = ("flat_scan", "lds_scene", "walk", "inline_max_tris", "walk_min_tris", "small_mesh_bytes", "arena_lds_limit")
    _DEFAULTS = dict(flat_scan=-1, lds_scene=-1, walk=-1, inline_max_tris=-1, walk_min_tris=-1, small_mesh_bytes=-1, arena_lds_limit=-1,
                     batch_paths=0, grid_mult=0, bounce_share=-1, repack=-1, tail_split=-1, tail_share=0.0, tail_divide=4,
                     shade_sorted=-1, overlap=-1, scene_walk=-1, swalk_lds=-1, accumulate=0,
                     walk_block=0, walk_single=-1, walk_lds_stack=-1, walk_refill_min=0, walk_leaf_min=0, walk_grid_mult=0, quads_in_scan=-1)

    def __init__(self, **over):
        super().__init__()
        self.struct_bytes = C.sizeof(Tuning)
        for k, v in self._DEFAULTS.items():
            setattr(self, k, v)
        for k, v in over.items():
            if k not in self._DEFAULTS:
                raise TypeError("tinsel_hip_tuning has no field %r" % k)
            setattr(self, k, v)

    def replace(self, **over):
        t = Tuning(**{k: getattr(self, k) for k in self._DEFAULTS})
        for k, v in over.items():
            if k not in self._DEFAULTS:
                raise TypeError("tinsel_hip_tuning has no field %r" % k)
            setattr(t, k, v)
        return t

    def as_dict(self):
        return {k: getattr(self, k) for k in self._DEFAULTS}


ACCUMULATE_AUTO, ACCUMULATE_TILED, ACCUMULATE_WIDE, ACCUMULATE_PIPED = 0, 1, 2, 3
COMM_ID_BYTES = 128
MODE_NORMALS, MODE_COMPLEXITY, MODE_PATHTRACE = 0, 1, 2
BVH_REFERENCE, BVH_LBVH, BVH_PLOC = 0, 1, 2
SCENE_BVH_NODES, SCENE_BVH_DEVICE = 0, 1
ARITH_EXACT, ARITH_FAST = 0, 1
PROBE_CDF, PROBE_ALIAS = 0, 1
LOOKAHEAD_OFF, LOOKAHEAD_ON, LOOKAHEAD_PIN_OUTPUT = 0, 1, 2
FILTER_BOX, FILTER_GAUSSIAN = 0, 1
GEOM_SPHERE, GEOM_PLANE, GEOM_MESH = 0, 1, 2
PIPELINE_WAVEFRONT, PIPELINE_MEGAKERNEL, PIPELINE_WAVEFRONT_SPLIT, PIPELINE_AUTO, PIPELINE_WAVEFRONT_PAIRED = 0, 1, 2, 3, 4

assert C.sizeof(Transform) == 32 and C.sizeof(BVHNode) == 32 and C.sizeof(Camera) == 40
assert C.sizeof(Material) == 128 and C.sizeof(MeshGeometry) == 64 and C.sizeof(Primitive) == 272
assert C.sizeof(Filter) == 16 and C.sizeof(Options) == 48 and C.sizeof(PackHeader) == 256
assert Primitive.geo.offset == 72 and Primitive.material.offset == 136 and Primitive.light_samples.offset == 264
