"""ctypes access to the CHECKERS under oracle/ (test infrastructure only).

`RefOracle`  -> oracle/_ref/libtinsel_ref.so  : the reference's own CPU path (render.cpp:230 PathTrace),
                                                 compiled unmodified by oracle/Makefile.
`PortOracle` -> oracle/libtinsel_oracle.so    : the plain-C restatement (oracle/tinsel_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os

import numpy as np

from tinsel_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libtinsel_ref.so")
REF_FAST_SO = os.path.join(ROOT, "oracle", "_ref", "libtinsel_ref_fast.so")
PORT_SO = os.path.join(ROOT, "oracle", "libtinsel_oracle.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def have_ref():
    return os.path.exists(REF_SO)


def have_port():
    return os.path.exists(PORT_SO)


def _fp(a):
    return a.ctypes.data_as(C.c_void_p)


class _OracleBase:
    """Common surface of both checkers: the same extern "C" names with a prefix."""

    prefix = ""

    def __init__(self, path):
        self.lib = C.CDLL(path)
        L, p = self.lib, self.prefix
        f = getattr(L, p + "scene_load_pack")
        f.restype = C.c_void_p
        f.argtypes = [C.c_void_p, C.c_size_t]
        f = getattr(L, p + "scene_free")
        f.restype = None
        f.argtypes = [C.c_void_p]
        f = getattr(L, p + "scene_get")
        f.restype = None
        f.argtypes = [C.c_void_p, C.POINTER(abi.Camera), C.POINTER(abi.Options)]
        f = getattr(L, p + "render_seeded")
        f.restype = C.c_double
        f.argtypes = [C.c_void_p, C.POINTER(abi.Camera), C.POINTER(abi.Options), C.c_uint32, C.c_uint32,
                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        f = getattr(L, p + "pass_seed")
        f.restype = C.c_uint32
        f.argtypes = [C.c_uint32]
        f = getattr(L, p + "leaf_random")
        f.restype = None
        f.argtypes = [C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]

    # -- scenes -----------------------------------------------------------
    def load_pack(self, path_or_bytes):
        data = path_or_bytes
        if isinstance(data, str):
            with open(data, "rb") as fh:
                data = fh.read()
        buf = C.create_string_buffer(data, len(data))
        h = getattr(self.lib, self.prefix + "scene_load_pack")(buf, len(data))
        if not h:
            raise RuntimeError("oracle: bad scene pack")
        return h

    def free(self, h):
        getattr(self.lib, self.prefix + "scene_free")(h)

    def camera_options(self, h):
        cam, opt = abi.Camera(), abi.Options()
        getattr(self.lib, self.prefix + "scene_get")(h, C.byref(cam), C.byref(opt))
        return cam, opt

    # -- rendering --------------------------------------------------------
    def render_seeded(self, h, cam, opt, pass_begin=0, passes=1, window=None, threads=0,
                      want_accum=True, want_radiance=False):
        """Per-path-seeded oracle render.  Returns (accum[H,W,4] or None, radiance[passes,h,w,3] or None, seconds)."""
        W, H = opt.width, opt.height
        x0, y0, x1, y1 = window if window else (0, 0, W, H)
        if threads <= 0:
            threads = os.cpu_count() or 1
        accum = np.zeros((H, W, 4), np.float32) if want_accum else None
        rad = np.zeros((passes, y1 - y0, x1 - x0, 3), np.float32) if want_radiance else None
        secs = getattr(self.lib, self.prefix + "render_seeded")(
            h, C.byref(cam), C.byref(opt), pass_begin, passes, x0, y0, x1, y1,
            _fp(accum) if want_accum else None, _fp(rad) if want_radiance else None, threads)
        return accum, rad, secs

    def pass_seed(self, i):
        return getattr(self.lib, self.prefix + "pass_seed")(i)

    def leaf_random(self, seed, n):
        r = np.zeros(n, np.uint32)
        f = np.zeros(n, np.float32)
        getattr(self.lib, self.prefix + "leaf_random")(seed, n, _fp(r), _fp(f))
        return r, f


class RefOracle(_OracleBase):
    prefix = "ref_"

    def __init__(self, fast=False):
        super().__init__(REF_FAST_SO if fast else REF_SO)
        L = self.lib
        L.ref_scene_load_tin.restype = C.c_void_p
        L.ref_scene_load_tin.argtypes = [C.c_char_p]
        L.ref_scene_write_pack.restype = C.c_size_t
        L.ref_scene_write_pack.argtypes = [C.c_void_p, C.c_char_p]
        L.ref_scene_add_standin_mesh.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.POINTER(abi.Material), C.c_int]
        L.ref_scene_set_procedural_probe.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ref_scene_num_primitives.argtypes = [C.c_void_p]
        L.ref_scene_get_primitive.argtypes = [C.c_void_p, C.c_int, C.POINTER(abi.Primitive)]
        L.ref_scene_get_primitive.restype = None
        L.ref_make_filter.argtypes = [C.c_int, C.c_float, C.c_float, C.POINTER(abi.Filter)]
        L.ref_make_filter.restype = None
        L.ref_render_faithful.restype = C.c_double
        L.ref_render_faithful.argtypes = [C.c_void_p, C.POINTER(abi.Camera), C.POINTER(abi.Options), C.c_int, C.c_void_p]
        L.ref_leaf_camera_rays.argtypes = [C.POINTER(abi.Camera), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_leaf_camera_rays.restype = None
        L.ref_leaf_bsdf_eval.argtypes = [C.POINTER(abi.Material), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_leaf_bsdf_eval.restype = None
        L.ref_leaf_bsdf_sample.argtypes = [C.POINTER(abi.Material), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_leaf_bsdf_sample.restype = None
        L.ref_leaf_primitive_intersect.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_leaf_primitive_intersect.restype = None
        L.ref_leaf_primitive_sample.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_leaf_primitive_sample.restype = None
        L.ref_leaf_probe.argtypes = [C.c_void_p, C.c_int, C.c_void_p] + [C.c_void_p] * 5
        L.ref_leaf_probe.restype = None
        L.ref_hardware_threads.restype = C.c_int
        if hasattr(L, "ref_add_sample_agrees"):
            L.ref_add_sample_agrees.restype = C.c_int
            L.ref_add_sample_agrees.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_int]

    def load_tin(self, path):
        h = self.lib.ref_scene_load_tin(path.encode())
        if not h:
            raise RuntimeError("reference LoadTin failed: %s" % path)
        return h

    def write_pack(self, h, path):
        n = self.lib.ref_scene_write_pack(h, path.encode())
        if n == 0:
            raise RuntimeError("pack write failed: %s" % path)
        return n

    def primitive(self, h, i):
        p = abi.Primitive()
        self.lib.ref_scene_get_primitive(h, i, C.byref(p))
        return p

    def num_primitives(self, h):
        return self.lib.ref_scene_num_primitives(h)

    def make_filter(self, ftype, width, falloff):
        f = abi.Filter()
        self.lib.ref_make_filter(ftype, width, falloff, C.byref(f))
        return f

    def render_faithful(self, h, cam, opt, passes=1):
        out = np.zeros((opt.height, opt.width, 4), np.float32)
        secs = self.lib.ref_render_faithful(h, C.byref(cam), C.byref(opt), passes, _fp(out))
        return out, secs

    def camera_rays(self, cam, W, H, raster_xy):
        raster_xy = np.ascontiguousarray(raster_xy, np.float32)
        n = raster_xy.shape[0]
        out = np.zeros((n, 6), np.float32)
        self.lib.ref_leaf_camera_rays(C.byref(cam), W, H, n, _fp(raster_xy), _fp(out))
        return out

    def bsdf_eval(self, mat, rows):
        rows = np.ascontiguousarray(rows, np.float32)
        n = rows.shape[0]
        f = np.zeros((n, 3), np.float32)
        pdf = np.zeros(n, np.float32)
        self.lib.ref_leaf_bsdf_eval(C.byref(mat), n, _fp(rows), _fp(f), _fp(pdf))
        return f, pdf

    def bsdf_sample(self, mat, rows, seeds):
        rows = np.ascontiguousarray(rows, np.float32)
        seeds = np.ascontiguousarray(seeds, np.uint32)
        n = rows.shape[0]
        L = np.zeros((n, 3), np.float32)
        pdf = np.zeros(n, np.float32)
        typ = np.zeros(n, np.int32)
        st = np.zeros((n, 2), np.uint32)
        self.lib.ref_leaf_bsdf_sample(C.byref(mat), n, _fp(rows), _fp(seeds), _fp(L), _fp(pdf), _fp(typ), _fp(st))
        return L, pdf, typ, st

    def primitive_intersect(self, h, prim, rows):
        rows = np.ascontiguousarray(rows, np.float32)
        n = rows.shape[0]
        hit = np.zeros(n, np.int32)
        t = np.zeros(n, np.float32)
        nrm = np.zeros((n, 3), np.float32)
        self.lib.ref_leaf_primitive_intersect(h, prim, n, _fp(rows), _fp(hit), _fp(t), _fp(nrm))
        return hit, t, nrm

    def primitive_sample(self, h, prim, times, seeds):
        times = np.ascontiguousarray(times, np.float32)
        seeds = np.ascontiguousarray(seeds, np.uint32)
        n = times.shape[0]
        pos = np.zeros((n, 3), np.float32)
        nrm = np.zeros((n, 3), np.float32)
        st = np.zeros((n, 2), np.uint32)
        self.lib.ref_leaf_primitive_sample(h, prim, n, _fp(times), _fp(seeds), _fp(pos), _fp(nrm), _fp(st))
        return pos, nrm, st

    def probe(self, h, seeds):
        seeds = np.ascontiguousarray(seeds, np.uint32)
        n = seeds.shape[0]
        d = np.zeros((n, 3), np.float32)
        c = np.zeros((n, 3), np.float32)
        pdf = np.zeros(n, np.float32)
        pdf2 = np.zeros(n, np.float32)
        ev = np.zeros((n, 3), np.float32)
        self.lib.ref_leaf_probe(h, n, _fp(seeds), _fp(d), _fp(c), _fp(pdf), _fp(pdf2), _fp(ev))
        return d, c, pdf, pdf2, ev


def _bounds_api():
    def primitive_bounds(self, h, prim):
        """PrimitiveBounds (intersection.h:906-939): the leaf box Scene::Build gives the scene BVH builder -> (lower, upper)"""
        fn = getattr(self.lib, self.prefix + "primitive_bounds")
        fn.restype = None
        fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        out = np.zeros(6, np.float32)
        fn(h, prim, _fp(out))
        return out[:3].copy(), out[3:].copy()
    _OracleBase.primitive_bounds = primitive_bounds


_bounds_api()


class PortOracle(_OracleBase):
    prefix = "port_"

    def __init__(self):
        super().__init__(PORT_SO)
        L = self.lib
        L.port_render_sharded.restype = C.c_double
        L.port_render_sharded.argtypes = [C.c_void_p, C.POINTER(abi.Camera), C.POINTER(abi.Options), C.c_uint32,
                                          C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.port_render_normals.restype = None
        L.port_render_normals.argtypes = [C.c_void_p, C.POINTER(abi.Camera), C.POINTER(abi.Options), C.c_void_p]
        L.port_render_seeded_counts.restype = C.c_double
        L.port_render_seeded_counts.argtypes = [C.c_void_p, C.POINTER(abi.Camera), C.POINTER(abi.Options), C.c_uint32,
                                                C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                C.c_int, C.c_void_p]

    def render_seeded_counts(self, h, cam, opt, pass_begin=0, passes=1, window=None, threads=0):
        """Like render_seeded, also returns the traversal counters dict used for B_ray."""
        W, H = opt.width, opt.height
        x0, y0, x1, y1 = window if window else (0, 0, W, H)
        if threads <= 0:
            threads = os.cpu_count() or 1
        accum = np.zeros((H, W, 4), np.float32)
        counts = np.zeros(8, np.uint64)
        secs = self.lib.port_render_seeded_counts(h, C.byref(cam), C.byref(opt), pass_begin, passes, x0, y0, x1, y1,
                                                  _fp(accum), None, threads, _fp(counts))
        names = ["rays", "samples", "internal_visits", "tri_tests", "prim_tests", "shadow_rays", "node_fetches", "_"]
        return accum, dict(zip(names, (int(c) for c in counts))), secs


def _port_extra():
    def render_sharded(self, h, cam, opt, rank, world, tile=32, pass_begin=0, passes=1, threads=0):
        if threads <= 0:
            threads = os.cpu_count() or 1
        accum = np.zeros((opt.height, opt.width, 4), np.float32)
        counts = np.zeros(8, np.uint64)
        self.lib.port_render_sharded(h, C.byref(cam), C.byref(opt), pass_begin, passes, 0, 0, 0, 0, _fp(accum), None,
                                     threads, rank, world, tile, _fp(counts))
        return accum, int(counts[1])

    def render_normals(self, h, cam, opt):
        out = np.zeros((opt.height, opt.width, 4), np.float32)
        self.lib.port_render_normals(h, C.byref(cam), C.byref(opt), _fp(out))
        return out

    PortOracle.render_sharded = render_sharded
    PortOracle.render_normals = render_normals


_port_extra()


def _display_api():
    """Display stage (main.cpp:258-282, nlm.cpp, png.cpp, pfm.cpp) on both oracles: the reference's own code
    (ref_*) and the C restatement (port_*)."""
    def present(self, pixels, exposure=1.0, limit=1.5):
        pixels = np.ascontiguousarray(pixels, np.float32)
        out = np.empty_like(pixels)
        fn = getattr(self.lib, self.prefix + "present")
        fn.restype = None
        fn.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p]
        fn(_fp(pixels), pixels.size//4, exposure, limit, _fp(out))
        return out

    def nlm(self, image, falloff=200.0, radius=1):
        image = np.ascontiguousarray(image, np.float32)
        out = np.empty_like(image)
        fn = getattr(self.lib, self.prefix + "nlm")
        fn.restype = None
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int]
        fn(_fp(image), _fp(out), image.shape[1], image.shape[0], falloff, radius)
        return out

    def write_png(self, image, path):
        image = np.ascontiguousarray(image, np.float32)
        self.lib.ref_write_png.restype = None
        self.lib.ref_write_png.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p]
        self.lib.ref_write_png(_fp(image), image.shape[1], image.shape[0], path.encode())

    def pfm_save(self, rgb, path):
        rgb = np.ascontiguousarray(rgb, np.float32)
        assert rgb.shape[-1] == 3
        self.lib.ref_pfm_save.restype = None
        self.lib.ref_pfm_save.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p]
        self.lib.ref_pfm_save(_fp(rgb), rgb.shape[1], rgb.shape[0], path.encode())

    def quantize_rgb8(self, image):
        image = np.ascontiguousarray(image, np.float32)
        out = np.empty(image.shape[:2] + (3,), np.uint8)
        self.lib.port_quantize_rgb8.restype = None
        self.lib.port_quantize_rgb8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        self.lib.port_quantize_rgb8(_fp(image), image.shape[1], image.shape[0], _fp(out))
        return out

    for cls in (RefOracle, PortOracle):
        cls.present = present
        cls.nlm = nlm
    RefOracle.write_png = write_png
    RefOracle.pfm_save = pfm_save
    PortOracle.quantize_rgb8 = quantize_rgb8


_display_api()


def png_pixels(data):
    """Decodes the 8-bit RGB pixels of a PNG file (any encoder) -> [H,W,3] uint8; filter type 0 rows only."""
    import struct
    import zlib
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    off, idat, w, h = 8, b"", 0, 0
    while off < len(data):
        n, tag = struct.unpack(">I4s", data[off:off + 8])
        body = data[off + 8:off + 8 + n]
        if tag == b"IHDR":
            w, h = struct.unpack(">II", body[:8])
        elif tag == b"IDAT":
            idat += body
        off += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w*3 + 1)
    assert (rows[:, 0] == 0).all()
    return rows[:, 1:].reshape(h, w, 3).copy()


def image_l2(a, b):
    """Per-pixel L2 of SURVEY.md 8c: sqrt(mean_px ||rgb_a/w_a - rgb_b/w_b||^2)."""
    wa = np.where(a[..., 3:4] > 0, a[..., 3:4], 1.0)
    wb = np.where(b[..., 3:4] > 0, b[..., 3:4], 1.0)
    d = a[..., :3] / wa - b[..., :3] / wb
    return float(np.sqrt(np.mean(np.sum(d.astype(np.float64) ** 2, axis=-1))))
