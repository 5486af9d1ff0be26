"""Image output of the reference's frame loop: WritePng (png.cpp:323-371) and PfmSave (pfm.cpp:70-85).

The float -> 8-bit step (a serial dither stream) is the library's `tinsel_image_quantize_rgb8`; the containers are
written here from their specifications, laid out as the reference's writers lay them out, so that the files
are byte-identical to the ones the reference produces from the same float image:

  PNG  8-bit RGB, no interlace, filter 0 on every row, ONE IDAT holding a zlib stream of STORED deflate blocks
       of at most 65535 bytes (header 0x08 0x1D), Adler-32, IEND.
  PFM  "PF\\n<W> <H>\\n-<max>\\n" then W*H*3 little-endian floats, rows in the order given.
"""
import ctypes as C
import struct
import zlib

import numpy as np

from .renderer import load_library, _check


def quantize_rgb8(image):
    """[H,W,4] float32 -> [H,W,3] uint8 exactly as WritePng quantises (png.cpp:331-343)."""
    image = np.ascontiguousarray(image, np.float32)
    h, w = image.shape[:2]
    out = np.empty((h, w, 3), np.uint8)
    _check(load_library().tinsel_image_quantize_rgb8(image.ctypes.data_as(C.c_void_p), w, h, out.ctypes.data_as(C.c_void_p)),
           "tinsel_image_quantize_rgb8")
    return out


def _chunk(tag, payload):
    return struct.pack(">I", len(payload)) + tag + payload + struct.pack(">I", zlib.crc32(tag + payload) & 0xffffffff)


def png_bytes(rgb8):
    """Encodes [H,W,3] uint8 the way the reference's PNG writer does (stored deflate blocks, one IDAT)."""
    rgb8 = np.ascontiguousarray(rgb8, np.uint8)
    h, w = rgb8.shape[:2]
    rows = np.zeros((h, w*3 + 1), np.uint8)       # filter-type byte 0 in front of every row
    rows[:, 1:] = rgb8.reshape(h, w*3)
    raw = rows.tobytes()
    stream = bytearray(b"\x08\x1d")
    for off in range(0, len(raw), 65535):
        block = raw[off:off + 65535]
        final = 1 if off + 65535 >= len(raw) else 0
        stream += struct.pack("<BHH", final, len(block), len(block) ^ 0xffff) + block
    stream += struct.pack(">I", zlib.adler32(raw) & 0xffffffff)
    ihdr = struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)
    return b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", bytes(stream)) + _chunk(b"IEND", b"")


def write_png(path, image):
    """WritePng(pixels, W, H, path): `image` is the float image the display stage produced."""
    with open(path, "wb") as f:
        f.write(png_bytes(quantize_rgb8(image)))


def _c_format_f(x):
    """printf("%f") of a float, including glibc's spelling of the non-finite values."""
    if np.isnan(x):
        return "-nan" if np.signbit(x) else "nan"
    if np.isinf(x):
        return "-inf" if x < 0 else "inf"
    return "%f" % x


def pfm_bytes(image):
    """PfmSave of a depth-1 image: `image` [H,W,3 or 4] float32 (the alpha channel is dropped)."""
    rgb = np.ascontiguousarray(np.asarray(image, np.float32)[..., :3])
    h, w = rgb.shape[:2]
    flat = rgb.reshape(-1)
    # std::max_element (pfm.cpp:81): `largest < x` comparisons, so NaNs after the first element are skipped
    if np.isnan(flat[0]) or np.isnan(flat).all():
        top = float(flat[0])
    else:
        top = float(np.nanmax(flat))
    return ("PF\n%d %d\n-%s\n" % (w, h, _c_format_f(top))).encode() + rgb.astype("<f4").tobytes()


def write_pfm(path, image):
    with open(path, "wb") as f:
        f.write(pfm_bytes(image))
