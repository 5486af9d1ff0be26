"""CPU tests of the display stage (the frame loop around the boundary, main.cpp:258-312).

The oracle's C restatement (oracle/tinsel_oracle.c: port_present / port_nlm / port_quantize_rgb8) and the host-side
pieces of the product (tn_powf.h compiled for the host, tinsel_image_quantize_rgb8, the PNG / PFM writers of
tinsel_amd.display) against tests/golden/display.golden.npz -- outputs of the reference's OWN ToneMap /
LinearToSrgb / NonLocalMeansFilter / WritePng / PfmSave (tests/golden/make_display.py)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests.oracle_api import GOLDEN, PortOracle, have_port, png_pixels

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "display.golden.npz"))


def same_bits(a, b):
    """Bit-identical where finite, NaN where NaN (payloads of invalid-operation NaNs are not part of the contract)."""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    nan = np.isnan(a)
    return np.array_equal(nan, np.isnan(b)) and np.array_equal(a.view(np.uint32)[~nan], b.view(np.uint32)[~nan])


needs_port = pytest.mark.skipif(not have_port(), reason="oracle/libtinsel_oracle.so not built")


@needs_port
@pytest.mark.parametrize("tag", ["a", "b"])
def test_port_present_matches_reference(gold, tag):
    P = PortOracle()
    out = P.present(gold["accum_" + tag], float(gold["exposure_" + tag]), 1.5)
    assert same_bits(out, gold["filtered_" + tag])


@needs_port
@pytest.mark.parametrize("tag", ["a", "b"])
def test_port_nlm_matches_reference(gold, tag):
    P = PortOracle()
    assert same_bits(P.nlm(gold["nlm_in_" + tag], 200.0, 1), gold["nlm_r1_" + tag])
    assert same_bits(P.nlm(gold["nlm_in_" + tag], 50.0, 2), gold["nlm_r2_" + tag])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_png_file_is_byte_identical_to_the_reference_writer(gold, tag):
    """tinsel_image_quantize_rgb8 (the serial dither stream of WritePng) + the container written by
    tinsel_amd.display == the file the reference's WritePng wrote from the same float image."""
    from tinsel_amd.display import png_bytes, quantize_rgb8
    filt = gold["filtered_" + tag]
    rgb = quantize_rgb8(filt)
    ref_file = gold["png_" + tag].tobytes()
    assert np.array_equal(rgb, png_pixels(ref_file))
    mine = png_bytes(rgb)
    assert mine == ref_file
    assert np.array_equal(png_pixels(mine), rgb)       # and it is a valid zlib stream / PNG


@needs_port
@pytest.mark.parametrize("tag", ["a", "b"])
def test_port_quantize_matches_reference_png(gold, tag):
    P = PortOracle()
    assert np.array_equal(P.quantize_rgb8(gold["filtered_" + tag]), png_pixels(gold["png_" + tag].tobytes()))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_pfm_file_is_byte_identical_to_the_reference_writer(gold, tag):
    from tinsel_amd.display import pfm_bytes
    assert pfm_bytes(gold["filtered_" + tag]) == gold["pfm_" + tag].tobytes()


def test_powf_restatement_equals_host_libm(tmp_path):
    """tn_powf.h (the text the device compiles) built for the host: bit-identical to this host's powf for the two
    display exponents on 8 M floats spread over the whole non-negative range plus the special values.  (The
    exhaustive sweep over all 2^31 non-negative floats x 5 exponents is recorded in DESIGN.md.)"""
    src = tmp_path / "p.cpp"
    src.write_text('#include "tn_powf.h"\nextern "C" void f(int n, const float* x, float y, float* o)'
                   '{ for (int i = 0; i < n; ++i) o[i] = tn::m_powf(x[i], y); }\n')
    so = tmp_path / "p.so"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I", os.path.join(ROOT, "tinsel_amd", "csrc"),
                    "-o", str(so), str(src)], check=True)
    mine = C.CDLL(str(so))
    ref_src = tmp_path / "r.c"
    ref_src.write_text("#include <math.h>\nvoid g(int n, const float* x, float y, float* o){ for (int i = 0; i < n; ++i) o[i] = powf(x[i], y); }\n")
    ref_so = tmp_path / "r.so"
    subprocess.run(["gcc", "-O1", "-fno-builtin", "-shared", "-fPIC", "-o", str(ref_so), str(ref_src), "-lm"], check=True)
    ref = C.CDLL(str(ref_so))
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 0x7f800000, 8_000_000, dtype=np.uint32)
    bits[:8] = [0, 1, 0x007fffff, 0x00800000, 0x3f800000, 0x7f7fffff, 0x7f800000, 0x3b83126f]
    x = bits.view(np.float32)
    x = np.concatenate([x, np.array([-0.0, -1.5, np.nan, -np.inf], np.float32), rng.random(1_000_000, dtype=np.float32)])
    for y in (2.2, 1.0/2.2):
        a = np.empty_like(x)
        b = np.empty_like(x)
        mine.f(len(x), x.ctypes.data_as(C.c_void_p), C.c_float(y), a.ctypes.data_as(C.c_void_p))
        ref.g(len(x), x.ctypes.data_as(C.c_void_p), C.c_float(y), b.ctypes.data_as(C.c_void_p))
        assert same_bits(a, b), "powf(x, %g)" % y
