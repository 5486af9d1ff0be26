#!/usr/bin/env python3
"""Regenerates tests/golden/display.golden.npz from the reference's OWN display-stage code (oracle/_ref:
ToneMap / LinearToSrgb of main.cpp:262-271, NonLocalMeansFilter, WritePng, PfmSave).  Runs where
/root/reference is mounted; the GPU box uses the committed file.

  accum_a      = the cornell fixture's accumulated framebuffer (tests/golden/cornell.golden.npz)
  accum_b      = a synthetic 40x56 framebuffer with the awkward cases: zero weight (0/0 and x/0), negative and huge
                 radiance, denormals, values straddling the tone curve's 0.004 toe
  filtered_*   = g_filtered   (exposure 1.0 / 0.37)
  nlm_*        = NonLocalMeansFilter(filtered, falloff, radius) for (200, 1) and (50, 2)
  png_*, pfm_* = the bytes of the files WritePng / PfmSave write from filtered_*
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.oracle_api import RefOracle  # noqa: E402


def synthetic():
    rng = np.random.default_rng(11)
    H, W = 40, 56
    w = (rng.random((H, W)) * 30 + 0.5).astype(np.float32)
    rgb = (rng.random((H, W, 3)) ** 3 * 4.0).astype(np.float32) * w[..., None]
    a = np.concatenate([rgb, w[..., None]], axis=-1).astype(np.float32)
    a[0, 0] = [0, 0, 0, 0]                      # 0/0
    a[0, 1] = [1, 2, 3, 0]                      # x/0
    a[0, 2] = [-1, 0.5, 1e30, 1]                # negative, huge
    a[0, 3] = [1e-40, 1e-38, 1e-30, 1]          # denormal / tiny
    a[0, 4] = [0.004, np.nextafter(np.float32(0.004), np.float32(1)), np.nextafter(np.float32(0.004), np.float32(0)), 1]
    a[0, 5] = [3e38, 3e38, 3e38, 2]
    a[1, :8, :3] = np.linspace(0.0, 0.02, 24, dtype=np.float32).reshape(8, 3)
    a[1, :8, 3] = 1.0
    return a


def main():
    R = RefOracle()
    out = {}
    g = np.load(os.path.join(HERE, "cornell.golden.npz"))
    for tag, accum, exposure in (("a", g["accum"], 1.0), ("b", synthetic(), 0.37)):
        filt = R.present(accum, exposure, 1.5)
        d = tempfile.mkdtemp()
        R.write_png(filt, os.path.join(d, "x.png"))
        R.pfm_save(filt[..., :3], os.path.join(d, "x.pfm"))
        out["accum_" + tag] = accum
        out["exposure_" + tag] = np.float32(exposure)
        out["filtered_" + tag] = filt
        with np.errstate(all="ignore"):
            clean = np.nan_to_num(filt, nan=0.25, posinf=1.0, neginf=0.0)     # NLM input must be finite to be comparable
        out["nlm_in_" + tag] = clean
        out["nlm_r1_" + tag] = R.nlm(clean, 200.0, 1)
        out["nlm_r2_" + tag] = R.nlm(clean, 50.0, 2)
        out["png_" + tag] = np.frombuffer(open(os.path.join(d, "x.png"), "rb").read(), np.uint8)
        out["pfm_" + tag] = np.frombuffer(open(os.path.join(d, "x.pfm"), "rb").read(), np.uint8)
        print(tag, accum.shape, "filtered mean", np.nanmean(filt[..., :3]), "png bytes", out["png_" + tag].size,
              "non-finite", int((~np.isfinite(filt)).sum()))
    np.savez_compressed(os.path.join(HERE, "display.golden.npz"), **out)


if __name__ == "__main__":
    main()
