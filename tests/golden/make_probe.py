#!/usr/bin/env python3
"""Builds the real-probe fixture: tests/golden/scenes/env_loft.tin.in lit by the reference's
data/probes/loft.hdr (1600x800 lat-long HDR; the probe every shipped scene names, vankleef.hdr, is not in the
reference tree).  The reference's own loader builds the probe CDF tables (probe.h:31-79).

  tests/golden/large/env_loft.pack        scene pack incl. the probe texels + pdf/cdf tables (~31 MB, git-ignored,
                                          travels to the GPU box like the built .so files)
  tests/golden/env_loft.golden.npz        the reference's PathTrace under the per-path seed contract at 96x48, 4 passes
                                          (committed; same layout as the other fixtures)
Needs /root/reference."""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.oracle_api import RefOracle  # noqa: E402
from tests.golden.make_golden import make_outputs  # noqa: E402


def main(ref="/root/reference"):
    R = RefOracle()
    d = tempfile.mkdtemp()
    probe = os.path.relpath(os.path.join(ref, "data", "probes", "loft.hdr"), d)
    text = open(os.path.join(HERE, "scenes", "env_loft.tin.in")).read().replace("@PROBE@", probe)
    tin = os.path.join(d, "env_loft.tin")
    open(tin, "w").write(text)
    h = R.load_tin(tin)
    out = os.path.join(HERE, "large", "env_loft.pack")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    n = R.write_pack(h, out)
    print("wrote %s (%.1f MB)" % (out, n/1e6))
    make_outputs(R, h, "env_loft", 96, 48, 4, None)
    R.free(h)


if __name__ == "__main__":
    main(*sys.argv[1:2])
