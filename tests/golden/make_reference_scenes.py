#!/usr/bin/env python3
"""The reference's OTHER shipped scenes as parity fixtures: data/table.tin, transmission.tin, meshlight.tin, example.tin, env.tin -- every
scene of /root/reference/data whose meshes are in the tree (ajax.tin, ajaxenv.tin and sportscar.tin reference meshes that are not), beyond
the ones tests/golden/make_golden.py already covers.  Several large meshes per scene (table: seven mesh primitives of up to 30,240
triangles; transmission: seven walked meshes, maxDepth 16; meshlight: a 36,752-triangle mesh AS the light).  example, env and
transmission are lit by probes/vankleef.hdr, which is not in the reference tree (they would render black): they get the harness's
procedural 64x32 probe, built into CDF tables by the reference's own Probe::BuildCDF (the real loft.hdr is tests/golden/make_probe.py's).

  tests/golden/large/<name>.pack     written by the reference's own loader + Scene::Build (git-ignored: 1-35 MB; travels with the tree)
  tests/golden/<name>.golden.npz     the reference's PathTrace + AddSample at a small size, under the per-path seed contract (committed)
Needs /root/reference."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from tests.oracle_api import RefOracle  # noqa: E402
from make_golden import make_outputs  # noqa: E402

# name -> (W, H, passes, maxDepth override or None, procedural probe)
SCENES = {
    "table": (80, 50, 3, None, False),
    "transmission": (100, 25, 3, None, True),
    "meshlight": (64, 64, 3, None, False),
    "example": (100, 40, 3, None, True),
    "env": (64, 64, 3, None, True),
}


def main(ref="/root/reference"):
    R = RefOracle()
    os.makedirs(os.path.join(HERE, "large"), exist_ok=True)
    for name, (W, H, passes, depth, probe) in SCENES.items():
        h = R.load_tin(os.path.join(ref, "data", name + ".tin"))
        if probe:
            R.lib.ref_scene_set_procedural_probe(h, 64, 32)
        n = R.write_pack(h, os.path.join(HERE, "large", name + ".pack"))
        print("%s.pack %.1f MB, %d primitives" % (name, n/1e6, R.num_primitives(h)))
        # make_outputs writes tests/golden/<name>.golden.npz
        make_outputs(R, h, name, W, H, passes, depth)
        R.free(h)


if __name__ == "__main__":
    main(*sys.argv[1:2])
