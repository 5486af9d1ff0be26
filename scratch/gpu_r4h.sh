#!/bin/bash
# round 4, call H: full-frame read-backs into pageable arrays through the library's own staged copy (StagedCopy): tests that read frames
# back, then the API call pattern's rates by thread count and chunk size
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4h; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_group.py tests/test_gpu_shim.py tests/test_gpu_display.py tests/test_gpu_multirank.py -x -q 2>&1 | tail -6 ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
python - <<'PY' 2>&1 | tee $O/ab_staged_copy.md
import os, subprocess, sys, json
ROOT=os.environ["GRAFT_REPO_ROOT"]
code = r'''
import os, sys, time, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import tinsel_amd
from tinsel_amd import abi
W = H = int(sys.argv[1])
scene = tinsel_amd.Scene.load_pack(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests/golden/cornell.pack"))
cam, opt = scene.camera, scene.options.copy(); opt.width, opt.height, opt.mode = W, H, abi.MODE_PATHTRACE
r = tinsel_amd.create_gpu_renderer(scene); r.init(W, H); r.reserve(16, opt.max_depth)
out = np.empty((H, W, 4), np.float32)
r.render(cam, opt, output=out, passes=1)
res = {}
def calls(n):
    r.render(cam, opt, output=out, passes=1)
    t0 = time.perf_counter()
    for _ in range(n): r.render(cam, opt, output=out, passes=1)
    return n*W*H/(time.perf_counter() - t0)/1e6
res["plain"] = calls(64)
r.set_lookahead(abi.LOOKAHEAD_ON); res["lookahead"] = calls(128)
r.set_lookahead(abi.LOOKAHEAD_PIN_OUTPUT); res["pinned"] = calls(128)
r.set_lookahead(abi.LOOKAHEAD_OFF)
t0 = time.perf_counter(); r.render(cam, opt, output=out, passes=16); res["16pass"] = 16*W*H/(time.perf_counter()-t0)/1e6
r.close()
print(" | ".join("%s %.0f" % kv for kv in res.items()))
'''
print("| frame | TINSEL_HIP_COPY_THREADS / CHUNK_KB | Msamples/s: 1 pass + read-back per call, plain / look-ahead (pageable) / look-ahead (page-locked) / 16 passes per read-back |")
print("|---|---|---|")
for W in (1024, 512, 2048):
    for env in ({"TINSEL_HIP_COPY_THREADS": "0"}, {}, {"TINSEL_HIP_COPY_THREADS": "4"}, {"TINSEL_HIP_COPY_THREADS": "16"}, {"TINSEL_HIP_COPY_THREADS": "32"},
                {"TINSEL_HIP_COPY_CHUNK_KB": "256"}, {"TINSEL_HIP_COPY_CHUNK_KB": "4096"}, {"TINSEL_HIP_COPY_THREADS": "16", "TINSEL_HIP_COPY_CHUNK_KB": "512"}):
        p = subprocess.run([sys.executable, "-c", code, str(W)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        print("| %d^2 | %s | %s |" % (W, ", ".join("%s=%s" % (k.replace("TINSEL_HIP_COPY_", ""), v) for k, v in env.items()) or "default (8 threads, 1024 KB)", (p.stdout.strip().splitlines() or [p.stderr[-300:]])[-1]), flush=True)
PY
