"""Per-launch durations of ONE batch of the split pipeline, in launch order (run under rocprofv3 --kernel-trace --output-format csv,
then: launch_timeline.py --parse <kernel_trace.csv>).  usage: launch_timeline.py <pack> <W> <H> <depth> <passes>"""
import os, sys, csv, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if sys.argv[1] == "--parse":
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [re.search(r"(k_\w+)", r["Kernel_Name"]) for r in rows]
    rows = [(m.group(1), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))/1e3) for m, r in zip(names, rows) if m]
    # the last batch: from the last k_generate on
    last = max(i for i, (n, _) in enumerate(rows) if n == "k_generate")
    bounce = -1
    line = []
    for n, us in rows[last:]:
        if n in ("k_extend",) and line and any(x.startswith("k_extend") for x in line):
            print("  ".join(line)); line = []
        line.append("%s %.0f" % (n, us))
    print("  ".join(line))
    sys.exit(0)
import tinsel_amd
from tinsel_amd import abi
pack, W, H, depth, passes = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
scene = tinsel_amd.Scene.load_pack(os.path.join(ROOT, "tests/golden", pack + ".pack"))
cam, opt = scene.camera, scene.options.copy(); opt.width, opt.height, opt.max_depth, opt.mode = W, H, depth, abi.MODE_PATHTRACE
r = tinsel_amd.create_gpu_renderer(scene); r.init(W, H); r.reserve(passes, depth)
for _ in range(2):
    r.render(cam, opt, passes=passes, readback=False)
print("K =", r.nee_per_path, "queue counts (extension, shadow):", r.queue_counts() if hasattr(r._L, "tinsel_hip_queue_counts") else "n/a")
r.close()
