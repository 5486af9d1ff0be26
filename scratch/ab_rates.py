#!/usr/bin/env python3
"""A/B of library builds by fresh processes: for every (build, workload) pair, `--repeat` runs of bench.py in processes of their own
(TINSEL_HIP_LIB picks the build), rates and the per-kernel milliseconds of the timed block as a markdown table with min / median / max --
the spread over PROCESSES is what the split pipeline's streaming kernels were quoted as ranges for.

  python scratch/ab_rates.py --lib name=path [--lib name=path ...] [--repeat 5] [--steps 20] workload [workload ...]
  (--lib name=tuning:{"walk_refill_min": 16}  runs the in-tree library under that tinsel_hip_tuning instead of another build;
   --lib 'name=args:--pipeline paired'        the in-tree library with extra bench.py arguments)
  workloads: cornell, veach4k, glass, cfg3, aphrodite, many_spheres, motionblur, cfg1, table, transmission, meshlight
"""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORK = {
    "cornell": ["--scene", "cornell", "--width", "1024", "--height", "1024"],
    "cfg1": ["--scene", "cornell", "--width", "256", "--height", "256"],
    "veach4k": ["--scene", "veach", "--width", "3840", "--height", "2160"],
    "glass": ["--scene", "glass", "--width", "1920", "--height", "1080", "--maxdepth", "12"],
    "cfg3": ["--scene", "large/ajax_standin", "--width", "1920", "--height", "1080", "--maxdepth", "4"],
    "aphrodite": ["--scene", "large/ajax_aphrodite", "--width", "1920", "--height", "1080", "--maxdepth", "4"],
    "many_spheres": ["--scene", "many_spheres", "--width", "1024", "--height", "768"],
    "motionblur": ["--scene", "motionblur", "--width", "1920", "--height", "1080"],
    "table": ["--scene", "large/table", "--width", "1920", "--height", "1080"],
    "transmission": ["--scene", "large/transmission", "--width", "1920", "--height", "1080"],
    "meshlight": ["--scene", "large/meshlight", "--width", "1920", "--height", "1080"],
}


def one(lib, work, steps):
    """lib: a library path, or `tuning:{json}` for the in-tree library under a tinsel_hip_tuning"""
    tuning, extra, path = None, [], ""
    for part in lib.split(";"):         # `path`, `tuning:{json}`, `args:...`, or several joined by ';'
        if part.startswith("tuning:"):
            tuning = part[len("tuning:"):]
        elif part.startswith("args:"):
            extra = part[len("args:"):].split()
        else:
            path = part
    lib = path
    env = dict(os.environ, TINSEL_HIP_LIB=lib) if lib else dict(os.environ)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "5", "--no-pmc", "--no-fast", "--no-api", "--no-ubench",
           "--no-cpu-baseline", "--no-second-config", "--no-more-configs"] + WORK[work] + (["--tuning", tuning] if tuning else []) + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    if p.returncode != 0:
        return None, {}
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    det = [l for l in p.stderr.splitlines() if l.startswith("bench_detail: ")]
    kms = json.loads(det[-1][len("bench_detail: "):])["roofline"]["kernel_ms"] if det else {}
    return line["value"], kms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", action="append", default=[])
    ap.add_argument("--repeat", type=int, default=5)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("work", nargs="+")
    a = ap.parse_args()
    libs = [x.split("=", 1) for x in a.lib] or [["default", ""]]
    print("| workload | build | Msamples/s min / median / max | spread | kernels, ms per %d passes (median; min-max where > 2 %%) |" % a.steps)
    print("|---|---|---|---|---|")
    for w in a.work:
        # interleaved: build A, build B, build A, ... so that a drift of the box hits both alike
        runs = {name: [] for name, _ in libs}
        for _ in range(a.repeat):
            for name, path in libs:
                v, k = one(path, w, a.steps)
                if v is not None:
                    runs[name].append((v, k))
        for name, _ in libs:
            rs = runs[name]
            if not rs:
                print("| %s | %s | failed | | |" % (w, name))
                continue
            vals = sorted(v for v, _ in rs)
            med = statistics.median(vals)
            kernels = sorted({k for _, ks in rs for k in ks})
            cells = []
            for k in kernels:
                xs = sorted(ks.get(k, 0.0) for _, ks in rs)
                m = statistics.median(xs)
                cells.append("%s %.2f" % (k, m) + (" (%.2f-%.2f)" % (xs[0], xs[-1]) if m > 0 and (xs[-1] - xs[0])/m > 0.02 else ""))
            print("| %s | %s | %.0f / %.0f / %.0f | +-%.1f %% | %s |" % (w, name, vals[0], med, vals[-1], 50.0*(vals[-1] - vals[0])/med, ", ".join(cells)), flush=True)


if __name__ == "__main__":
    main()
