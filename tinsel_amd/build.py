"""Builds tinsel_amd/libtinsel_hip.so (the C-ABI of include/tinsel_hip.h) with hipcc for gfx950.

The library is built IN-TREE so that it travels with the repository snapshot to the GPU box.  Two translation units,
two floating-point contracts (tn_launch.h):
  csrc/tinsel_hip.hip    host side + every kernel under the PARITY contract (DESIGN.md "Arithmetic"):
      -ffp-contract=off   no FMA contraction: every fp32 op rounds once, like the CPU oracle's
      (no -ffast-math)    IEEE division / sqrt; sinf/cosf/expf/acosf/atan2f restate glibc 2.35's algorithms
  csrc/tinsel_fast.hip   the path kernels again under the opt-in TOLERANCE contract (tinsel_hip_set_arithmetic):
      -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -freciprocal-math -fgpu-flush-denormals-to-zero
      (never -ffinite-math-only: the traversal relies on 1/0 = inf like the reference)
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "tinsel_hip.hip")
SRC_FAST = os.path.join(HERE, "csrc", "tinsel_fast.hip")
OUT = os.path.join(HERE, "libtinsel_hip.so")
OBJ_DIR = os.path.join(HERE, "csrc", "_obj")
DEPS = [os.path.join(HERE, "csrc", f) for f in sorted(os.listdir(os.path.join(HERE, "csrc"))) if not f.startswith("_")] + \
       [os.path.join(ROOT, "include", "tinsel_hip.h")]

# -fno-slp-vectorize: the SLP vectoriser pairs neighbouring fp32 multiplies / adds into v_pk_mul_f32 / v_pk_add_f32.  On gfx950 a
# packed fp32 op takes as long as the two plain ones (scratch/ubench/valu_bench.hip: 4.9 vs 2 x 2.65 cycles) but wants its
# operands in aligned register PAIRS: k_bounce carried 1,000 extra v_mov and 100-192 B of scratch for it, k_extend / k_shadow
# 128 VGPRs instead of 98 / 77, k_walk 83 instead of 65.  Same arithmetic (a packed op rounds each half like the plain one),
# measured: cornell 2894 -> 3056 Msamples/s, glass 1191 -> 1303, many_spheres 1430 -> 1648, the 524k-triangle config 2028 -> 2114.
COMMON_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize",
    "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
]
HIPCC_FLAGS = COMMON_FLAGS + ["-ffp-contract=off", "-fno-fast-math"]
FAST_FLAGS = COMMON_FLAGS + ["-DTN_FAST=1", "-ffp-contract=fast", "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-freciprocal-math",
                             "-fgpu-flush-denormals-to-zero"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built (there is no CPU fallback)")


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(d) <= t for d in DEPS)


def _compile(obj_dir, extra=(), verbose=True):
    """Compiles the two translation units side by side into obj_dir (always: no up-to-date shortcut)."""
    os.makedirs(obj_dir, exist_ok=True)
    cc = hipcc()
    # (run INSIDE obj_dir with a relative output name: the object embeds the output path as given, and two compiles of the same source
    # with the same command line are byte-identical only then -- what build_verified compares)
    jobs = [
        [cc] + HIPCC_FLAGS + list(extra) + ["-c", SRC, "-o", "tinsel_hip.o"],
        [cc] + FAST_FLAGS + list(extra) + ["-c", SRC_FAST, "-o", "tinsel_fast.o"],
    ]
    procs = []
    for cmd in jobs:
        if verbose:
            print("[tinsel_amd.build] (in %s)" % obj_dir, " ".join(cmd), flush=True)
        procs.append(subprocess.Popen(cmd, cwd=obj_dir))
    for p, cmd in zip(procs, jobs):
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)


def _resource_job(tmp):
    """The parity translation unit's device code once more, with -Rpass-analysis=kernel-resource-usage: registers, scratch and occupancy
    of every kernel as the compiler reports them.  (A compile of its own: the flag changes the object's bytes, and the shipped object
    must be the one build() makes.)  Returns the process; parse its stderr with _parse_resources."""
    cmd = [hipcc()] + HIPCC_FLAGS + ["--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", SRC, "-o", "resources_only.o"]
    return subprocess.Popen(cmd, cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)


def _kernel_name(mangled):
    """_ZN2tn8k_bounceILb0ELb1ELb0ELi4EEEvNS_8DevSceneE... -> k_bounce<0,1,0,4>   (template arguments of the kernels here are bools and ints).
    The name is read by its Itanium LENGTH PREFIX (a lazy match up to the first v / E / P / N cut k_lbvh_leaves down to k_lb: ADVICE r04)."""
    import re
    m = re.match(r"_ZN2tn(\d+)", mangled)
    if not m:
        return mangled
    n = int(m.group(1))
    name, rest = mangled[m.end():m.end() + n], mangled[m.end() + n:]
    if len(name) != n or not name.startswith("k_"):
        return mangled
    t = re.match(r"I((?:L[bij]n?\d+E)+)E", rest)
    args = re.findall(r"L[bij](n?\d+)E", t.group(1)) if t else []
    return name + ("<" + ",".join(a.replace("n", "-") for a in args) + ">" if args else "")


def _parse_resources(text):
    import re
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass-analysis", line)
        if not m:
            continue
        body = m.group(1)
        if body.startswith("Function Name:"):
            cur = out.setdefault(_kernel_name(body.split(":", 1)[1].strip()), {})
        elif cur is not None and ":" in body:
            k, v = body.split(":", 1)
            key = {"VGPRs": "vgprs", "AGPRs": "agprs", "TotalSGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch_bytes",
                   "Occupancy [waves/SIMD]": "waves_per_simd", "LDS Size [bytes/block]": "static_lds_bytes",
                   "SGPRs Spill": "sgpr_spills", "VGPRs Spill": "vgpr_spills"}.get(k.strip())
            if key:
                cur[key] = int(v.strip())
    return out


def _link(obj_dir, out, verbose=True):
    link = [hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out, os.path.join(obj_dir, "tinsel_hip.o"), os.path.join(obj_dir, "tinsel_fast.o")]
    if verbose:
        print("[tinsel_amd.build]", " ".join(link), flush=True)
    subprocess.run(link, check=True)


def build(force=False, verbose=True, extra=()):
    if not force and up_to_date():
        return OUT
    _compile(OBJ_DIR, extra, verbose)
    _link(OBJ_DIR, OUT, verbose)
    return OUT


def _sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def build_verified(verbose=True):
    """What __graft_entry__.build() runs: ALWAYS compiles both translation units for gfx950 (into a temporary directory, with the
    shipped flags), so that a prebuilt library in the tree cannot pass for a build.  hipcc is deterministic for a given source and
    command line: when the fresh objects are byte-identical to the ones the in-tree library was linked from, the library is kept
    ("verified"); otherwise they replace them and the library is linked again ("rebuilt").  Returns (and prints) the build record."""
    t0 = time.time()
    names = ("tinsel_hip.o", "tinsel_fast.o")
    with tempfile.TemporaryDirectory(prefix="tinsel_build_") as tmp:
        res_proc = _resource_job(tmp)               # (beside the two compiles below: the box has the cores)
        _compile(tmp, (), verbose)
        res_err = res_proc.communicate()[1]
        resources = _parse_resources(res_err) if res_proc.returncode == 0 else {}
        fresh = {n: _sha(os.path.join(tmp, n)) for n in names}
        have = {n: (_sha(os.path.join(OBJ_DIR, n)) if os.path.exists(os.path.join(OBJ_DIR, n)) else None) for n in names}
        same = all(fresh[n] == have[n] for n in names) and os.path.exists(OUT) and \
            all(os.path.getmtime(OUT) >= os.path.getmtime(os.path.join(OBJ_DIR, n)) for n in names)
        if not same:
            os.makedirs(OBJ_DIR, exist_ok=True)
            for n in names:
                shutil.copy2(os.path.join(tmp, n), os.path.join(OBJ_DIR, n))
            _link(OBJ_DIR, OUT, verbose)
    ver = subprocess.run([hipcc(), "--version"], capture_output=True, text=True).stdout.splitlines()
    record = {
        "compiled": [os.path.relpath(SRC, ROOT), os.path.relpath(SRC_FAST, ROOT)], "arch": "gfx950", "seconds": round(time.time() - t0, 1),
        "object_sha256": fresh, "library": os.path.relpath(OUT, ROOT), "library_sha256": _sha(OUT),
        "action": "verified: the in-tree library was linked from byte-identical objects" if same else "rebuilt: objects replaced, library linked again",
        "hipcc": ver[0] if ver else None,
    }
    # per-kernel registers / scratch / occupancy of THIS build (tests/test_resources.py holds the hot kernels to their budgets)
    with open(os.path.join(OBJ_DIR, "resources.json"), "w") as f:
        json.dump({"object_sha256": fresh["tinsel_hip.o"], "kernels": resources}, f, indent=1, sort_keys=True)
    record["kernels_reported"] = len(resources)
    with open(os.path.join(OBJ_DIR, "build_record.json"), "w") as f:
        json.dump(record, f, indent=1)
    print("[tinsel_amd.build] record:", json.dumps(record), flush=True)
    return record


if __name__ == "__main__":
    build(force="--force" in sys.argv, extra=[a for a in sys.argv[1:] if a.startswith("-") and a != "--force"])
