#!/usr/bin/env python3
"""bench.py -- Msamples/s and Mrays/s of the hot path on N MI355X (one process per GPU).

A "step" is one pass of the hot path over one batch of synthetic input: one sample per pixel.  The default run (N = 1)
times the FOUR GPU configurations of BASELINE.json and prints ONE SMALL JSON line (<= 6 KB: the contract's fields for the headline,
`configs`: one compact object per other configuration); the full records -- per-kernel tables, counter calibration, yard-sticks, API
legs, per-rank statistics -- go to bench_detail.json beside this file (and to stderr behind "bench_detail: ").

  headline     BASELINE configs[1]: data/cornell.tin 1024x1024 maxDepth 4 (spp 256 <=> --steps 256).  The scene (2.6 KB) lives in
               LDS: the path is bound by VALU issue, and `roofline` says so (bound "valu": wave-instructions issued per second
               against SIMDs x clock / 2).
  configs[0]   BASELINE configs[2]: data/ajax.tin with the 524,288-triangle stand-in mesh (ajax.obj is not in the reference tree)
               at 1920x1080 maxDepth 4 -- the configuration whose scene lives in HBM / the Infinity Cache (HBM fractions from
               counter bytes; the node-visit rate of k_walk next to this GPU's record-chase ceilings in the detail file).
  configs[1]   the same configuration with a REAL scanned mesh: the reference's Aphrodite_from_jotero_com.obj, 1 -> 4 subdivided to 427,384
               triangles by the reference's own importer and builder (tests/golden/make_large.py aphrodite) -- an irregular tree.
  configs[2]   BASELINE configs[3]: data/glass.tin 1920x1080 maxDepth 12.
  configs[3]   BASELINE configs[4]: data/veach.tin 3840x2160 (at N = 1 the whole frame; at N > 1 every rank's pixel tiles of it).

Scenes come from scene packs written by the reference's own loader (tests/golden/*.pack); camera rays, RNG seeds and
everything downstream are generated on the GPU, so inputs are resident in HBM when a timed region starts, and the
accumulation buffer stays in HBM (`api_msamples_s` reports the API's per-call D2H patterns separately).

Timing: W untimed warm-up steps, then blocks of EXACTLY K steps, each bracketed by barrier + torch.cuda.synchronize()
on both sides, repeated until >= 0.5 s have been timed; `ms_per_step` is the MEDIAN block / K, `value` the samples of
one block / that median (max over ranks for N > 1).

Counters: `roofline.traffic` (HBM bytes per launch of the dominant kernel) and the VALU instruction count behind
`roofline.achieved` are measured IN THIS RUN, on this box, by re-running the same workload under
`rocprofv3 --pmc` (separate passes for FETCH_SIZE, WRITE_SIZE, the SQ counters and TCC hits / misses).  What one count of
FETCH_SIZE / WRITE_SIZE stands for is CALIBRATED in the same passes on kernels with a known byte count
(tinsel_hip_ubench: a float4 stream copy for the streaming kernels -- the guide's gfx950 x2 -- and dependent 64-B record
chases through a 1 GiB table, beyond the Infinity Cache, for the walking kernels); the factors are in the detail file
(`counter_calibration`).  When rocprofv3 is not available the fields are null -- nothing is read from a stored file.

N > 1 (launched by torch.distributed.run; the ranks meet at once, under a 90 s watchdog): `value` is WEAK scaling -- every rank
traces its interleaved pixel tiles for K*N passes (same paths per GPU as N = 1), then ONE RCCL sum-reduce of the float4 accumulator
to rank 0 inside the timed region (tinsel_amd.distributed.reduce_accum).  `strong_msamples_s` is the FIXED-WORK leg beside it: the
same K full-frame passes as N = 1 split over the ranks' tiles + the reduce (north_star's 8-GPU configuration, veach 4K at 4096
spp, is a fixed job); veach 4K gets both legs too (`configs`).  AFTER the ranks' timed regions rank 0 times the SAME N devices
through the library's own multi-GPU path (`--group`: tinsel_hip_group, what the reference's single-process C++ caller gets through
the shim) in child processes limited to 60 s each: `group`, `group_cfg5`.
"""
import argparse
import csv
import ctypes as C
import glob
import json
import os
import re
import shutil
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
SIMDS = 256*4                   # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9                # max clock, same guide
VALU_PEAK = SIMDS*CLOCK_HZ/2    # a wave64 VALU instruction issues over 2 cycles on a SIMD-32 (same guide; scratch/ubench/valu_bench.hip: 2.6)
MIN_TIMED_S = 0.5
GROUP_LEG_LIMIT_S = 60            # each tinsel_hip_group leg of an N-rank run (child process of rank 0, after the ranks' timed regions)
LARGE = "large/ajax_standin"
REAL_MESH = "large/ajax_aphrodite"   # config 3 once more: ajax.tin with the reference's own Aphrodite scan, 1 -> 4 subdivided (427,384 triangles, an irregular tree)
YARD = [None]                   # this run's yard-sticks (yard_sticks(): stream copy, record chases), N = 1 only


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--scene", default="cornell")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--maxdepth", type=int, default=0, help="0 = the scene's own / BASELINE value")
    ap.add_argument("--pipeline", choices=["auto", "wavefront", "mega", "split", "paired"], default="auto")
    ap.add_argument("--bvh", choices=["reference", "lbvh", "ploc"], default="reference",
                    help="mesh BVHs: the reference's host-built trees (parity path) or rebuilt on the device")
    ap.add_argument("--roulette", type=int, default=0, help="opt-in Russian roulette from this bounce on (0 = the reference's behaviour)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU-core-seconds of oracle work")
    ap.add_argument("--tile", type=int, default=64, help="pixel-tile edge of the multi-GPU shard")
    ap.add_argument("--no-second-config", action="store_true", help="skip BASELINE config 3 (the 524k-triangle mesh)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc passes (traffic / VALU counts become null)")
    ap.add_argument("--no-fast", action="store_true", help="skip the opt-in tolerance-arithmetic leg (fast_msamples_s / fast_l2)")
    ap.add_argument("--no-api", action="store_true", help="skip the API call-pattern legs (pcie_inclusive / api_1pass), e.g. under rocprofv3 --stats")
    ap.add_argument("--arith", choices=["exact", "fast"], default="exact", help="arithmetic arm of the TIMED run (default: the bit-exact parity path)")
    ap.add_argument("--no-more-configs", action="store_true", help="skip BASELINE configs 4 and 5 (glass depth 12, veach 4K)")
    ap.add_argument("--no-ubench", action="store_true", help="skip the stream-copy / record-chase yard-sticks (and the counter calibration)")
    ap.add_argument("--group", action="store_true", help="time the library's own multi-GPU path (tinsel_hip_group over --gpus devices, one process) instead")
    ap.add_argument("--no-group-leg", action="store_true", help="N > 1: do not time the tinsel_hip_group path from rank 0 before the ranks meet")
    ap.add_argument("--tuning", default="", help='A/B: a tinsel_hip_tuning as JSON, e.g. \'{"walk_refill_min": 16}\' (tinsel_amd.abi.Tuning; the default run sets nothing)')
    ap.add_argument("--force-comm", action="store_true", help="N = 1 validation: make the library's RCCL communicator of ONE rank and put its ncclReduce inside the timed region")
    ap.add_argument("--inner-pmc", action="store_true", help=argparse.SUPPRESS)      # the child process the PMC passes profile
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (the reference's own code; checker infrastructure, never the product path)

def cpu_baseline(scene_name, cam, opt, target_core_seconds):
    """The reference's own PathTrace (oracle/_ref, compiled unmodified with the reference's -O3 -ffast-math flags) on
    this box's host cores, on a bounded sample of the same workload."""
    from tests.oracle_api import GOLDEN, REF_FAST_SO, REF_SO, RefOracle
    if not os.path.exists(REF_SO):
        return None
    fast = os.path.exists(REF_FAST_SO)
    R = RefOracle(fast=fast)
    h = R.load_pack(os.path.join(GOLDEN, scene_name + ".pack"))
    cores = os.cpu_count() or 1
    # 1-core faithful loop (CpuRenderer::Render exactly as main.cpp:246-250 drives it) on a 256x256 frame
    small = opt.copy()
    small.width, small.height = 256, 256
    _, t1 = R.render_faithful(h, cam, small, 2)
    one_core = 2*256*256/t1
    # all cores: per-path-seeded oracle, full frame, as many passes as ~target_core_seconds of work
    passes = max(1, int(round(target_core_seconds*one_core/(opt.width*opt.height))))
    t0 = time.perf_counter()
    _, _, trace_s = R.render_seeded(h, cam, opt, 0, passes, threads=cores, want_accum=False)
    wall = time.perf_counter() - t0
    R.free(h)
    samples = passes*opt.width*opt.height
    return {
        "value": samples/trace_s/1e6, "unit": "Msamples/s", "cores": cores, "kind": "reference",
        "sample": "%s %dx%d maxDepth=%d, %d of the passes (%.1f s wall); reference render.cpp PathTrace, g++ %s" % (
            scene_name, opt.width, opt.height, opt.max_depth, passes, wall, "-O3 -ffast-math (reference makefile:4)" if fast else "-O2"),
        "one_core_faithful_msamples_s": one_core/1e6,
    }


# ---------------------------------------------------------------------------------------------------------------------
# live PMC passes: this same script (--inner-pmc) under rocprofv3, counters summed per kernel

PMC_SETS = {
    "sq": ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_BUSY_CYCLES", "SQ_WAVES", "GRBM_GUI_ACTIVE"],
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE"],
    "tcc": ["TCC_HIT_sum", "TCC_MISS_sum"],
}
# kernels whose memory traffic is random record gathers (BVH walks); every other kernel of the path streams
GATHER_KERNELS = ("k_walk", "k_extend", "k_shadow")
# kernels that run only when the whole scene is LDS-resident: their bytes never leave the CU, what binds them is instruction issue
LDS_SCENE_KERNELS = ("k_bounce", "k_mega")
UB_COPY_BYTES = 256 << 20           # per buffer
UB_BIG_TABLE = 1 << 30              # beyond the 256 MiB Infinity Cache
UB_TREE_TABLE = 32 << 20            # the 524,288-triangle tree: 33.5 MB of Node64
UB_L2_TABLE = 2 << 20               # inside one XCD's 4 MiB L2
UB_STEPS = 64


def _kernel_key(name):
    """rocprofv3's kernel name -> the name the library's own timers file the launch under (tinsel_hip_kernel_times)"""
    m = re.search(r"(k_\w+)(<[^>]*>)?", name)
    if not m:
        return None
    k, targs = m.group(1), m.group(2) or ""
    if k in ("k_extend", "k_shadow", "k_bounce", "k_mega") and targs.startswith("<true"):
        return None                         # detail-counting variants (COUNT = true, their first template argument) are not the product kernels
    if k == "k_ub_gather":
        return "k_ub_gather" + targs
    if k.startswith("k_accumulate"):
        return "k_accumulate"
    if k in ("k_seg_prefix", "k_seg_expand", "k_seg_expand_all", "k_region_order"):
        return "k_seg"
    if k == "k_walk_rays":                  # several walked primitives: the same timer
        return "k_walk"
    if k == "k_swalk":                      # the scene-level walk stands in for k_extend / k_shadow (its first template argument: shadow rays)
        return "k_shadow" if targs.startswith("<true") else "k_extend"
    if k == "k_shade_sorted":
        return "k_shade"
    return k


def pmc_pass(args, scene, width, height, maxdepth, steps, counters, timeout=150):
    """Runs `steps` passes of the workload under rocprofv3 --pmc <counters>; returns {kernel: {counter: sum, 'launches': n}}."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    tmp = tempfile.mkdtemp(prefix="tinsel_pmc_")
    try:
        cmd = [exe, "--kernel-trace", "--pmc"] + counters + ["-d", tmp, "-o", "p", "--output-format", "csv", "--",
               sys.executable, os.path.abspath(__file__), "--inner-pmc", "--scene", scene, "--width", str(width), "--height", str(height),
               "--maxdepth", str(maxdepth), "--steps", str(steps), "--pipeline", args.pipeline, "--bvh", args.bvh, "--roulette", str(args.roulette)]
        if args.no_ubench:
            cmd.append("--no-ubench")
        if args.tuning:
            cmd += ["--tuning", args.tuning]
        env = dict(os.environ, TMPDIR="/tmp")
        p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
        files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
        if p.returncode != 0 or not files:
            return None
        out = {}
        for f in files:
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = _kernel_key(row["Kernel_Name"])
                    if k is None:
                        continue
                    d = out.setdefault(k, {"_ids": set()})
                    d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                    d["_ids"].add(row["Dispatch_Id"])
        for d in out.values():
            d["launches"] = len(d.pop("_ids"))
        return out
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def inner_pmc(args):
    """Child of pmc_pass: the same renderer set-up as the timed run, `steps` passes, no output."""
    import tinsel_amd
    from tinsel_amd import abi
    scene = tinsel_amd.Scene.load_pack(os.path.join(ROOT, "tests", "golden", args.scene + ".pack"))
    cam, opt = scene.camera, scene.options.copy()
    opt.width, opt.height, opt.mode = args.width, args.height, abi.MODE_PATHTRACE
    if args.maxdepth > 0:
        opt.max_depth = args.maxdepth
    r = tinsel_amd.create_gpu_renderer(scene, 0)
    if args.bvh != "reference":
        r.set_mesh_bvh(abi.BVH_LBVH if args.bvh == "lbvh" else abi.BVH_PLOC)
    if args.roulette > 0:
        r.set_russian_roulette(args.roulette)
    r.set_pipeline({"auto": abi.PIPELINE_AUTO, "wavefront": abi.PIPELINE_WAVEFRONT, "mega": abi.PIPELINE_MEGAKERNEL, "split": abi.PIPELINE_WAVEFRONT_SPLIT, "paired": abi.PIPELINE_WAVEFRONT_PAIRED}[args.pipeline])
    r.init(opt.width, opt.height)
    r.reserve(args.steps, opt.max_depth)
    r.render(cam, opt, passes=args.steps, readback=False)
    r.close()
    if not args.no_ubench:
        # the counters' calibration kernels, in the same profiled process: known byte counts
        tinsel_amd.ubench(tinsel_amd.renderer.UBENCH_COPY, UB_COPY_BYTES)
        tinsel_amd.ubench(tinsel_amd.renderer.UBENCH_GATHER_BEYOND_CACHE, UB_BIG_TABLE, UB_STEPS)


def calibrate(pmc):
    """Bytes one count of FETCH_SIZE / WRITE_SIZE (KB units) stands for, from the yard-stick kernels of the same passes.
    every k_ub_copy launch (a few shapes, warm-up + timed each) reads and writes UB_COPY_BYTES; k_ub_gather<0> visits
    grid*256*UB_STEPS records of 64 B per launch, all of them misses down to HBM (1 GiB table)."""
    cal = {"stream_bytes_per_fetch_count": None, "gather_bytes_per_fetch_count": None, "bytes_per_write_count": None}
    f, w = pmc.get("fetch") or {}, pmc.get("write") or {}
    if "k_ub_copy" in f and f["k_ub_copy"].get("FETCH_SIZE"):
        cal["stream_bytes_per_fetch_count"] = UB_COPY_BYTES*f["k_ub_copy"]["launches"]/(f["k_ub_copy"]["FETCH_SIZE"]*1024.0)
    if "k_ub_copy" in w and w["k_ub_copy"].get("WRITE_SIZE"):
        cal["bytes_per_write_count"] = UB_COPY_BYTES*w["k_ub_copy"]["launches"]/(w["k_ub_copy"]["WRITE_SIZE"]*1024.0)
    g = f.get("k_ub_gather<0>")
    if g and g.get("FETCH_SIZE"):
        cal["gather_bytes_per_fetch_count"] = 64.0*g["launches"]*cal_gather_visits()/(g["FETCH_SIZE"]*1024.0)
    return cal


_GATHER_VISITS = [None]


def cal_gather_visits():
    """records one k_ub_gather launch visits on this GPU (grid = 16 workgroups per CU x 256 lanes x UB_STEPS)"""
    if _GATHER_VISITS[0] is None:
        import torch
        _GATHER_VISITS[0] = torch.cuda.get_device_properties(0).multi_processor_count*16*256*UB_STEPS
    return _GATHER_VISITS[0]


def yard_sticks():
    """Stream-copy and record-chase rates of THIS GPU, now (N = 1, rank 0): the ceilings the path kernels are quoted against."""
    import tinsel_amd
    from tinsel_amd import renderer as R
    out = {}
    try:
        ms, units = tinsel_amd.ubench(R.UBENCH_COPY, 1 << 30)
        out["stream_copy_GBs"] = units/(ms*1e-3)/1e9
        for key, kind, table in (("gather_beyond_cache", R.UBENCH_GATHER_BEYOND_CACHE, UB_BIG_TABLE), ("gather_tree_sized", R.UBENCH_GATHER_TREE, UB_TREE_TABLE),
                                 ("gather_l2_resident", R.UBENCH_GATHER_L2, UB_L2_TABLE)):
            ms, units = tinsel_amd.ubench(kind, table, 256)
            out[key + "_Grecords_s"] = units/(ms*1e-3)/1e9
        out["what"] = ("tinsel_hip_ubench on this GPU in this run: float4 copy of 1 GiB (bytes read + written per second); dependent chases of 64-B records, "
                       "one chain per lane, 16 waves per CU, through tables of 1 GiB / 32 MiB (a 524k-triangle tree) / 2 MiB (inside one L2)")
    except Exception as e:
        out["error"] = str(e)
    return out


# ---------------------------------------------------------------------------------------------------------------------

def run_config(args, scene_name, width, height, maxdepth, rank, world, local, dist, backend, torch, with_extras, cfg_spp=256):
    """Times one configuration per the contract; returns the dict bench_detail.json keeps for it (rank 0; None elsewhere).
    `cfg_spp`: the samples per pixel BASELINE.json quotes the configuration at (the tolerance arm's L2 is taken at that spp)."""
    import tinsel_amd
    from tinsel_amd import abi

    pack = os.path.join(ROOT, "tests", "golden", scene_name + ".pack")
    scene = tinsel_amd.Scene.load_pack(pack)
    cam = scene.camera
    opt = scene.options.copy()
    opt.width, opt.height = width, height
    if maxdepth > 0:
        opt.max_depth = maxdepth
    elif scene_name == "glass":
        opt.max_depth = 12      # BASELINE.json configs[3]
    opt.mode = abi.MODE_PATHTRACE

    r = tinsel_amd.create_gpu_renderer(scene, local)
    bvh_build_ms = r.set_mesh_bvh(abi.BVH_LBVH if args.bvh == "lbvh" else abi.BVH_PLOC) if args.bvh != "reference" else None
    if args.roulette > 0:
        r.set_russian_roulette(args.roulette)
    r.set_pipeline({"auto": abi.PIPELINE_AUTO, "wavefront": abi.PIPELINE_WAVEFRONT, "mega": abi.PIPELINE_MEGAKERNEL, "split": abi.PIPELINE_WAVEFRONT_SPLIT, "paired": abi.PIPELINE_WAVEFRONT_PAIRED}[args.pipeline])
    if args.arith == "fast":
        r.set_arithmetic(abi.ARITH_FAST)
    if world > 1:
        r.set_shard(rank, world, args.tile)     # path slots are rank-local: the same batch size as N = 1 holds the same number of live paths
    accum = torch.zeros((opt.height, opt.width, 4), dtype=torch.float32, device="cuda")
    r.init(opt.width, opt.height, accum_tensor=accum)
    stream = torch.cuda.current_stream().cuda_stream
    r.reserve(max(args.steps, args.warmup, 1)*world, opt.max_depth)      # no hipMalloc inside the timed region

    passes_per_step = world         # weak scaling: K*N passes over 1/N of the pixels each
    from tinsel_amd import distributed
    # THE collective: the library's own ncclReduce (tinsel_hip_comm_*: the code a tinsel_hip_group's threads run too), its communicator made
    # here, outside every timed region; torch only carries the id.  rccl_seen = the ranks RCCL itself counts (ncclCommCount).  Not with a
    # CPU backend (gloo: the one-device stand-in, two ranks sharing a GPU, which RCCL refuses) -- and if ANY rank fails to join, every rank
    # falls back to torch's reduce and the line says so (`reduce_impl`).
    rccl_seen = 0
    if (world > 1 and backend == "nccl") or (world == 1 and args.force_comm):
        rccl_seen = distributed.init_library_comm(r, rank, world)
    use_lib = rccl_seen > 0
    reducing = world > 1 or use_lib
    reduce_impl = ("library ncclReduce (tinsel_hip_comm_reduce_accum)" if use_lib else
                   ("torch.distributed reduce, %s backend%s" % (backend, " (FALLBACK: the library communicator did not come up)" if backend == "nccl" else " (one-device stand-in)")
                    if world > 1 else None))
    total_buf = torch.empty_like(accum) if reducing else None     # the reduce target: allocated outside the timed region
    reduced = [accum]

    def run(steps, per_step=None):
        r.render_async(cam, opt, passes=steps*(passes_per_step if per_step is None else per_step), stream=stream)
        if reducing:
            # the ONE collective of the path, out of place: `accum` keeps this rank's own partial sums (a later render + reduce
            # cannot count a sample twice); RCCL over xGMI on the render stream, or the gloo stand-in through host memory
            reduced[0] = distributed.reduce_accum(accum, dst=0, out=total_buf, renderer=r if use_lib else None, rank=rank)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- algorithmic-bytes constants of this workload (device counters, untimed) --------------
    r.set_detail_counters(True)
    r.reset_stats()
    r.render_async(cam, opt, passes=passes_per_step, stream=stream)
    torch.cuda.synchronize()
    c = r.stats()
    r.set_detail_counters(False)
    # (the counted pass may run another pipeline -- the paired one counts nothing -- and so give the path buffers back: reserve again, outside
    # every timed region)
    r.reserve(max(args.steps, args.warmup, 1)*world, opt.max_depth)
    rays_c = max(1, c["rays"])
    I_bar, T_bar, P_bar = c["internal_visits"]/rays_c, c["tri_tests"]/rays_c, c["prim_tests"]/rays_c
    B_ray = 48.0 + 64.0*I_bar + 48.0*T_bar + 84.0*P_bar
    fw = opt.filter.width
    # pixels AddSample touches per sample (render.cpp:426-429: [int(x - fw), int(x + fw)] per axis, x uniform in the pixel):
    # 2 fw + 1 per axis on average away from the frame edge -- 6.25 for the default 0.75 (4..9)
    K_fp = (2.0*fw + 1.0)**2
    B_fb = 32.0*K_fp

    # ---- warmup --------------------------------------------------------------------------------
    if args.warmup > 0:
        run(args.warmup)
    sync()

    # ---- timed blocks of exactly K steps, repeated until MIN_TIMED_S ----------------------------
    first_timed_pass = None
    blocks, stats_blocks, ktimes = [], None, {}
    total = 0.0
    # The kernels' HIP events (two records per launch, on the launch stream) ride in ONE timed block, the LAST: every record is a barrier
    # packet of a few microseconds, which a small batch (cfg1: 0.32 ms of kernels) feels in every block -- and the first block after the
    # warm-up runs 10-15 % below the others (clocks), which made its kernel times disagree with rocprofv3's averages of the same command.
    while True:
        last = total >= MIN_TIMED_S or len(blocks) >= 1000          # (the block after the time is up: the one with the events)
        accum.zero_()
        r.reset_stats()
        r.enable_kernel_timing(last)
        sync()
        if first_timed_pass is None:
            first_timed_pass = r.get_pass_index()
        t0 = time.perf_counter()
        run(args.steps)
        sync()
        elapsed = time.perf_counter() - t0
        if world > 1:
            small_dev = "cuda" if backend == "nccl" else "cpu"
            tt = torch.tensor([elapsed], dtype=torch.float64, device=small_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        blocks.append(elapsed)
        total += elapsed
        if stats_blocks is None:
            stats_blocks = r.stats()
        if last:
            ktimes = r.kernel_times()           # of the last timed block's render call
            break
    r.enable_kernel_timing(False)
    elapsed = statistics.median(blocks)
    st = stats_blocks

    # ---- N > 1: the FIXED-WORK leg (north_star's 8-GPU configuration is a fixed job: veach 4K at 4096 spp).  The same K full-frame
    # passes as N = 1, split over the ranks' pixel tiles (K passes over 1/N of the pixels each) + the one reduce; same bracketing.
    strong = None
    if world > 1:
        sblocks, stotal = [], 0.0
        while stotal < MIN_TIMED_S/2 and len(sblocks) < 1000:
            accum.zero_()
            sync()
            t0 = time.perf_counter()
            run(args.steps, per_step=1)
            sync()
            el = time.perf_counter() - t0
            tt = torch.tensor([el], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            sblocks.append(float(tt.item()))
            stotal += sblocks[-1]
        sel = statistics.median(sblocks)
        strong = {"msamples_s": args.steps*opt.width*opt.height/sel/1e6, "ms_per_step": sel*1e3/args.steps, "timed_blocks": len(sblocks),
                  "what": "fixed work: the same %d full-frame passes as N = 1, each rank its own pixel tiles, one reduce" % args.steps}

    if world > 1:
        small_dev = "cuda" if backend == "nccl" else "cpu"
        cc = torch.tensor([st["rays"], st["samples"], st["shadow_rays"]], dtype=torch.float64, device=small_dev)
        dist.all_reduce(cc, op=dist.ReduceOp.SUM)
        tot_rays, tot_samples, tot_shadow = (float(x) for x in cc.tolist())
    else:
        tot_rays, tot_samples, tot_shadow = float(st["rays"]), float(st["samples"]), float(st["shadow_rays"])

    # per-rank view of the run (load balance of the pixel tiles): every rank's samples, rays, kernel time and median block
    per_rank = None
    if world > 1:
        mine = {"rank": rank, "samples": int(st["samples"]), "rays": int(st["rays"]), "kernel_ms": round(sum(v[2] for v in ktimes.values()), 3),
                "median_block_ms": round(statistics.median(blocks)*1e3, 3), "device": int(local)}
        gathered = [None]*world
        dist.all_gather_object(gathered, mine)
        per_rank = gathered

    if rank != 0:
        r.close()
        return None

    # validation mode only: the reduced image of the N-rank run must equal an unsharded render of the same passes
    if os.environ.get("TINSEL_BENCH_ONE_DEVICE") and world > 1:
        torch.cuda.synchronize()
        last_passes = args.steps            # (the last render was the fixed-work leg's: K passes)
        last_first = r.get_pass_index() - last_passes
        chk = tinsel_amd.create_gpu_renderer(scene, local)
        chk.init(opt.width, opt.height)
        chk.set_pass_index(last_first)
        want = chk.render(cam, opt, passes=last_passes)
        chk.close()
        got = reduced[0].cpu().numpy()
        ok = np.allclose(got, want, rtol=1e-4, atol=1e-5)
        print("validation: %d-rank reduced image vs unsharded render of passes [%d, %d): %s (max abs diff %.3e)" % (
            world, last_first, last_first + last_passes, "ok" if ok else "MISMATCH",
            float(np.abs(got - want).max())), file=sys.stderr, flush=True)
        if not ok:
            raise SystemExit(3)

    if world == 1 and use_lib:
        torch.cuda.synchronize()
        same = bool(torch.equal(reduced[0], accum))
        print("validation: 1-rank library ncclReduce of the accumulator: %s" % ("ok" if same else "MISMATCH"), file=sys.stderr, flush=True)
        if not same:
            raise SystemExit(3)

    # ---- the API's own call pattern, N = 1 only -------------------------------------------------
    pcie = api_1pass = api_1pass_plain = api_1pass_pinned = None
    if world == 1 and with_extras and not args.no_api:
        # a renderer of its own, with a library-owned accumulator like the C++ shim's (the timed one renders into a torch tensor)
        ra = tinsel_amd.create_gpu_renderer(scene, local)
        ra.init(opt.width, opt.height)
        ra.reserve(16, opt.max_depth)
        out = np.empty((opt.height, opt.width, 4), np.float32)
        ra.render(cam, opt, output=out, passes=1)               # first touch of the host buffer
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ra.render(cam, opt, output=out, passes=16)
        t3 = time.perf_counter()
        pcie = 16*opt.width*opt.height/(t3 - t2)/1e6
        # Renderer::Render exactly as main.cpp:246-250 calls it: ONE pass and the full-frame running sum per call
        # (render.cu:1099-1102) -- plain, then with the look-ahead the C++ shim turns on (tinsel_hip_set_lookahead)
        calls = 64
        t2 = time.perf_counter()
        for _ in range(calls):
            ra.render(cam, opt, output=out, passes=1)
        t3 = time.perf_counter()
        api_1pass_plain = calls*opt.width*opt.height/(t3 - t2)/1e6
        ra.set_lookahead(abi.LOOKAHEAD_ON)                      # what the C++ shim turns on: the caller's array stays pageable
        ra.render(cam, opt, output=out, passes=1)               # starts the pipeline
        t2 = time.perf_counter()
        for _ in range(calls):
            ra.render(cam, opt, output=out, passes=1)
        t3 = time.perf_counter()
        api_1pass = calls*opt.width*opt.height/(t3 - t2)/1e6
        ra.set_lookahead(abi.LOOKAHEAD_PIN_OUTPUT)              # opt-in: the caller's array page-locked in place
        ra.render(cam, opt, output=out, passes=1)
        t2 = time.perf_counter()
        for _ in range(calls):
            ra.render(cam, opt, output=out, passes=1)
        t3 = time.perf_counter()
        api_1pass_pinned = calls*opt.width*opt.height/(t3 - t2)/1e6
        ra.set_lookahead(abi.LOOKAHEAD_OFF)
        ra.close()

    # ---- the opt-in tolerance-arithmetic arm (tinsel_hip_set_arithmetic): rate and distance, N = 1 only ---------
    fast = None
    if world == 1 and not args.no_fast and args.arith == "exact":
        try:
            spp = cfg_spp           # the configuration's own spp (north_star's tolerance is quoted per configuration)
            # the distance is per pixel at that spp: where the whole frame at that spp is more than ~4 G samples per arm (veach 4K x 4096) it is
            # taken on the same view at 1/f of the resolution in both axes (same camera, same seeds rule; SURVEY.md 8c allows sub-sampling)
            f = 1
            while (opt.width//f)*(opt.height//f)*spp > 4.0e9 and f < 8:
                f += 1
            optL = opt.copy()
            optL.width, optL.height = opt.width//f, opt.height//f
            rf = tinsel_amd.create_gpu_renderer(scene, local)
            rf.set_pipeline({"auto": abi.PIPELINE_AUTO, "wavefront": abi.PIPELINE_WAVEFRONT, "mega": abi.PIPELINE_MEGAKERNEL, "split": abi.PIPELINE_WAVEFRONT_SPLIT, "paired": abi.PIPELINE_WAVEFRONT_PAIRED}[args.pipeline])
            rf.init(optL.width, optL.height)
            rf.reserve(max(spp, args.steps), opt.max_depth)
            exact_img = rf.render(cam, optL, passes=spp)
            rf.set_arithmetic(abi.ARITH_FAST)
            rf.init(optL.width, optL.height)
            rf.set_pass_index(0)
            fast_img = rf.render(cam, optL, passes=spp)
            wa = np.where(exact_img[..., 3:4] > 0, exact_img[..., 3:4], 1.0)
            wb = np.where(fast_img[..., 3:4] > 0, fast_img[..., 3:4], 1.0)
            dd = (exact_img[..., :3]/wa - fast_img[..., :3]/wb).astype(np.float64)
            l2 = float(np.sqrt(np.mean(np.sum(dd*dd, axis=-1))))
            rf.init(opt.width, opt.height)          # (back at the configuration's own frame, still on the tolerance arm)
            # one pass, path by path: how many paths left the exact path's track (radiance off by > 1e-3 relative)
            rf.set_pass_index(0); rf.render(cam, opt, passes=1, readback=False); rad_f = rf.batch_radiance(1, opt.height, opt.width)
            rf.set_arithmetic(abi.ARITH_EXACT)
            rf.set_pass_index(0); rf.render(cam, opt, passes=1, readback=False); rad_e = rf.batch_radiance(1, opt.height, opt.width)
            rel = np.abs(rad_f - rad_e).max(axis=-1)/np.maximum(1e-3, np.abs(rad_e).max(axis=-1))
            rf.set_arithmetic(abi.ARITH_FAST)
            fblocks, ftotal = [], 0.0
            rf.render(cam, opt, passes=max(1, args.warmup), readback=False)
            while ftotal < 0.25 and len(fblocks) < 1000:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                rf.render(cam, opt, passes=args.steps, readback=False)
                fblocks.append(time.perf_counter() - t0)
                ftotal += fblocks[-1]
            rf.close()
            fast = {"msamples_s": args.steps*opt.width*opt.height/statistics.median(fblocks)/1e6, "l2_vs_exact_at_spp": [l2, spp], "l2_frame": [optL.width, optL.height],
                    "divergent_paths_fraction": float((rel > 1e-3).mean()), "identical_paths_fraction": float((rad_f == rad_e).all(axis=-1).mean())}
        except Exception as e:
            fast = {"msamples_s": None, "error": str(e)}

    # ---- roofline: every kernel of the block, the dominant one (by time, over ALL kernels) in the headline fields ------------
    # a kernel's time: (launches, sum of the launches' durations, union of their intervals).  Where a call's two chunks run on two streams
    # (tinsel_hip.hip render_impl) launches of one kernel overlap: RATES divide by the union ("busy"); the average launch duration is the
    # sum over the launches, which is what rocprofv3 --stats reports, and `concurrent_launches` = sum / union says how many ran at once
    gpu_ms = sum(v[2] for v in ktimes.values())
    dom = max(ktimes.items(), key=lambda kv: kv[1][2], default=(None, (0, 0.0, 0.0)))
    dom_name, (dom_launches, dom_ms, dom_busy) = dom
    avg_launch_s = dom_ms*1e-3/max(1, dom_launches)
    dom_concurrency = (dom_ms/dom_busy) if dom_busy > 0 else 1.0
    rays = st["rays"]
    # algorithmic bytes by kernel (SURVEY.md 8d's B_ray split over the kernels that do the work): the mesh walk (Node64 visits + triangle
    # tests) belongs to k_walk when it runs, the primitive tests and ray/hit records to the scan kernels; path-state streaming (k_shade,
    # k_lights, k_generate, k_seg) and the framebuffer gather have no algorithmic figure in that model (8d: "an implementation artefact")
    alg = {"k_bounce": rays*B_ray, "k_mega": rays*B_ray}
    if "k_walk" in ktimes:
        alg["k_walk"] = rays*(64.0*I_bar + 48.0*T_bar)
        alg["k_extend"] = (rays - st["shadow_rays"])*(48.0 + 84.0*P_bar)
        alg["k_shadow"] = st["shadow_rays"]*(48.0 + 84.0*P_bar)
    else:
        alg["k_extend"] = (rays - st["shadow_rays"])*B_ray
        alg["k_shadow"] = st["shadow_rays"]*B_ray
    alg["k_accumulate"] = st["samples"]*B_fb
    job_bytes = rays*B_ray + st["samples"]*B_fb
    # what a wavefront path tracer with the scene on chip MUST move through HBM: ray in / hit out per ray, the footprint's RMW per sample
    job_compulsory = rays*48.0 + st["samples"]*B_fb

    pmc = {"sq": None, "fetch": None, "write": None, "tcc": None}
    if world == 1 and not args.no_pmc and dom_name:
        for key in ("sq", "fetch", "write", "tcc"):
            pmc[key] = pmc_pass(args, scene_name, width, height, opt.max_depth, args.steps, PMC_SETS[key])
    cal = calibrate(pmc)
    # what a count stands for: calibrated on this run's yard-stick kernels; without them the guide's figures (x2 for wide
    # coalesced streaming reads, MI355X_MICROARCH.md HBM section; everything else taken at face value) -- and the line says which
    f_stream = cal["stream_bytes_per_fetch_count"] or 2.0
    f_gather = cal["gather_bytes_per_fetch_count"] or 1.0
    f_write = cal["bytes_per_write_count"] or 1.0

    def kernel_traffic(name):
        """counter bytes per launch of one kernel: FETCH_SIZE / WRITE_SIZE (KB) x the calibrated bytes per count of its access shape"""
        if not (pmc["fetch"] and pmc["write"] and name in pmc["fetch"] and name in pmc["write"]):
            return None
        f, w = pmc["fetch"][name], pmc["write"][name]
        ff = f_gather if name in GATHER_KERNELS else f_stream
        return (ff*f["FETCH_SIZE"]/f["launches"] + f_write*w["WRITE_SIZE"]/w["launches"])*1024.0

    copy_gbs = YARD[0].get("stream_copy_GBs") if YARD[0] else None

    def kernel_row(name):
        """one row of roofline.kernels[]: time from the library's HIP events (this block), counters from this run's rocprofv3 passes"""
        launches, sum_ms, ms = ktimes[name]           # (every rate below is over `ms`, the kernel's busy time)
        row = {"kernel": name, "launches": launches, "ms": round(ms, 4), "sum_of_launch_ms": round(sum_ms, 4), "share_of_gpu_time": (ms/gpu_ms) if gpu_ms > 0 else None,
               "frac_model": None, "frac": None, "counter_GB": None, "counter_GBs": None, "frac_hbm_counter": None, "frac_of_stream_copy": None,
               "algorithmic_GBs": None, "valu_frac_of_issue_peak": None, "valu_lanes_active": None, "wave_cycles_waiting": None, "waves_per_simd": None, "l2_hit_rate": None}
        t = kernel_traffic(name)
        if t and ms > 0:
            row["counter_GB"] = t*launches/1e9
            row["counter_GBs"] = t*launches/(ms*1e-3)/1e9
            row["frac_hbm_counter"] = row["counter_GBs"]/HBM_PEAK_GBS
            if copy_gbs:
                row["frac_of_stream_copy"] = row["counter_GBs"]/copy_gbs
        if name in alg and ms > 0:
            row["algorithmic_GBs"] = alg[name]/(ms*1e-3)/1e9
        if pmc["sq"] and name in pmc["sq"]:
            q = pmc["sq"][name]
            iv = q.get("SQ_INSTS_VALU", 0.0)
            if iv and ms > 0:
                # (the profiled pass launches what the timed block launches: per-launch counts x this block's launches / this block's time)
                row["valu_frac_of_issue_peak"] = iv/q["launches"]*launches/(ms*1e-3)/VALU_PEAK
                row["valu_lanes_active"] = q.get("SQ_THREAD_CYCLES_VALU", 0.0)/(64.0*iv)
            if q.get("SQ_WAVE_CYCLES"):
                row["wave_cycles_waiting"] = q.get("SQ_WAIT_ANY", 0.0)/q["SQ_WAVE_CYCLES"]
                if q.get("GRBM_GUI_ACTIVE"):
                    row["waves_per_simd"] = 4.0*q["SQ_WAVE_CYCLES"]/(q["GRBM_GUI_ACTIVE"]/8.0*SIMDS)
        if pmc["tcc"] and name in pmc["tcc"]:
            q = pmc["tcc"][name]
            if q.get("TCC_HIT_sum", 0.0) + q.get("TCC_MISS_sum", 0.0) > 0:
                row["l2_hit_rate"] = q["TCC_HIT_sum"]/(q["TCC_HIT_sum"] + q["TCC_MISS_sum"])
        # which model the kernel's `frac` comes from: instruction issue where the scene never leaves the CU, else HBM bytes (counter
        # bytes when this run measured them, the algorithmic figure otherwise)
        if name in LDS_SCENE_KERNELS:
            row["frac_model"], row["frac"] = "valu_issue", row["valu_frac_of_issue_peak"]
        elif row["frac_hbm_counter"] is not None:
            row["frac_model"], row["frac"] = "hbm_counter", row["frac_hbm_counter"]
        elif row["algorithmic_GBs"] is not None:
            row["frac_model"], row["frac"] = "hbm_algorithmic", row["algorithmic_GBs"]/HBM_PEAK_GBS
        return row

    kernels = sorted((kernel_row(k) for k in ktimes), key=lambda r: -r["ms"])
    drow = kernels[0] if kernels else {}
    traffic = kernel_traffic(dom_name) if dom_name else None
    alg_gbs = drow.get("algorithmic_GBs") or 0.0
    valu_per_launch = None
    if pmc["sq"] and dom_name in pmc["sq"] and pmc["sq"][dom_name].get("SQ_INSTS_VALU"):
        valu_per_launch = pmc["sq"][dom_name]["SQ_INSTS_VALU"]/pmc["sq"][dom_name]["launches"]
    counter_job = sum(r["counter_GB"] for r in kernels if r["counter_GB"]) if any(r["counter_GB"] for r in kernels) else None

    common = {
        "kernel": dom_name, "launches": dom_launches, "avg_launch_ms": avg_launch_s*1e3, "concurrent_launches": dom_concurrency, "traffic": traffic,
        "frac_model": drow.get("frac_model"),
        "algorithmic_GBs": drow.get("algorithmic_GBs"), "frac_hbm_algorithmic": (alg_gbs/HBM_PEAK_GBS) if alg_gbs else None,
        # SURVEY.md 8(d)'s own number, as written there: (rays x B_ray + samples x B_fb) / seconds / 8e12 over the TIMED step (the driver's
        # clock, not the kernel's).  Above 1 wherever the scene is on chip: the model bills every node / primitive fetch to HBM
        "frac_survey_8d": job_bytes/elapsed/(HBM_PEAK_GBS*1e9) if elapsed > 0 else None,
        # what the job MUST move through HBM with the scene on chip (48 B per ray + the footprint's RMW per sample) over the same seconds
        "frac_hbm_compulsory": job_compulsory/elapsed/(HBM_PEAK_GBS*1e9) if elapsed > 0 else None,
        # issue utilisation x lanes active: the share of the chip's peak LANE-operations per second the dominant kernel's instructions use
        # (every issued instruction still counts as useful: the parity arm's IEEE divide / sqrt expansions are in it)
        "useful_frac": (drow.get("valu_frac_of_issue_peak")*drow.get("valu_lanes_active")) if (drow.get("valu_frac_of_issue_peak") and drow.get("valu_lanes_active")) else None,
        "valu_frac_of_issue_peak": drow.get("valu_frac_of_issue_peak"),
        "counter_GBs": drow.get("counter_GBs"), "frac_hbm_counter": drow.get("frac_hbm_counter"),
        "l2_hit_rate": drow.get("l2_hit_rate"),
        "valu_wave_insts_per_launch": valu_per_launch, "valu_lanes_active": drow.get("valu_lanes_active"), "wave_cycles_waiting": drow.get("wave_cycles_waiting"),
        "waves_per_simd": drow.get("waves_per_simd"),
        "B_ray": B_ray, "I": I_bar, "T": T_bar, "P": P_bar, "B_fb": B_fb, "K_fp": K_fp,
        "job_algorithmic_GBs": job_bytes/(gpu_ms*1e-3)/1e9 if gpu_ms > 0 else 0.0,
        # the job as a whole: HBM bytes the counters saw over what it must move with the scene on chip (48 B per ray + the footprint per sample)
        "job_counter_GB": counter_job, "job_compulsory_GB": job_compulsory/1e9,
        "job_counter_over_compulsory": (counter_job*1e9/job_compulsory) if (counter_job and job_compulsory > 0) else None,
        "kernels": kernels,
        "kernel_ms": {k: round(v[2], 3) for k, v in ktimes.items()},
        "kernel_traffic_GB": {r["kernel"]: (round(r["counter_GB"], 3) if r["counter_GB"] else None) for r in kernels} if any(pmc.values()) else None,
        "counter_calibration": dict(cal, source=("k_ub_copy / k_ub_gather<0> in the same rocprofv3 passes" if any(cal.values()) else
                                                 "none measured: FETCH_SIZE x2 on streaming kernels (the guide), x1 elsewhere"),
                                    gather_kernels=list(GATHER_KERNELS)),
        "counters": "rocprofv3 --pmc passes of this run (same workload, same passes per launch)" if any(pmc.values()) else None,
    }
    if copy_gbs:
        common["stream_copy_GBs"] = copy_gbs
        common["frac_of_stream_copy"] = drow.get("frac_of_stream_copy")
    # (only where k_walk is the dominant kernel: there it does ALL the mesh visits the device counters I, T count -- config 3; in a scene
    # that also walks small meshes inline, glass, the per-ray constants are not k_walk's alone)
    wrow = next((r for r in kernels if r["kernel"] == "k_walk"), None) if dom_name == "k_walk" else None
    if wrow and wrow["ms"] > 0:
        # what the walk is made of: Node64 visits (and triangle tests) per second, next to the record-chase rates this GPU
        # sustains on a table of the tree's size and on one that fits an XCD's L2 (tinsel_hip_ubench, this run)
        walk = {"node_visits_G_s": rays*I_bar/(wrow["ms"]*1e-3)/1e9, "triangle_tests_G_s": rays*T_bar/(wrow["ms"]*1e-3)/1e9}
        if YARD[0]:
            walk["record_chase_ceilings_G_s"] = {k: YARD[0].get(k) for k in ("gather_beyond_cache_Grecords_s", "gather_tree_sized_Grecords_s", "gather_l2_resident_Grecords_s")}
            # records per second (a 48-B triangle = 0.75 of a 64-B record) against the two ceilings: above the first (the tree top in
            # LDS and the L2 serve most visits), below the second (what a CU's vector-memory front end retires from L2)
            recs = walk["node_visits_G_s"] + 0.75*walk["triangle_tests_G_s"]
            if YARD[0].get("gather_tree_sized_Grecords_s"):
                walk["frac_of_tree_sized_chase"] = recs/YARD[0]["gather_tree_sized_Grecords_s"]
            if YARD[0].get("gather_l2_resident_Grecords_s"):
                walk["frac_of_l2_resident_chase"] = recs/YARD[0]["gather_l2_resident_Grecords_s"]
        common.update(walk)         # (flat, as before) ...
        common["k_walk"] = walk     # ... and under the kernel's name, whichever kernel is the dominant one
    if drow.get("frac_model") == "valu_issue":
        # the scene never leaves the CU: the HBM model counts bytes that are LDS reads.  What binds is instruction issue.
        # (per-launch instructions / average launch duration x the launches that ran at once = all of them / the kernel's busy time)
        ach = (valu_per_launch*dom_concurrency/avg_launch_s/1e9) if (valu_per_launch and avg_launch_s > 0) else None
        roofline = dict({"bound": "valu", "achieved": ach, "peak": VALU_PEAK/1e9, "unit": "G wave-instructions/s",
                         "frac": (ach/(VALU_PEAK/1e9)) if ach else None}, **common)
    else:
        # counter bytes when this run measured them, else the algorithmic figure (never a stored constant)
        ach = common["counter_GBs"] if common["counter_GBs"] else alg_gbs
        roofline = dict({"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (ach/HBM_PEAK_GBS) if ach else None,
                         "achieved_is": "counter bytes (calibrated)" if common["counter_GBs"] else "algorithmic bytes"}, **common)

    cpu = None
    if not args.no_cpu_baseline and world == 1:        # the CPU leg is timed at N = 1 only
        try:
            cpu = cpu_baseline(scene_name, cam, opt, args.cpu_seconds)
        except Exception as e:      # a checker built for another box must not kill the bench line
            cpu = {"value": None, "unit": "Msamples/s", "cores": 0, "kind": "reference", "sample": "unavailable: %s" % e}

    msamples = tot_samples/elapsed/1e6
    res = {
        "metric": "Msamples/s (%s.tin %dx%d maxDepth=%d spp=%d, wavefront path; Mrays/s alongside)" % (
            scene_name, opt.width, opt.height, opt.max_depth, args.steps*passes_per_step),
        "value": msamples, "unit": "Msamples/s", "ms_per_step": elapsed*1e3/args.steps,
        "timed_blocks": len(blocks), "timed_seconds": total, "block_ms_min_median_max": [min(blocks)*1e3, elapsed*1e3, max(blocks)*1e3],
        "config": {"workload": "%s.tin %dx%d maxDepth=%d, %d pass(es) per step, pipeline=%s" % (
            scene_name, opt.width, opt.height, opt.max_depth, passes_per_step,
            args.pipeline if args.pipeline != "auto" else "auto->" + ("wavefront(fused)" if "k_bounce" in ktimes else "wavefront(%s%s)" % ("paired" if "k_step" in ktimes else "split", "+k_walk" if "k_walk" in ktimes else ""))),
            "scene_pack": os.path.relpath(pack, ROOT), "parallelism": "pixel-tile shard x%d + RCCL reduce" % world if world > 1 else "1 GPU",
            "filter": "gaussian w=%.2f" % fw, "rays_per_sample": tot_rays/max(1.0, tot_samples),
            "mesh_bvh": args.bvh, "mesh_bvh_build_ms": bvh_build_ms, "russian_roulette_from_bounce": args.roulette},
        "mrays_per_s": tot_rays/elapsed/1e6,
        "shadow_ray_fraction": tot_shadow/max(1.0, tot_rays),
        "gpu_kernel_ms_total": gpu_ms,
        "pcie_inclusive_msamples_s": pcie,
        "api_1pass_msamples_s": api_1pass,
        "api_1pass_plain_msamples_s": api_1pass_plain,
        "api_1pass_pinned_output_msamples_s": api_1pass_pinned,
        "arithmetic": args.arith,
        "reduce": {"impl": reduce_impl, "rccl_ranks_seen": rccl_seen if use_lib else None},
        "fast_msamples_s": fast["msamples_s"] if fast else None,
        "fast_l2": fast["l2_vs_exact_at_spp"][0] if (fast and fast.get("l2_vs_exact_at_spp")) else None,
        # what bit-exactness costs: the opt-in tolerance arm's rate over the timed (exact) arm's, same workload
        "fast_over_exact": (fast["msamples_s"]/msamples) if (fast and fast.get("msamples_s") and args.arith == "exact") else None,
        "fast": fast,
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    if world > 1:
        res["ranks"] = {"communicator_world_size": dist.get_world_size(), "backend": backend, "per_rank": per_rank}
        res["strong"] = strong
    r.close()
    return res


def group_bench(args):
    """`--group`: the library's OWN multi-GPU path -- tinsel_hip_group over args.gpus devices inside this one process (what the
    reference's single-threaded C++ caller gets through shim/hip_renderer.cpp): thread per device, pixel-tile shards, one
    ncclReduce per read-back, D2H.  Timed the two ways a caller uses it: K*N passes per call with one read-back (the headless
    driver), and the reference's own pattern -- ONE pass + a full-frame read-back per call (main.cpp:246-250) -- plain and with
    the group's look-ahead.  Prints one JSON line."""
    import torch
    import tinsel_amd
    from tinsel_amd import abi
    n = max(1, args.gpus)
    one_device = torch.cuda.device_count() < n
    if one_device:
        os.environ["TINSEL_HIP_GROUP_ONE_DEVICE"] = "1"         # validation of the code path on a smaller box, not a measurement
    scene = tinsel_amd.Scene.load_pack(os.path.join(ROOT, "tests", "golden", args.scene + ".pack"))
    cam, opt = scene.camera, scene.options.copy()
    opt.width, opt.height, opt.mode = args.width, args.height, abi.MODE_PATHTRACE
    if args.maxdepth > 0:
        opt.max_depth = args.maxdepth
    grp = tinsel_amd.HipRendererGroup(scene, n, args.tile)
    grp.init(opt.width, opt.height)
    out = np.empty((opt.height, opt.width, 4), np.float32)
    K = args.steps*n
    grp.render(cam, opt, output=out, passes=max(1, args.warmup))
    blocks, total = [], 0.0
    while total < MIN_TIMED_S and len(blocks) < 1000:
        t0 = time.perf_counter()
        grp.render(cam, opt, output=out, passes=K)
        blocks.append(time.perf_counter() - t0)
        total += blocks[-1]
    kpass = K*opt.width*opt.height/statistics.median(blocks)/1e6

    def one_pass_calls(calls):
        grp.render(cam, opt, output=out, passes=1)
        t0 = time.perf_counter()
        for _ in range(calls):
            grp.render(cam, opt, output=out, passes=1)
        return calls*opt.width*opt.height/(time.perf_counter() - t0)/1e6

    calls = 64
    plain = one_pass_calls(calls)
    grp.set_lookahead(abi.LOOKAHEAD_ON)
    ahead = one_pass_calls(calls)
    grp.set_lookahead(abi.LOOKAHEAD_PIN_OUTPUT)
    ahead_pinned = one_pass_calls(calls)
    grp.close()
    print(json.dumps({
        "metric": "Msamples/s through tinsel_hip_group (%s.tin %dx%d maxDepth=%d), read-back to host memory included" % (args.scene, opt.width, opt.height, opt.max_depth),
        "n_gpus": n, "one_device_validation": one_device, "unit": "Msamples/s",
        "kpass_msamples_s": kpass, "kpass_passes_per_call": K, "kpass_calls_timed": len(blocks),
        "api_1pass_plain_msamples_s": plain, "api_1pass_lookahead_msamples_s": ahead, "api_1pass_lookahead_pinned_output_msamples_s": ahead_pinned,
        "calls": calls}), flush=True)


def group_leg(args, world, scene=None, width=None, height=None, maxdepth=None):
    """N > 1, rank 0, before the ranks meet: the same N devices through tinsel_hip_group in a child process with a time limit
    (a hung collective must not take the scaling run with it).  Default: the run's own workload; main() also asks for BASELINE
    configs[4] (veach.tin at 4K: north_star's 8-GPU configuration)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK")}
    if os.environ.get("TINSEL_BENCH_ONE_DEVICE"):
        env["TINSEL_HIP_GROUP_ONE_DEVICE"] = "1"
    cmd = [sys.executable, os.path.abspath(__file__), "--group", "--gpus", str(world), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--scene", scene or args.scene, "--width", str(width or args.width), "--height", str(height or args.height),
           "--maxdepth", str(args.maxdepth if maxdepth is None else maxdepth), "--tile", str(args.tile)]
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=GROUP_LEG_LIMIT_S)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"unavailable": "exit %d: %s" % (p.returncode, (p.stderr or "").strip().splitlines()[-1:] or "")}
    except subprocess.TimeoutExpired:
        return {"unavailable": "timed out after %d s" % GROUP_LEG_LIMIT_S}
    except Exception as e:
        return {"unavailable": str(e)}


def self_spawn(args):
    """`python bench.py --gpus N` started WITHOUT a launcher (no RANK in the environment): re-executes itself under torch.distributed.run
    with one process per GPU, exactly as the driver's own N-rank command line would -- so both launch styles give N ranks and one JSON
    line with n_gpus = N."""
    import socket
    import torch
    n = args.gpus
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n and not os.environ.get("TINSEL_BENCH_ONE_DEVICE"):
        raise SystemExit("bench.py --gpus %d: %d GPU(s) visible (TINSEL_BENCH_ONE_DEVICE=1 + TINSEL_BENCH_BACKEND=gloo run the N-rank code "
                         "path on one device: a validation of the path, not a measurement)" % (n, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: no launcher in the environment -- starting %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def first_collective_watchdog(args, rank, world, backend, seconds=None):
    """A timer around the rendezvous + first all-reduce.  When it fires, rank 0 prints the contract's JSON line with value null and a
    readable `unavailable` reason, and every rank leaves (os._exit: a hung collective cannot be interrupted)."""
    import threading
    limit = seconds if seconds is not None else 90.0

    def fire():
        if rank == 0:
            print(json.dumps({
                "metric": "Msamples/s (%s.tin %dx%d, wavefront path)" % (args.scene, args.width, args.height), "value": None, "unit": "Msamples/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32",
                "unavailable": "the %d ranks did not complete their rendezvous + first all-reduce (%s backend%s) within %.0f s: a rank that never "
                               "started, or a collective that does not come up on this node" % (world, backend, " = RCCL over xGMI" if backend == "nccl" else "", limit)}), flush=True)
        os._exit(4)

    t = threading.Timer(limit, fire)
    t.daemon = True
    t.start()
    return t


# ---------------------------------------------------------------------------------------------------------------------
# the contract line: SMALL (the driver reads it from a bounded window of stdout: round 4's 28.5 KB line was not parsed).  Everything
# else -- per-kernel tables, calibration, yard-sticks, per-rank statistics, the API legs' details -- goes to bench_detail.json.

LINE_LIMIT_BYTES = 6144
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "frac_model", "kernel", "launches", "avg_launch_ms", "traffic",
                 "frac_survey_8d", "useful_frac", "frac_hbm_compulsory", "frac_hbm_counter", "frac_hbm_algorithmic", "valu_frac_of_issue_peak",
                 "job_counter_over_compulsory", "valu_lanes_active", "wave_cycles_waiting", "waves_per_simd", "l2_hit_rate")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample")
BASELINE_INDEX = {"cornell": 1, LARGE: 2, REAL_MESH: 2, "glass": 3, "veach": 4}      # scene -> index into BASELINE.json's `configs`
CONFIG_SPP = {"cornell": 256, LARGE: 512, REAL_MESH: 512, "glass": 1024, "veach": 4096}


def _round(o, digits=5):
    """floats to `digits` significant digits, recursively (the line is for reading and parsing, the detail file keeps everything)"""
    if isinstance(o, float):
        return float("%.*g" % (digits, o)) if o == o and abs(o) != float("inf") else None
    if isinstance(o, dict):
        return {k: _round(v, digits) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_round(v, digits) for v in o]
    return o


def _pick(d, keys):
    return {k: d.get(k) for k in keys} if isinstance(d, dict) else None


def compact_config(res):
    """one BASELINE configuration in a few fields (the whole record is in bench_detail.json)"""
    if not isinstance(res, dict) or "value" not in res:
        return {"workload": (res or {}).get("config", {}).get("workload"), "unavailable": (res or {}).get("unavailable", "not run")}
    rf, cpu = res.get("roofline") or {}, res.get("cpu_baseline") or {}
    out = {"workload": res["config"]["workload"], "baseline_config": res.get("baseline_config"), "value": res["value"], "ms_per_step": res["ms_per_step"],
           "mrays_per_s": res.get("mrays_per_s"), "kernel": rf.get("kernel"), "avg_launch_ms": rf.get("avg_launch_ms"), "bound": rf.get("bound"),
           "frac": rf.get("frac"), "frac_model": rf.get("frac_model"), "frac_survey_8d": rf.get("frac_survey_8d"), "useful_frac": rf.get("useful_frac"),
           "frac_hbm_compulsory": rf.get("frac_hbm_compulsory"), "frac_hbm_counter": rf.get("frac_hbm_counter"),
           "job_counter_over_compulsory": rf.get("job_counter_over_compulsory"),
           "cpu_msamples_s": cpu.get("value"), "cpu_cores": cpu.get("cores"),
           "fast_over_exact": res.get("fast_over_exact"), "fast_l2_at_spp": (res.get("fast") or {}).get("l2_vs_exact_at_spp")}
    if res.get("strong"):
        out["strong_msamples_s"] = out["value_fixed_work"] = res["strong"]["msamples_s"]
    return out


def compact_group(g):
    if not isinstance(g, dict):
        return None
    if "unavailable" in g:
        return {"unavailable": str(g["unavailable"])[:160]}
    return _pick(g, ("kpass_msamples_s", "api_1pass_plain_msamples_s", "api_1pass_lookahead_msamples_s", "one_device_validation"))


def contract_line(args, world, head, more=(), group=None, group5=None, detail_file=None):
    """The ONE JSON line of the contract, from the full records (`head`: the headline configuration's; `more`: the other BASELINE
    configurations').  Held under LINE_LIMIT_BYTES at any N (tests/test_bench_host.py)."""
    line = {
        "metric": head["metric"], "value": head["value"], "unit": head["unit"], "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        # the same K full-frame passes as N = 1 split over the ranks (north_star's 8-GPU configuration is a FIXED job); at N = 1 the two coincide
        "value_fixed_work": (head.get("strong") or {}).get("msamples_s") if world > 1 else head["value"],
        "reduce_impl": (head.get("reduce") or {}).get("impl"), "rccl_ranks_seen": (head.get("reduce") or {}).get("rccl_ranks_seen"),
        "data": "synthetic: the reference's scene files as scene packs (ajax.obj is a missing blob: a procedural 524,288-triangle stand-in, and the reference's Aphrodite scan x4); rays, seeds and all downstream generated on the GPU",
        "config": _pick(head["config"], ("workload", "scene_pack", "parallelism", "rays_per_sample")),
        "mrays_per_s": head.get("mrays_per_s"),
        "roofline": _pick(head.get("roofline"), ROOFLINE_KEYS),
        "cpu_baseline": _pick(head.get("cpu_baseline"), CPU_KEYS),
        "fast_over_exact": head.get("fast_over_exact"),
        "fast_l2_at_spp": (head.get("fast") or {}).get("l2_vs_exact_at_spp"),
        "api_msamples_s": {"kpass_readback": head.get("pcie_inclusive_msamples_s"), "one_pass_plain": head.get("api_1pass_plain_msamples_s"),
                           "one_pass_lookahead": head.get("api_1pass_msamples_s"), "one_pass_pinned_output": head.get("api_1pass_pinned_output_msamples_s")},
    }
    if world > 1:
        line["api_msamples_s"] = None
        st = head.get("strong") or {}
        line["strong_msamples_s"], line["strong_ms_per_step"] = st.get("msamples_s"), st.get("ms_per_step")
        rk = head.get("ranks") or {}
        kms = [x.get("kernel_ms") for x in (rk.get("per_rank") or []) if x and x.get("kernel_ms") is not None]
        line["ranks"] = {"communicator_world_size": rk.get("communicator_world_size"), "backend": rk.get("backend"),
                         "kernel_ms_min_max": [min(kms), max(kms)] if kms else None}
        if group is not None:
            line["group"] = compact_group(group)
        if group5 is not None:
            line["group_cfg5"] = compact_group(group5)
    if more:
        line["configs"] = [compact_config(m) for m in more]
    if detail_file:
        line["detail"] = detail_file
    return _round(line)


def write_detail(detail):
    """bench_detail.json beside bench.py (and under gpurun_out/ when that exists, so that it travels back from the GPU box); the same
    record on stderr behind a prefix -- never on stdout, which carries the one contract line."""
    text = json.dumps(detail)
    name = None
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    f.write(text + "\n")
                name = name or "bench_detail.json"
            except OSError:
                pass
    print("bench_detail: " + text, file=sys.stderr, flush=True)
    return name


def main():
    args = parse()
    if args.tuning:
        from tinsel_amd import abi, renderer
        renderer.DEFAULT_TUNING = abi.Tuning(**json.loads(args.tuning))     # (every renderer this process creates; never set by the default run)
    if args.inner_pmc:
        return inner_pmc(args)
    if args.group:
        return group_bench(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        return self_spawn(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # TINSEL_BENCH_BACKEND=gloo + TINSEL_BENCH_ONE_DEVICE=1: run the N-rank code path on ONE GPU (validation of the
    # launch / shard / reduce / reporting logic on a single-GPU box; not a measurement)
    backend = os.environ.get("TINSEL_BENCH_BACKEND", "nccl")
    if os.environ.get("TINSEL_BENCH_ONE_DEVICE"):
        local = 0
    torch.cuda.set_device(local)
    default_headline = (args.scene, args.width, args.height) == ("cornell", 1024, 1024)
    if world == 1 and not args.no_ubench:
        YARD[0] = yard_sticks()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the FIRST collective under a watchdog: if the ranks cannot meet (a rank that never started, an RCCL ring that does not come up),
        # rank 0 still prints ONE readable JSON line instead of hanging until the driver's clock runs out.  Nothing runs before it: the
        # ranks meet within seconds of their start (the tinsel_hip_group legs come AFTER the ranks' timed regions).
        dog = first_collective_watchdog(args, rank, world, backend)
        import datetime
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local), timeout=datetime.timedelta(seconds=300))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
        hello = torch.ones(1, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(hello)
        if backend == "nccl":
            torch.cuda.synchronize()
        dog.cancel()
        if int(hello.item()) != world:
            raise SystemExit("bench.py: the first all-reduce over %d ranks summed to %d" % (world, int(hello.item())))

    head = run_config(args, args.scene, args.width, args.height, args.maxdepth, rank, world, local, dist, backend, torch, with_extras=True,
                      cfg_spp=CONFIG_SPP.get(args.scene, 256))
    if head is not None:
        head["baseline_config"] = BASELINE_INDEX.get(args.scene) if default_headline else None

    more = []

    def another(name, w, h, d, what):
        try:
            res = run_config(args, name, w, h, d, rank, world, local, dist, backend, torch, with_extras=False, cfg_spp=CONFIG_SPP.get(name, 256))
            if res is not None:
                res["baseline_config"] = BASELINE_INDEX.get(name)
                more.append(res)
        except Exception as e:
            if world > 1:
                raise           # (a rank that leaves a collective sequence would hang the others: fail the run loudly instead)
            more.append({"config": {"workload": what}, "unavailable": "failed: %s" % e})

    if world == 1 and default_headline and not args.no_second_config:
        # BASELINE configs[2]: the scene that lives in HBM
        what = "ajax stand-in (524,288 triangles) 1920x1080 maxDepth=4"
        if os.path.exists(os.path.join(ROOT, "tests", "golden", LARGE + ".pack")):
            another(LARGE, 1920, 1080, 4, what)
        else:
            more.append({"config": {"workload": what},
                         "unavailable": "tests/golden/large/ajax_standin.pack is not on this box (48 MB, git-ignored; written by tests/golden/make_large.py)"})
        # ... and with a REAL scanned mesh in the stand-in's place (SURVEY.md 0.1's other option): an irregular SAH tree
        if os.path.exists(os.path.join(ROOT, "tests", "golden", REAL_MESH + ".pack")):
            another(REAL_MESH, 1920, 1080, 4, "ajax.tin + Aphrodite scan (427,384 triangles) 1920x1080 maxDepth=4")
    if default_headline and not args.no_more_configs:
        if world == 1:
            another("glass", 1920, 1080, 12, "glass.tin 1920x1080 maxDepth=12")     # BASELINE configs[3]
        # BASELINE configs[4], north_star's multi-GPU configuration: at N > 1 the ranks' pixel tiles of the 4K frame, weak and fixed-work
        another("veach", 3840, 2160, 0, "veach.tin 3840x2160")

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return

    # N > 1, rank 0, AFTER the ranks' timed regions (the other ranks have left): the same N devices through the library's own multi-GPU
    # path (tinsel_hip_group: thread per device, one ncclReduce per read-back) in child processes with a 60 s limit each
    group = group5 = None
    if world > 1 and not args.no_group_leg:
        group = group_leg(args, world)
        if default_headline and not args.no_more_configs:
            group5 = group_leg(args, world, "veach", 3840, 2160, 0)

    detail = dict(head, n_gpus=world, steps=args.steps, warmup=args.warmup, yard_sticks=YARD[0], group=group, group_cfg5=group5, configs=more)
    line = contract_line(args, world, head, more, group, group5, detail_file=None)
    line["detail"] = write_detail(detail)
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
