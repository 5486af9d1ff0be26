#!/usr/bin/env python3
"""bench.py -- Msamples/s and Mrays/s of the hot path on N MI355X (one process per GPU).

A "step" is one pass of the hot path over one batch of synthetic input: one sample per pixel.  The default run (N = 1)
times TWO configurations of BASELINE.json and prints ONE JSON line:

  configs[0] / headline   BASELINE configs[1]: data/cornell.tin 1024x1024 maxDepth 4 (spp 256 <=> --steps 256).  The
                          scene (2.6 KB) lives in LDS: the path is bound by VALU issue, and `roofline` says so
                          (bound "valu": wave-instructions issued per second against SIMDs x clock / 2).
  configs[1]              BASELINE configs[2]: data/ajax.tin with the 524,288-triangle stand-in mesh (ajax.obj is not in
                          the reference tree) at 1920x1080 maxDepth 4 -- the configuration whose scene lives in HBM / the
                          Infinity Cache.  Its roofline carries BOTH HBM fractions: algorithmic bytes (SURVEY.md 8d's
                          B_ray model) and counter bytes, each divided by time and by 8 TB/s.

Scenes come from scene packs written by the reference's own loader (tests/golden/*.pack); camera rays, RNG seeds and
everything downstream are generated on the GPU, so inputs are resident in HBM when a timed region starts, and the
accumulation buffer stays in HBM (`pcie_inclusive_*` and `api_1pass_*` report the API's per-call D2H separately).

Timing: W untimed warm-up steps, then blocks of EXACTLY K steps, each bracketed by barrier + torch.cuda.synchronize()
on both sides, repeated until >= 0.5 s have been timed; `ms_per_step` is the MEDIAN block / K, `value` the samples of
one block / that median (max over ranks for N > 1).

Counters: `roofline.traffic` (HBM bytes per launch of the dominant kernel) and the VALU instruction count behind
`roofline.achieved` are measured IN THIS RUN, on this box, by re-running the same workload under
`rocprofv3 --pmc` (separate passes for FETCH_SIZE, WRITE_SIZE and the SQ counters, the gfx950 x2 on FETCH_SIZE of
MI355X_MICROARCH.md); when rocprofv3 is not available the fields are null -- nothing is read from a stored file.

N > 1 (launched by torch.distributed.run): weak scaling -- every rank traces its interleaved pixel tiles for K*N passes
(same paths per GPU as N = 1), then ONE RCCL sum-reduce of the float4 accumulator to rank 0 inside the timed region.
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
LARGE = "large/ajax_standin"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--scene", default="cornell")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--maxdepth", type=int, default=0, help="0 = the scene's own / BASELINE value")
    ap.add_argument("--pipeline", choices=["auto", "wavefront", "mega", "split"], default="auto")
    ap.add_argument("--bvh", choices=["reference", "lbvh"], default="reference",
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
}


def _kernel_key(name):
    m = re.search(r"(k_\w+)(<[^>]*>)?", name)
    if not m:
        return None
    if m.group(2) and m.group(2).startswith("<true"):
        return None                         # detail-counting variants (COUNT = true) are not the product kernels
    return "k_accumulate" if m.group(1).startswith("k_accumulate") else m.group(1)


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
    if args.bvh == "lbvh":
        r.set_mesh_bvh(abi.BVH_LBVH)
    if args.roulette > 0:
        r.set_russian_roulette(args.roulette)
    r.set_pipeline({"auto": abi.PIPELINE_AUTO, "wavefront": abi.PIPELINE_WAVEFRONT, "mega": abi.PIPELINE_MEGAKERNEL, "split": abi.PIPELINE_WAVEFRONT_SPLIT}[args.pipeline])
    r.init(opt.width, opt.height)
    r.reserve(args.steps, opt.max_depth)
    r.render(cam, opt, passes=args.steps, readback=False)
    r.close()


# ---------------------------------------------------------------------------------------------------------------------

def run_config(args, scene_name, width, height, maxdepth, rank, world, local, dist, backend, torch, with_extras):
    """Times one configuration per the contract; returns (dict for the JSON line, elapsed seconds of the median block)."""
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
    bvh_build_ms = r.set_mesh_bvh(abi.BVH_LBVH) if args.bvh == "lbvh" else None
    if args.roulette > 0:
        r.set_russian_roulette(args.roulette)
    r.set_pipeline({"auto": abi.PIPELINE_AUTO, "wavefront": abi.PIPELINE_WAVEFRONT, "mega": abi.PIPELINE_MEGAKERNEL, "split": abi.PIPELINE_WAVEFRONT_SPLIT}[args.pipeline])
    if args.arith == "fast":
        r.set_arithmetic(abi.ARITH_FAST)
    if world > 1:
        r.set_shard(rank, world, args.tile)     # path slots are rank-local: the same batch size as N = 1 holds the same number of live paths
    accum = torch.zeros((opt.height, opt.width, 4), dtype=torch.float32, device="cuda")
    r.init(opt.width, opt.height, accum_tensor=accum)
    stream = torch.cuda.current_stream().cuda_stream
    r.reserve(max(args.steps, args.warmup, 1)*world, opt.max_depth)      # no hipMalloc inside the timed region

    passes_per_step = world         # weak scaling: K*N passes over 1/N of the pixels each

    def run(steps):
        r.render_async(cam, opt, passes=steps*passes_per_step, stream=stream)
        if world > 1:
            if backend == "nccl":
                dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)         # RCCL over xGMI, on the render stream
            else:
                torch.cuda.synchronize()
                host = accum.cpu()
                dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
                if rank == 0:
                    accum.copy_(host)

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
    rays_c = max(1, c["rays"])
    I_bar, T_bar, P_bar = c["internal_visits"]/rays_c, c["tri_tests"]/rays_c, c["prim_tests"]/rays_c
    B_ray = 48.0 + 64.0*I_bar + 48.0*T_bar + 84.0*P_bar
    fw = opt.filter.width
    K_fp = (2*int(fw) + 1)**2
    B_fb = 32.0*K_fp

    # ---- warmup --------------------------------------------------------------------------------
    if args.warmup > 0:
        run(args.warmup)
    sync()

    # ---- timed blocks of exactly K steps, repeated until MIN_TIMED_S ----------------------------
    first_timed_pass = None
    blocks, stats_blocks, ktimes = [], None, {}
    total = 0.0
    r.enable_kernel_timing(True)
    while True:
        accum.zero_()
        r.reset_stats()
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
        ktimes = r.kernel_times()               # of this block's render call
        if total >= MIN_TIMED_S or len(blocks) >= 1000:
            break
    r.enable_kernel_timing(False)
    elapsed = statistics.median(blocks)
    st = stats_blocks

    if world > 1:
        small_dev = "cuda" if backend == "nccl" else "cpu"
        cc = torch.tensor([st["rays"], st["samples"], st["shadow_rays"]], dtype=torch.float64, device=small_dev)
        dist.all_reduce(cc, op=dist.ReduceOp.SUM)
        tot_rays, tot_samples, tot_shadow = (float(x) for x in cc.tolist())
    else:
        tot_rays, tot_samples, tot_shadow = float(st["rays"]), float(st["samples"]), float(st["shadow_rays"])

    if rank != 0:
        r.close()
        return None

    # validation mode only: the reduced image of the N-rank run must equal an unsharded render of the same passes
    if os.environ.get("TINSEL_BENCH_ONE_DEVICE") and world > 1:
        torch.cuda.synchronize()
        last_first = r.get_pass_index() - args.steps*passes_per_step
        chk = tinsel_amd.create_gpu_renderer(scene, local)
        chk.init(opt.width, opt.height)
        chk.set_pass_index(last_first)
        want = chk.render(cam, opt, passes=args.steps*passes_per_step)
        chk.close()
        got = accum.cpu().numpy()
        ok = np.allclose(got, want, rtol=1e-4, atol=1e-5)
        print("validation: %d-rank reduced image vs unsharded render of passes [%d, %d): %s (max abs diff %.3e)" % (
            world, last_first, last_first + args.steps*passes_per_step, "ok" if ok else "MISMATCH",
            float(np.abs(got - want).max())), file=sys.stderr, flush=True)
        if not ok:
            raise SystemExit(3)

    # ---- the API's own call pattern, N = 1 only -------------------------------------------------
    pcie = api_1pass = api_1pass_plain = None
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
        ra.set_lookahead(True)
        ra.render(cam, opt, output=out, passes=1)               # pins the host array, starts the pipeline
        t2 = time.perf_counter()
        for _ in range(calls):
            ra.render(cam, opt, output=out, passes=1)
        t3 = time.perf_counter()
        api_1pass = calls*opt.width*opt.height/(t3 - t2)/1e6
        ra.close()

    # ---- the opt-in tolerance-arithmetic arm (tinsel_hip_set_arithmetic): rate and distance, N = 1 only ---------
    fast = None
    if world == 1 and not args.no_fast and args.arith == "exact":
        try:
            spp = 256
            rf = tinsel_amd.create_gpu_renderer(scene, local)
            rf.set_pipeline({"auto": abi.PIPELINE_AUTO, "wavefront": abi.PIPELINE_WAVEFRONT, "mega": abi.PIPELINE_MEGAKERNEL, "split": abi.PIPELINE_WAVEFRONT_SPLIT}[args.pipeline])
            rf.init(opt.width, opt.height)
            rf.reserve(max(spp, args.steps), opt.max_depth)
            exact_img = rf.render(cam, opt, passes=spp)
            rf.set_arithmetic(abi.ARITH_FAST)
            rf.init(opt.width, opt.height)
            rf.set_pass_index(0)
            fast_img = rf.render(cam, opt, passes=spp)
            wa = np.where(exact_img[..., 3:4] > 0, exact_img[..., 3:4], 1.0)
            wb = np.where(fast_img[..., 3:4] > 0, fast_img[..., 3:4], 1.0)
            dd = (exact_img[..., :3]/wa - fast_img[..., :3]/wb).astype(np.float64)
            l2 = float(np.sqrt(np.mean(np.sum(dd*dd, axis=-1))))
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
            fast = {"msamples_s": args.steps*opt.width*opt.height/statistics.median(fblocks)/1e6, "l2_vs_exact_at_spp": [l2, spp],
                    "divergent_paths_fraction": float((rel > 1e-3).mean()), "identical_paths_fraction": float((rad_f == rad_e).all(axis=-1).mean())}
        except Exception as e:
            fast = {"msamples_s": None, "error": str(e)}

    # ---- roofline of the dominant kernel ----------------------------------------------------------
    gpu_ms = sum(v[1] for v in ktimes.values())
    trace_kernels = ("k_walk", "k_extend", "k_shadow", "k_mega", "k_bounce")
    dom = max(((k, v) for k, v in ktimes.items() if k in trace_kernels), key=lambda kv: kv[1][1], default=(None, (0, 0.0)))
    dom_name, (dom_launches, dom_ms) = dom
    avg_launch_s = dom_ms*1e-3/max(1, dom_launches)
    rays = st["rays"]
    # algorithmic bytes by kernel (SURVEY.md 8d's B_ray split over the kernels that do the work):
    #   the mesh walk (Node64 visits + triangle tests) belongs to k_walk when it runs, the primitive tests and ray/hit records to the scan kernels
    alg = {"k_bounce": rays*B_ray, "k_mega": rays*B_ray}
    if "k_walk" in ktimes:
        alg["k_walk"] = rays*(64.0*I_bar + 48.0*T_bar)
        alg["k_extend"] = (rays - st["shadow_rays"])*(48.0 + 84.0*P_bar)
        alg["k_shadow"] = st["shadow_rays"]*(48.0 + 84.0*P_bar)
    else:
        alg["k_extend"] = (rays - st["shadow_rays"])*B_ray
        alg["k_shadow"] = st["shadow_rays"]*B_ray
    dom_bytes = alg.get(dom_name, 0.0)
    alg_gbs = dom_bytes/(dom_ms*1e-3)/1e9 if dom_ms > 0 else 0.0
    job_bytes = rays*B_ray + st["samples"]*B_fb

    pmc = {"sq": None, "fetch": None, "write": None}
    if world == 1 and not args.no_pmc and dom_name:
        for key in ("sq", "fetch", "write"):
            pmc[key] = pmc_pass(args, scene_name, width, height, opt.max_depth, args.steps, PMC_SETS[key])
    traffic = valu_per_launch = lanes = wait = None
    if pmc["fetch"] and pmc["write"] and dom_name in pmc["fetch"] and dom_name in pmc["write"]:
        f, w = pmc["fetch"][dom_name], pmc["write"][dom_name]
        # KB units; x2 on FETCH_SIZE: gfx950 tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section)
        traffic = (2.0*f["FETCH_SIZE"]/f["launches"] + w["WRITE_SIZE"]/w["launches"])*1024.0
    if pmc["sq"] and dom_name in pmc["sq"]:
        q = pmc["sq"][dom_name]
        valu_per_launch = q.get("SQ_INSTS_VALU", 0.0)/q["launches"]
        if q.get("SQ_INSTS_VALU"):
            lanes = q.get("SQ_THREAD_CYCLES_VALU", 0.0)/(64.0*q["SQ_INSTS_VALU"])
        if q.get("SQ_WAVE_CYCLES"):
            wait = q.get("SQ_WAIT_ANY", 0.0)/q["SQ_WAVE_CYCLES"]

    scene_in_lds = "k_bounce" in ktimes or "k_mega" in ktimes       # the fused arms run only when the whole scene is LDS-resident
    common = {
        "kernel": dom_name, "launches": dom_launches, "avg_launch_ms": avg_launch_s*1e3, "traffic": traffic,
        "algorithmic_GBs": alg_gbs, "frac_hbm_algorithmic": alg_gbs/HBM_PEAK_GBS,
        "counter_GBs": (traffic/avg_launch_s/1e9) if (traffic and avg_launch_s > 0) else None,
        "frac_hbm_counter": (traffic/avg_launch_s/1e9/HBM_PEAK_GBS) if (traffic and avg_launch_s > 0) else None,
        "valu_wave_insts_per_launch": valu_per_launch, "valu_lanes_active": lanes, "wave_cycles_waiting": wait,
        "B_ray": B_ray, "I": I_bar, "T": T_bar, "P": P_bar, "B_fb": B_fb,
        "job_algorithmic_GBs": job_bytes/(gpu_ms*1e-3)/1e9 if gpu_ms > 0 else 0.0,
        "kernel_ms": {k: round(v[1], 3) for k, v in ktimes.items()},
        "counters": "rocprofv3 --pmc passes of this run (same workload, same passes per launch)" if any(pmc.values()) else None,
    }
    if scene_in_lds:
        # the scene never leaves the CU: the HBM model counts bytes that are LDS reads.  What binds is instruction issue.
        ach = (valu_per_launch/avg_launch_s/1e9) if (valu_per_launch and avg_launch_s > 0) else None
        roofline = dict({"bound": "valu", "achieved": ach, "peak": VALU_PEAK/1e9, "unit": "G wave-instructions/s",
                         "frac": (ach/(VALU_PEAK/1e9)) if ach else None}, **common)
    else:
        # counter bytes when this run measured them, else the algorithmic figure (never a stored constant)
        ach = common["counter_GBs"] if common["counter_GBs"] else alg_gbs
        roofline = dict({"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach/HBM_PEAK_GBS,
                         "achieved_is": "counter bytes" if common["counter_GBs"] else "algorithmic bytes"}, **common)

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
            args.pipeline if args.pipeline != "auto" else "auto->" + ("wavefront(fused)" if "k_bounce" in ktimes else "wavefront(split%s)" % ("+k_walk" if "k_walk" in ktimes else ""))),
            "scene_pack": os.path.relpath(pack, ROOT), "parallelism": "pixel-tile shard x%d + RCCL reduce" % world if world > 1 else "1 GPU",
            "filter": "gaussian w=%.2f" % fw, "rays_per_sample": tot_rays/max(1.0, tot_samples),
            "mesh_bvh": args.bvh, "mesh_bvh_build_ms": bvh_build_ms, "russian_roulette_from_bounce": args.roulette},
        "mrays_per_s": tot_rays/elapsed/1e6,
        "shadow_ray_fraction": tot_shadow/max(1.0, tot_rays),
        "gpu_kernel_ms_total": gpu_ms,
        "pcie_inclusive_msamples_s": pcie,
        "api_1pass_msamples_s": api_1pass,
        "api_1pass_plain_msamples_s": api_1pass_plain,
        "arithmetic": args.arith,
        "fast_msamples_s": fast["msamples_s"] if fast else None,
        "fast_l2": fast["l2_vs_exact_at_spp"][0] if (fast and fast.get("l2_vs_exact_at_spp")) else None,
        "fast": fast,
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    r.close()
    return res


def main():
    args = parse()
    if args.inner_pmc:
        return inner_pmc(args)
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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    head = run_config(args, args.scene, args.width, args.height, args.maxdepth, rank, world, local, dist, backend, torch, with_extras=True)

    second = None
    default_headline = (args.scene, args.width, args.height) == ("cornell", 1024, 1024)
    if world == 1 and default_headline and not args.no_second_config:
        if os.path.exists(os.path.join(ROOT, "tests", "golden", LARGE + ".pack")):
            try:
                second = run_config(args, LARGE, 1920, 1080, 4, rank, world, local, dist, backend, torch, with_extras=False)
            except Exception as e:
                second = {"config": {"workload": "ajax stand-in (524,288 triangles) 1920x1080 maxDepth=4"}, "unavailable": "failed: %s" % e}
        else:
            second = {"config": {"workload": "ajax stand-in (524,288 triangles) 1920x1080 maxDepth=4"},
                      "unavailable": "tests/golden/large/ajax_standin.pack is not on this box (72 MB, git-ignored; written by tests/golden/make_large.py "
                                     "from the reference's data/ajax.tin where /root/reference is mounted)"}

    if rank == 0:
        line = {
            "metric": head["metric"], "value": head["value"], "unit": head["unit"], "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
        }
        for k, v in head.items():
            if k not in line:
                line[k] = v
        if second is not None:
            line["configs"] = [{k: head[k] for k in ("metric", "value", "unit", "ms_per_step", "mrays_per_s", "config", "roofline", "cpu_baseline", "timed_blocks", "fast_msamples_s", "fast_l2")}, second]
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
