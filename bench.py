#!/usr/bin/env python3
"""bench.py -- Msamples/s and Mrays/s of the hot path on N MI355X (one process per GPU).

A "step" is one pass of the hot path over one batch of synthetic input: one sample per pixel of
BASELINE.json configs[1] -- data/cornell.tin at 1024x1024, maxDepth 4 (spp=256 <=> --steps 256,
the default).  The scene comes from the committed scene pack (tests/golden/cornell.pack, written by
the reference's own loader); camera rays, RNG seeds and everything downstream are generated on the
GPU, so inputs are resident in HBM when the timed region starts.  The accumulation buffer stays in
HBM (the D2H copy of the API's Render() is reported separately as `pcie_inclusive`).

N > 1 (launched by torch.distributed.run): weak scaling -- every rank traces its interleaved
32x32 pixel tiles for K*N passes (same paths per GPU as N = 1), then ONE RCCL sum-reduce of the
float4 accumulator to rank 0 inside the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


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
    return ap.parse_args()


def cpu_baseline(scene_name, cam, opt, target_core_seconds):
    """The reference's own PathTrace (oracle/_ref, compiled unmodified with the reference's
    -O3 -ffast-math flags) on this box's host cores, on a bounded sample of the same workload."""
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


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world

    import torch
    import torch.distributed as dist

    import tinsel_amd
    from tinsel_amd import abi

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

    pack = os.path.join(ROOT, "tests", "golden", args.scene + ".pack")
    scene = tinsel_amd.Scene.load_pack(pack)
    cam = scene.camera
    opt = scene.options.copy()
    opt.width, opt.height = args.width, args.height
    if args.maxdepth > 0:
        opt.max_depth = args.maxdepth
    elif args.scene == "glass":
        opt.max_depth = 12      # BASELINE.json configs[3]
    opt.mode = abi.MODE_PATHTRACE

    r = tinsel_amd.create_gpu_renderer(scene, local)
    bvh_build_ms = r.set_mesh_bvh(abi.BVH_LBVH) if args.bvh == "lbvh" else None
    if args.roulette > 0:
        r.set_russian_roulette(args.roulette)
    r.set_pipeline({"auto": abi.PIPELINE_AUTO, "wavefront": abi.PIPELINE_WAVEFRONT, "mega": abi.PIPELINE_MEGAKERNEL, "split": abi.PIPELINE_WAVEFRONT_SPLIT}[args.pipeline])
    if world > 1:
        r.set_shard(rank, world, args.tile)
        r.set_batch_paths((8 << 20)*world)       # keep the same number of LIVE paths per batch as N = 1
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
    K_fp = (2*int(fw) + 1)**2 if opt.filter.type == abi.FILTER_GAUSSIAN else (2*int(fw) + 1)**2
    B_fb = 32.0*K_fp

    # ---- warmup --------------------------------------------------------------------------------
    if args.warmup > 0:
        run(args.warmup)
    sync()

    # ---- timed region ----------------------------------------------------------------------------
    accum.zero_()
    r.reset_stats()
    r.enable_kernel_timing(True)
    sync()
    first_timed_pass = r.get_pass_index()
    t0 = time.perf_counter()
    run(args.steps)
    sync()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    st = r.stats()
    ktimes = r.kernel_times()
    r.enable_kernel_timing(False)

    # max over ranks of the elapsed time; sums of the counters
    if world > 1:
        small_dev = "cuda" if backend == "nccl" else "cpu"
        tt = torch.tensor([elapsed], dtype=torch.float64, device=small_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        cc = torch.tensor([st["rays"], st["samples"], st["shadow_rays"]], dtype=torch.float64, device=small_dev)
        dist.all_reduce(cc, op=dist.ReduceOp.SUM)
        tot_rays, tot_samples, tot_shadow = (float(x) for x in cc.tolist())
    else:
        tot_rays, tot_samples, tot_shadow = float(st["rays"]), float(st["samples"]), float(st["shadow_rays"])

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- PCIe-inclusive variant of the API call (D2H of the accumulator), N = 1 only --------------
    pcie = None
    if world == 1:
        out = np.empty((opt.height, opt.width, 4), np.float32)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        r.render(cam, opt, output=out, passes=16)
        t3 = time.perf_counter()
        pcie = 16*opt.width*opt.height/(t3 - t2)/1e6

    # ---- roofline of the dominant kernel ------------------------------------------------------------
    dom = max(ktimes.items(), key=lambda kv: kv[1][1]) if ktimes else (None, (0, 0.0))
    dom_name, (dom_launches, dom_ms) = dom
    rays_by_kernel = {"k_extend": st["rays"] - st["shadow_rays"], "k_shadow": st["shadow_rays"], "k_mega": st["rays"], "k_bounce": st["rays"]}
    if "k_shade" in ktimes and dom_name == "k_shade":
        # k_shade only moves path state (no algorithmic bytes by SURVEY 8(d)): rate the heaviest TRACE kernel instead
        dom_name, (dom_launches, dom_ms) = max(((k, v) for k, v in ktimes.items() if k in ("k_extend", "k_shadow")), key=lambda kv: kv[1][1])
    if dom_name in rays_by_kernel:
        dom_bytes = rays_by_kernel[dom_name]*B_ray
    elif dom_name == "k_accumulate":
        dom_bytes = st["samples"]*B_fb
    else:
        dom_bytes = 0.0     # k_shade/k_generate move path state only: no algorithmic bytes by SURVEY 8(d)'s definition
    achieved = dom_bytes/(dom_ms*1e-3)/1e9 if dom_ms > 0 else 0.0
    gpu_ms = sum(v[1] for v in ktimes.values())
    job_bytes = st["rays"]*B_ray + st["samples"]*B_fb

    # HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/
    # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; see the file's own note), same workload
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_traffic_%s.json" % args.scene)
    if os.path.exists(tpath) and (args.width, args.height) == (1024, 1024):
        try:
            traffic = json.load(open(tpath)).get(dom_name, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    roofline = {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved/HBM_PEAK_GBS,
        "traffic": traffic,
        "kernel": dom_name, "launches": dom_launches, "avg_launch_ms": dom_ms/max(1, dom_launches),
        "B_ray": B_ray, "I": I_bar, "T": T_bar, "P": P_bar, "B_fb": B_fb,
        "job_algorithmic_GBs": job_bytes/(gpu_ms*1e-3)/1e9 if gpu_ms > 0 else 0.0,
        "kernel_ms": {k: round(v[1], 3) for k, v in ktimes.items()},
    }

    # validation mode only: the reduced image of the N-rank run must equal an unsharded render of the same passes
    if os.environ.get("TINSEL_BENCH_ONE_DEVICE") and world > 1:
        torch.cuda.synchronize()
        chk = tinsel_amd.create_gpu_renderer(scene, local)
        chk.init(opt.width, opt.height)
        chk.set_pass_index(first_timed_pass)
        want = chk.render(cam, opt, passes=args.steps*passes_per_step)
        chk.close()
        got = accum.cpu().numpy()
        ok = np.allclose(got, want, rtol=1e-4, atol=1e-5)
        print("validation: %d-rank reduced image vs unsharded render of passes [%d, %d): %s (max abs diff %.3e)" % (
            world, first_timed_pass, first_timed_pass + args.steps*passes_per_step, "ok" if ok else "MISMATCH",
            float(np.abs(got - want).max())), file=sys.stderr, flush=True)
        if not ok:
            raise SystemExit(3)

    cpu = None
    if not args.no_cpu_baseline and world == 1:        # the CPU leg is timed at N = 1 only
        try:
            cpu = cpu_baseline(args.scene, cam, opt, args.cpu_seconds)
        except Exception as e:      # a checker built for another box must not kill the bench line
            cpu = {"value": None, "unit": "Msamples/s", "cores": 0, "kind": "reference", "sample": "unavailable: %s" % e}

    msamples = tot_samples/elapsed/1e6
    line = {
        "metric": "Msamples/s (%s.tin %dx%d maxDepth=%d spp=%d, wavefront path; Mrays/s alongside)" % (
            args.scene, opt.width, opt.height, opt.max_depth, args.steps*passes_per_step),
        "value": msamples, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed*1e3/args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s.tin %dx%d maxDepth=%d, %d pass(es) per step, pipeline=%s" % (
            args.scene, opt.width, opt.height, opt.max_depth, passes_per_step,
            args.pipeline if args.pipeline != "auto" else "auto->" + ("wavefront(fused)" if "k_bounce" in ktimes else "wavefront(split)")),
            "scene_pack": os.path.relpath(pack, ROOT), "parallelism": "pixel-tile shard x%d + RCCL reduce" % world if world > 1 else "1 GPU",
            "filter": "gaussian w=%.2f" % fw, "rays_per_sample": tot_rays/max(1.0, tot_samples),
            "mesh_bvh": args.bvh, "mesh_bvh_build_ms": bvh_build_ms, "russian_roulette_from_bounce": args.roulette},
        "mrays_per_s": tot_rays/elapsed/1e6,
        "shadow_ray_fraction": tot_shadow/max(1.0, tot_rays),
        "gpu_kernel_ms_total": gpu_ms,
        "pcie_inclusive_msamples_s": pcie,
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
