#!/usr/bin/env python3
"""bench_detail.json -> a markdown table of every INPUT of the contract line's roofline numbers, per configuration, with the formulas: a reader
recomputes `frac`, `frac_survey_8d`, `useful_frac`, `frac_hbm_compulsory`, `frac_hbm_counter` and `job_counter_over_compulsory` from the table
alone (committed as profiles/r0N_z_roofline_inputs.md).

  python scratch/roofline_table.py bench_detail.json > profiles/r06_z_roofline_inputs.md
"""
import json
import sys

HBM = 8.0e12
VALU_PEAK = 256*4*2.4e9/2


def one(res, steps):
    rf = res["roofline"]
    rays_per_sample = res["config"]["rays_per_sample"]
    step_s = res["ms_per_step"]*1e-3
    block_s = step_s*steps
    samples = res["value"]*1e6*block_s
    rays = samples*rays_per_sample
    B_ray, B_fb = rf["B_ray"], rf["B_fb"]
    job_bytes = rays*B_ray + samples*B_fb
    comp_bytes = rays*48.0 + samples*B_fb
    dom = rf["kernels"][0]
    out = []
    out.append("### %s" % res["config"]["workload"])
    out.append("")
    out.append("| input | value | where it comes from |")
    out.append("|---|---|---|")
    out.append("| timed block | %d passes, %.4f ms per pass = %.3f ms | `ms_per_step` (median block / K, barrier + synchronize on both sides) |" % (steps, res["ms_per_step"], block_s*1e3))
    out.append("| samples per block | %.4g | W x H x K |" % samples)
    out.append("| rays per sample | %.4f (rays per block %.4g) | device ray counter / samples (`tinsel_hip_stats_detail`) |" % (rays_per_sample, rays))
    out.append("| I, T, P per ray | %.3f, %.3f, %.3f | device counters of one counted pass: internal nodes visited, triangles tested, primitives tested |" % (rf["I"], rf["T"], rf["P"]))
    out.append("| B_ray = 48 + 64 I + 48 T + 84 P | %.1f B | SURVEY.md 8(d) |" % B_ray)
    out.append("| B_fb = 32 K_fp, K_fp = (2 fw + 1)^2 | %.1f B (K_fp %.2f) | SURVEY.md 8(d); fw = the filter's width |" % (B_fb, rf["K_fp"]))
    out.append("| dominant kernel | `%s`: %d launch(es), %.4f ms each | HIP events on the launch stream, last timed block (rocprofv3 --stats of the same command: r06_z_kernel_stats.md) |" % (
        rf["kernel"], rf["launches"], rf["avg_launch_ms"]))
    if rf.get("valu_wave_insts_per_launch"):
        out.append("| its SQ_INSTS_VALU per launch | %.4g wave-instructions | rocprofv3 --pmc pass of this run (r06_z_pmc_configs.md has the same counter from a separate call) |" % rf["valu_wave_insts_per_launch"])
    if rf.get("valu_lanes_active") is not None:
        out.append("| its lanes active | %.4f | SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU) |" % rf["valu_lanes_active"])
    if rf.get("traffic") is not None:
        cal = rf.get("counter_calibration") or {}
        out.append("| its HBM bytes per launch (counters) | %.4g B | FETCH_SIZE x 1024 x %.3f (%s) + WRITE_SIZE x 1024 x %.3f: calibrated in the run on kernels with known byte counts |" % (
            rf["traffic"], (cal.get("gather_bytes_per_fetch_count") if rf["kernel"] in (cal.get("gather_kernels") or []) else cal.get("stream_bytes_per_fetch_count")) or float("nan"),
            "random gathers" if rf["kernel"] in (cal.get("gather_kernels") or []) else "streaming reads: the guide's gfx950 x2", cal.get("bytes_per_write_count") or float("nan")))
    if rf.get("job_counter_GB") is not None:
        out.append("| HBM bytes of ALL kernels of the block (counters) | %.4g B | the same, summed over the kernels |" % (rf["job_counter_GB"]*1e9))
    out.append("")
    out.append("| number in the line | formula | = |")
    out.append("|---|---|---|")
    if rf.get("frac_model") == "valu_issue" and rf.get("valu_wave_insts_per_launch"):
        ach = rf["valu_wave_insts_per_launch"]/(rf["avg_launch_ms"]*1e-3)
        out.append("| `frac` (model `valu_issue`) | SQ_INSTS_VALU per launch / launch seconds / (256 CU x 4 SIMD x 2.4 GHz / 2) | %.4g / %.4g = **%.4f** |" % (ach, VALU_PEAK, ach/VALU_PEAK))
        out.append("| `useful_frac` | frac x lanes active | %.4f x %.4f = **%.4f** |" % (ach/VALU_PEAK, rf["valu_lanes_active"], ach/VALU_PEAK*rf["valu_lanes_active"]))
    else:
        if rf.get("traffic"):
            ach = rf["traffic"]/(rf["avg_launch_ms"]*1e-3)
            out.append("| `frac` = `frac_hbm_counter` (model `hbm_counter`) | counter bytes per launch / launch seconds / 8e12 | %.4g / 8e12 = **%.4f** |" % (ach, ach/HBM))
        if dom.get("valu_frac_of_issue_peak") and dom.get("valu_lanes_active"):
            out.append("| `useful_frac` | VALU issue x lanes active of the dominant kernel | %.4f x %.4f = **%.4f** |" % (dom["valu_frac_of_issue_peak"], dom["valu_lanes_active"], dom["valu_frac_of_issue_peak"]*dom["valu_lanes_active"]))
    out.append("| `frac_survey_8d` | (rays x B_ray + samples x B_fb) / block seconds / 8e12 | %.4g B / %.4g s / 8e12 = **%.4f**%s |" % (
        job_bytes, block_s, job_bytes/block_s/HBM, "  (> 1: the model bills the LDS-resident scene's node / primitive fetches to HBM)" if job_bytes/block_s/HBM > 1 else ""))
    out.append("| `frac_hbm_compulsory` | (rays x 48 + samples x B_fb) / block seconds / 8e12 | %.4g B / %.4g s / 8e12 = **%.4f** |" % (comp_bytes, block_s, comp_bytes/block_s/HBM))
    if rf.get("job_counter_GB") is not None:
        out.append("| `job_counter_over_compulsory` | counter bytes of all kernels / (rays x 48 + samples x B_fb) | %.4g / %.4g = **%.3f** |" % (rf["job_counter_GB"]*1e9, comp_bytes, rf["job_counter_GB"]*1e9/comp_bytes))
    out.append("")
    out.append("per kernel of the block: " + "; ".join("`%s` %d x, %.3f ms, VALU issue %s, lanes %s, counter %s GB" % (
        k["kernel"], k["launches"], k["ms"], ("%.3f" % k["valu_frac_of_issue_peak"]) if k.get("valu_frac_of_issue_peak") else "-",
        ("%.2f" % k["valu_lanes_active"]) if k.get("valu_lanes_active") else "-", ("%.3f" % k["counter_GB"]) if k.get("counter_GB") else "-") for k in rf["kernels"]))
    out.append("")
    return "\n".join(out)


def main():
    d = json.load(open(sys.argv[1]))
    steps = d["steps"]
    print("# The inputs of bench.py's roofline numbers (this run: %d GPU, --steps %d --warmup %d)\n" % (d["n_gpus"], steps, d["warmup"]))
    print("Peaks: HBM 8e12 B/s, VALU issue 256 CU x 4 SIMD x 2.4e9 Hz / 2 cycles = 1.2288e12 wave-instructions/s (/opt/skills/guides/MI355X_MICROARCH.md).\n")
    print(one(d, steps))
    for c in d.get("configs", []):
        if "roofline" in c:
            print(one(c, steps))


if __name__ == "__main__":
    main()
