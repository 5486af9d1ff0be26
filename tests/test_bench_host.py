"""bench.py's host logic that needs no GPU: the self-spawn guard of `--gpus N`, the watchdog around the first collective, and the
mapping from rocprofv3 kernel names to the names the library's timers use."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TINSEL_BENCH_ONE_DEVICE")}


def test_plain_multi_gpu_launch_refuses_loudly_without_the_gpus():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return      # (a real multi-GPU box: nothing to refuse)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "--gpus 2" in p.stderr and "GPU(s) visible" in p.stderr, p.stderr[-1000:]


def test_first_collective_watchdog_prints_a_readable_line():
    code = ("import sys, time; sys.path.insert(0, %r); sys.argv = ['bench.py']; import bench\n"
            "a = bench.parse(); a.gpus = 8\n"
            "bench.first_collective_watchdog(a, 0, 8, 'nccl', seconds=0.2); time.sleep(5); print('NOT REACHED')\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=120)
    assert p.returncode == 4 and "NOT REACHED" not in p.stdout
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["value"] is None and d["n_gpus"] == 8 and "did not complete" in d["unavailable"] and "RCCL" in d["unavailable"]


def test_profiler_kernel_names_map_to_the_timers_names():
    sys.path.insert(0, ROOT)
    import bench
    K = bench._kernel_key
    assert K("void tn::k_walk<1024, 8, 2>(tn::DevScene, tn::WalkJob)") == "k_walk"
    assert K("void tn::k_walk_rays<1024, 8>(tn::DevScene, tn::WalkJob)") == "k_walk"
    assert K("void tn::k_accumulate_tiled<4, 256>(tn::PathState, ...)") == "k_accumulate"
    assert K("tn::k_seg_prefix(unsigned int const*, ...)") == "k_seg" and K("tn::k_seg_expand_all(...)") == "k_seg" and K("tn::k_region_order(...)") == "k_seg"
    assert K("void tn::k_swalk<false, 1024, 1>(...)") == "k_extend" and K("void tn::k_swalk<true, 1024, 1>(...)") == "k_shadow"
    assert K("void tn::k_shade_sorted<true, true>(...)") == "k_shade" and K("void tn::k_shade<true, true>(...)") == "k_shade"
    assert K("void tn::k_bounce<true, true, false>(...)") is None and K("void tn::k_bounce<false, true, false>(...)") == "k_bounce"
    assert K("void tn::k_ub_gather<0>(...)") == "k_ub_gather<0>"


def _sample_detail():
    """a full record as bench.py keeps it in bench_detail.json (round 4's default run: four configurations, per-kernel tables, 28 KB)"""
    with open(os.path.join(ROOT, "tests", "golden", "bench_detail_sample.json")) as f:
        d = json.load(f)
    more = d.pop("configs")
    return d, more


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                 "config", "roofline", "cpu_baseline")


def test_contract_line_is_small_and_complete():
    """the driver reads the line from a bounded window; round 4's 28.5 KB line came back `parsed: null`.  The line is built
    from the full record by bench.contract_line and must stay under 6 KB with every field of the contract in it."""
    sys.path.insert(0, ROOT)
    import bench
    argv, sys.argv = sys.argv, ["bench.py", "--steps", "20", "--warmup", "5"]
    try:
        args = bench.parse()
    finally:
        sys.argv = argv
    head, more = _sample_detail()
    assert len(json.dumps(dict(head, configs=more))) > 20000          # (the input really is the big record)
    line = bench.contract_line(args, 1, head, more, detail_file="bench_detail.json")
    text = json.dumps(line)
    assert len(text) <= bench.LINE_LIMIT_BYTES and "\n" not in text, len(text)
    back = json.loads(text)
    for k in CONTRACT_KEYS:
        assert k in back, k
    assert back["n_gpus"] == 1 and back["steps"] == 20 and back["warmup"] == 5 and back["higher_is_better"] is True and back["vs_baseline"] is None
    assert abs(back["value"] - head["value"]) < 1e-3*head["value"] and back["unit"] == "Msamples/s" and back["dtype"] == "f32"
    assert back["config"]["workload"].startswith("cornell.tin 1024x1024")
    rf = back["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "frac_model"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"]/rf["peak"]) < 1e-3 and rf["kernel"] == "k_bounce" and "kernels" not in rf
    cpu = back["cpu_baseline"]
    assert cpu["kind"] == "reference" and cpu["cores"] == 256 and cpu["value"] > 0 and cpu["sample"]
    # one compact object per OTHER configuration (the headline is not repeated), each naming its BASELINE.json index
    assert [c["baseline_config"] for c in back["configs"]] == [2, 3, 4]
    for c in back["configs"]:
        assert c["value"] > 0 and c["kernel"] and c["frac"] is not None and c["cpu_msamples_s"] > 0 and "roofline" not in c
        assert len(json.dumps(c)) < 700


def test_contract_line_at_eight_ranks_is_small_too():
    sys.path.insert(0, ROOT)
    import bench
    argv, sys.argv = sys.argv, ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"]
    try:
        args = bench.parse()
    finally:
        sys.argv = argv
    head, more = _sample_detail()
    head = dict(head, cpu_baseline=None, pcie_inclusive_msamples_s=None,
                strong={"msamples_s": 20111.5, "ms_per_step": 0.0521, "timed_blocks": 200, "what": "fixed work"},
                ranks={"communicator_world_size": 8, "backend": "nccl",
                       "per_rank": [{"rank": r, "samples": 20971520, "rays": 119000000, "kernel_ms": 36.1 + r, "median_block_ms": 37.9, "device": r} for r in range(8)]})
    veach = dict(more[2], strong={"msamples_s": 15000.0, "ms_per_step": 0.55, "timed_blocks": 30, "what": "fixed work"}, cpu_baseline=None)
    group = {"metric": "x"*200, "n_gpus": 8, "one_device_validation": False, "kpass_msamples_s": 30000.0, "api_1pass_plain_msamples_s": 900.0,
             "api_1pass_lookahead_msamples_s": 2500.0, "api_1pass_lookahead_pinned_output_msamples_s": 3000.0, "calls": 64}
    line = bench.contract_line(args, 8, head, [veach], group, {"unavailable": "timed out after 60 s"}, detail_file="bench_detail.json")
    text = json.dumps(line)
    assert len(text) <= bench.LINE_LIMIT_BYTES, len(text)
    back = json.loads(text)
    assert back["n_gpus"] == 8 and back["scaling"] == "weak" and back["cpu_baseline"] is None
    assert back["strong_msamples_s"] == 20112.0 or abs(back["strong_msamples_s"] - 20111.5) < 1.0
    assert back["ranks"]["communicator_world_size"] == 8 and back["ranks"]["kernel_ms_min_max"] == [36.1, 43.1]
    assert back["configs"][0]["baseline_config"] == 4 and back["configs"][0]["strong_msamples_s"] == 15000.0
    assert back["group"]["kpass_msamples_s"] == 30000.0 and "metric" not in back["group"] and back["group_cfg5"]["unavailable"].startswith("timed out")
