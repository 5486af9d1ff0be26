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
    assert K("void tn::k_accumulate_tiled<4, 256>(tn::PathState, ...)") == "k_accumulate"
    assert K("tn::k_seg_prefix(unsigned int const*, ...)") == "k_seg" and K("tn::k_seg_expand_all(...)") == "k_seg" and K("tn::k_region_order(...)") == "k_seg"
    assert K("void tn::k_swalk<false, 1024, 1>(...)") == "k_extend" and K("void tn::k_swalk<true, 1024, 1>(...)") == "k_shadow"
    assert K("void tn::k_shade_sorted<true, true>(...)") == "k_shade" and K("void tn::k_shade<true, true>(...)") == "k_shade"
    assert K("void tn::k_bounce<true, true, false>(...)") is None and K("void tn::k_bounce<false, true, false>(...)") == "k_bounce"
    assert K("void tn::k_ub_gather<0>(...)") == "k_ub_gather<0>"
