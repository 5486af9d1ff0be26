"""bench.py's N-rank code path, end to end, on ONE GPU: `python -m torch.distributed.run --nproc-per-node 2 bench.py
--gpus 2` exactly as the driver launches it, with the two ranks sharing device 0 and `gloo` standing in for RCCL
(TINSEL_BENCH_ONE_DEVICE / TINSEL_BENCH_BACKEND: validation switches, not a measurement).  Checks the launch, the
pixel-tile shard, the reduce (weak and fixed-work legs), the small one-line JSON contract and -- inside bench.py -- that the reduced image equals an
unsharded render of the same passes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_device():
    env = dict(os.environ, TINSEL_BENCH_BACKEND="gloo", TINSEL_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "8", "--warmup", "1", "--width", "512", "--height", "384"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "validation: 2-rank reduced image vs unsharded render" in p.stderr and ": ok" in p.stderr, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout            # rank 0 prints ONE json line
    assert len(lines[0]) <= 6144                # ... a small one (the driver reads it from a bounded window)
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 8 and d["scaling"] == "weak" and d["unit"] == "Msamples/s"
    # weak scaling: 2 ranks x (8 steps x 2 passes) over half of the pixels each = 16 full-frame passes of samples
    assert abs(d["value"]*d["ms_per_step"]*1e-3*8*1e6 - 16*512*384) < 1e-3*16*512*384
    assert d["roofline"]["kernel"] and d["cpu_baseline"] is None
    # the fixed-work leg beside it: the same 8 full-frame passes split over the two ranks' tiles + the reduce
    assert d["strong_msamples_s"] > 0 and abs(d["strong_msamples_s"]*d["strong_ms_per_step"]*1e-3*8*1e6 - 8*512*384) < 1e-3*8*512*384
    # AFTER the ranks' timed regions rank 0 timed the SAME devices through the library's own multi-GPU path (tinsel_hip_group, one-device validation here)
    g = d["group"]
    assert g["one_device_validation"] is True and g["kpass_msamples_s"] > 0 and g["api_1pass_plain_msamples_s"] > 0 and g["api_1pass_lookahead_msamples_s"] > 0, g
    assert d["ranks"]["communicator_world_size"] == 2 and d["ranks"]["kernel_ms_min_max"][0] > 0
    # the fixed-work rate as a top-level key beside `value`; the gloo stand-in says that it is not the library's RCCL arm
    assert d["value_fixed_work"] == d["strong_msamples_s"] and d["rccl_ranks_seen"] is None and "gloo" in d["reduce_impl"]
    # the full record (stderr, behind a prefix; also bench_detail.json): every rank's share of the work
    full = json.loads([l for l in p.stderr.splitlines() if l.startswith("bench_detail: ")][-1][len("bench_detail: "):])
    rk = full["ranks"]
    assert [x["rank"] for x in rk["per_rank"]] == [0, 1]
    assert sum(x["samples"] for x in rk["per_rank"]) == 16*512*384 and all(x["kernel_ms"] > 0 for x in rk["per_rank"])
    assert full["group"]["n_gpus"] == 2 and full["strong"]["timed_blocks"] >= 1


def test_plain_launch_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with NO launcher in the environment (the way the driver starts `--gpus 1`): bench.py re-executes itself
    under torch.distributed.run, so the line still says n_gpus = 2 (a first 8-GPU lease must not be spent on one rank)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TINSEL_BENCH_BACKEND="gloo", TINSEL_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--width", "256", "--height", "256",
                        "--no-group-leg"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "starting 2 ranks" in p.stderr
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"]["communicator_world_size"] == 2 and d["value"] > 0 and d["strong_msamples_s"] > 0 and "group" not in d


def test_group_mode_prints_one_line():
    """`bench.py --group --gpus 2`: tinsel_hip_group in one process (all members on device 0 on this pool's boxes)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--group", "--gpus", "2", "--steps", "4", "--warmup", "1", "--width", "256", "--height", "256"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    g = json.loads(lines[0])
    assert g["n_gpus"] == 2 and g["kpass_passes_per_call"] == 8 and g["api_1pass_lookahead_pinned_output_msamples_s"] > 0


def test_library_reduce_arm_with_one_rank():
    """The process-per-GPU arm of the ONE collective (tinsel_hip_comm_*: the library's own ncclReduce, the code a tinsel_hip_group's threads
    run too) executed end to end with the only world size a one-GPU box allows: `bench.py --force-comm` makes the communicator from a unique
    id (ncclCommInitRank), puts the reduce inside the timed region and checks the reduced image against the accumulator.  The line says which
    implementation reduced and how many ranks RCCL ITSELF counted (ncclCommCount)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-comm", "--steps", "4", "--warmup", "1", "--width", "256", "--height", "256",
                        "--no-pmc", "--no-fast", "--no-api", "--no-ubench", "--no-cpu-baseline", "--no-second-config", "--no-more-configs"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "validation: 1-rank library ncclReduce of the accumulator: ok" in p.stderr, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["rccl_ranks_seen"] == 1 and d["reduce_impl"].startswith("library ncclReduce"), d
    assert d["value"] > 0 and d["value_fixed_work"] == d["value"]


def test_comm_entry_points_refuse_what_they_must():
    import tinsel_amd
    from tests.test_gpu_parity import _load
    scene, cam, opt, g = _load("cornell")
    r = tinsel_amd.create_gpu_renderer(scene)
    assert r.comm_size() == 0
    with pytest.raises(tinsel_amd.TinselHipError, match="no communicator"):
        r.comm_reduce_accum(None, 0, None)
    ident = r.comm_unique_id()
    assert len(ident) == 128 and any(ident)
    with pytest.raises(tinsel_amd.TinselHipError, match="bad arguments"):
        r.comm_init(ident, 2, 2)
    r.comm_init(ident, 0, 1)
    assert r.comm_size() == 1
    with pytest.raises(tinsel_amd.TinselHipError, match="accumulator"):
        r.comm_reduce_accum(None, 0, None)          # no init() yet
    r.close()
