"""The N > 1 path on CPU: world_size-2 `gloo` processes, each rendering its pixel-tile shard with the
(CPU) oracle into a full-size accumulator, then tinsel_amd.distributed.reduce_accum -- the same host
code the GPU ranks run with the `nccl` (RCCL) backend.  The reduced frame must equal the unsharded one."""
import os
import sys

import numpy as np
import pytest

from tinsel_amd import abi, distributed
from tests import oracle_api as oa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,tile", [(2, 8), (3, 16), (8, 32)])
def test_owned_masks_partition_the_frame(world, tile):
    W, H = 100, 70
    total = np.zeros((H, W), int)
    for r in range(world):
        total += distributed.owned_mask(W, H, r, world, tile)
    assert np.all(total == 1)


def _worker(rank, world, port, tmp):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = oa.PortOracle()
        g = np.load(os.path.join(oa.GOLDEN, "cornell.golden.npz"))
        cam = abi.Camera.from_buffer_copy(g["camera"].tobytes())
        opt = abi.Options.from_buffer_copy(g["options"].tobytes())
        h = P.load_pack(os.path.join(oa.GOLDEN, "cornell.pack"))
        acc, nsamples = P.render_sharded(h, cam, opt, rank, world, tile=8, passes=2, threads=2)
        mask = distributed.owned_mask(opt.width, opt.height, rank, world, 8)
        assert nsamples == 2*int(mask.sum())          # the oracle's shard rule == the host's owned_mask
        t = torch.from_numpy(acc)
        mine = t.clone()
        total = distributed.reduce_accum(t, dst=0)
        assert torch.equal(t, mine)                   # the rank's own accumulator is untouched by the reduce ...
        again = distributed.reduce_accum(t, dst=0)    # ... so reducing twice gives the same frame (no double counting)
        scratch = torch.full_like(t, 7.0)             # a caller-kept target (bench.py keeps one outside its timed region)
        third = distributed.reduce_accum(t, dst=0, out=scratch)
        assert torch.equal(t, mine)
        if rank == 0:
            assert torch.equal(total, again)
            assert third is scratch and torch.equal(third, total)
            whole, _, _ = P.render_seeded(h, cam, opt, 0, 2, threads=2)
            np.save(os.path.join(tmp, "reduced.npy"), total.numpy())
            np.save(os.path.join(tmp, "whole.npy"), whole)
        else:
            assert total is None and again is None and third is None
        P.free(h)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gloo_world2_reduce_equals_whole(tmp_path):
    import torch.multiprocessing as mp
    if not oa.have_port():
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "port"], check=True)
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    reduced = np.load(tmp_path / "reduced.npy")
    whole = np.load(tmp_path / "whole.npy")
    # identical up to float summation order (each pixel sums the same terms, grouped by rank)
    np.testing.assert_allclose(reduced, whole, rtol=2e-6, atol=1e-6)
    assert oa.image_l2(reduced, whole) < 1e-6


def test_reduce_accum_fills_out_with_one_rank():
    # a caller that reads its own `out` buffer must find the sum there for ANY number of ranks, one included
    import torch
    acc = torch.arange(24, dtype=torch.float32).reshape(2, 3, 4)
    out = torch.full_like(acc, -1.0)
    got = distributed.reduce_accum(acc, dst=0, out=out)
    assert got is out and torch.equal(out, acc)
    assert distributed.reduce_accum(acc, dst=0) is acc
