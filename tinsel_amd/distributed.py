"""Pixel-tile sharding of one frame over the GPUs of a node + the single reduce that re-assembles it.

The reference has no multi-device path.  Here the path shards
naturally: paths are independent, seeds depend on (pixel, pass) only, and the only shared state is
the accumulation buffer, a commutative float sum (render.cpp:439).  So

  * rank r traces the paths whose GENERATING pixel lies in a tile t with  t % world == r
    (tiles of `tile` x `tile` pixels in raster order; the same rule lives in the kernels:
    tn_kernels.h `pixel_owned`),
  * every rank keeps a FULL-size float4 accumulator (a splat may land outside the owning tile),
  * ONE `reduce(SUM)` to rank 0 per read-back re-assembles the frame (RCCL over xGMI with the
    "nccl" backend; "gloo" in the CPU tests).  4K: 132.7 MB, once per thousands of passes.

One process per GPU; launched by torch.distributed.run (bench.py --gpus N).  The same shard + reduce inside ONE process,
for the reference's single-threaded C++ caller, is the C-ABI's tinsel_hip_group (include/tinsel_hip.h;
tinsel_amd.HipRendererGroup).

ONE implementation of the collective on GPUs: the library's own ncclReduce (tn_host_group.h rccl_reduce_accum), entered by a group's
worker threads and -- here -- by every rank of a process-per-GPU job through tinsel_hip_comm_* (`init_library_comm` + `reduce_accum(...,
renderer=r)`).  torch.distributed is the launcher's plumbing: rendezvous, the 128-byte id's broadcast, barriers, the timing's max over
ranks.  Without a library communicator (CPU tests under gloo; the one-device stand-in of bench.py) `reduce_accum` is torch's reduce.
"""
import numpy as np


def init_library_comm(renderer, rank, world, group=None):
    """Collective over the torch.distributed group: rank 0 makes an RCCL unique id, torch carries it to the other ranks, every rank enters
    tinsel_hip_comm_init on its renderer.  Returns the number of ranks RCCL itself reports for the communicator (ncclCommCount) when
    EVERY rank succeeded, else 0 with the communicator dropped everywhere -- so the ranks agree on which reduce they will enter."""
    import torch
    import torch.distributed as dist
    ids = [renderer.comm_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(ids, src=0, group=group)
    ok, seen = 1, 0
    try:
        renderer.comm_init(ids[0], rank, world)
        seen = renderer.comm_size()
        ok = 1 if seen == world else 0
    except Exception:
        ok = 0
    if world > 1:
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        ok = int(flag.item())
    return seen if ok else 0


def owned_mask(width, height, rank, world, tile=32):
    """Boolean [H,W] mask of the pixels whose camera paths rank `rank` generates."""
    if world <= 1:
        return np.ones((height, width), bool)
    tiles_x = (width + tile - 1)//tile
    jj, ii = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
    t = (jj//tile)*tiles_x + (ii//tile)
    return (t % world) == rank


def reduce_accum(accum, dst=0, group=None, out=None, renderer=None, rank=None):
    """The one collective of the path: the sum over ranks of the per-rank accumulators [H,W,4], returned on rank `dst`
    (None elsewhere).  `renderer` with a library communicator (init_library_comm) and `accum` its device accumulator: the library's own
    ncclReduce on the current stream, into `out` (required on `dst`), then a wait for that stream.  `accum` itself is left untouched on every rank -- it keeps the rank's OWN partial sums since
    Init, so a later render + reduce_accum cannot count earlier samples twice (an in-place reduce would leave rank
    dst holding everyone's samples, and the gloo backend also overwrites the non-dst inputs).  The price is one
    accumulator-sized scratch tensor: `out` (same shape, dtype and device) when the caller keeps one -- it is filled
    and returned for any number of ranks, one included -- else a new one per call (one rank: `accum` itself).  With the `nccl` backend (RCCL) the copy and the reduce are enqueued on the current stream, nothing
    synchronises; a device tensor under a CPU backend (`gloo`: the one-device stand-in of bench.py) travels through
    host memory."""
    import torch.distributed as dist
    if renderer is not None and renderer.comm_size() > 0:
        import torch
        me = rank if rank is not None else (dist.get_rank(group) if dist.is_initialized() else 0)
        if me == dst and out is None:
            out = torch.empty_like(accum)
        renderer.comm_reduce_accum(out.data_ptr() if (out is not None and me == dst) else None, dst, torch.cuda.current_stream().cuda_stream)
        return out if me == dst else None
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        # one rank: the sum is the rank's own accumulator; a caller's `out` is filled like in the multi-rank paths
        if out is None:
            return accum
        out.copy_(accum)
        return out
    backend = dist.get_backend(group)
    if accum.is_cuda and backend != "nccl":
        host = accum.cpu()
        dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM, group=group)
        if dist.get_rank(group) != dst:
            return None
        if out is None:
            return host.to(accum.device)
        out.copy_(host)
        return out
    if out is None:
        total = accum.clone()
    else:
        total = out
        total.copy_(accum)
    dist.reduce(total, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return total if dist.get_rank(group) == dst else None
