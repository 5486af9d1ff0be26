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

One process per GPU; launched by torch.distributed.run.
"""
import numpy as np


def owned_mask(width, height, rank, world, tile=32):
    """Boolean [H,W] mask of the pixels whose camera paths rank `rank` generates."""
    if world <= 1:
        return np.ones((height, width), bool)
    tiles_x = (width + tile - 1)//tile
    jj, ii = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
    t = (jj//tile)*tiles_x + (ii//tile)
    return (t % world) == rank


def reduce_accum(accum, dst=0, group=None):
    """Sum the per-rank accumulators into rank `dst` (in place on `dst`).  `accum`: torch tensor [H,W,4]."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return accum
    dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return accum


class ShardedRenderer:
    """Renderer-shaped wrapper: `init` / `render` on this rank's shard, `gather()` = the reduce.

    `renderer` is anything with set_shard / init / render_async (the HipRenderer); `make_accum(h, w)`
    returns the torch tensor that backs the accumulator on this rank's device."""

    def __init__(self, renderer, rank, world, tile=32):
        self.renderer = renderer
        self.rank, self.world, self.tile = rank, world, tile
        renderer.set_shard(rank, world, tile)
        self.accum = None

    def init(self, width, height, accum):
        self.accum = accum
        self.renderer.init(width, height, accum_tensor=accum)

    def render(self, camera, options, passes=1, stream=None):
        self.renderer.render_async(camera, options, passes=passes, stream=stream)

    def gather(self, dst=0):
        """The one collective of the path: framebuffer sum-reduce to `dst`."""
        return reduce_accum(self.accum, dst=dst)
