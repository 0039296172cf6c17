"""Image-tile split across the GPUs of one node + framebuffer gather.

New functionality (the reference is single-device): one process per GPU, each rendering the
64x64 tiles with ``tile_id % world == rank`` from a full scene replica; no exchange while
tracing; one collective per displayed frame gathers the compact RGBA8 tile buffers on rank 0
(RCCL over xGMI when the process group is "nccl", gloo on CPU in the tests) where kernel K8
un-permutes them into the row-major image. The payload is <= 4 MiB per GPU at 4K, so a direct
gather to the root is latency-bound and no ring is needed.
"""
from __future__ import annotations

import numpy as np

TILE = 64


def num_tiles(width: int, height: int):
    ntx = width // TILE + (1 if width % TILE else 0)
    nty = height // TILE + (1 if height % TILE else 0)
    return ntx, nty


def local_tiles(width: int, height: int, rank: int, world: int):
    """Tile ids rendered by `rank` (round-robin: neighbouring tiles go to different GPUs, so
    expensive image regions are spread over all of them)."""
    ntx, nty = num_tiles(width, height)
    return list(range(rank, ntx * nty, world))


def padded_tiles(width: int, height: int, world: int) -> int:
    ntx, nty = num_tiles(width, height)
    return (ntx * nty + world - 1) // world


def gather_tile_buffers(local, group=None, dst: int = 0, async_op: bool = False):
    """Gather every rank's compact tile buffer (1-D uint8/int32 tensor, same length on all
    ranks) to `dst`. Returns the (world * n) tensor on dst, None elsewhere. async_op: returns
    (tensor or None, work handle) instead and does not wait -- the core double-buffers the tile
    buffer by frame parity, so the gather of frame f may run while frame f+1 is traced; wait on the
    handle before frame f+2 is rendered."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    out, parts = None, None
    if rank == dst:
        import torch
        out = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
        parts = list(out.chunk(world))
    work = dist.gather(local, parts, dst=dst, group=group, async_op=async_op)
    return (out, work) if async_op else out


def reduce_ray_stats(rays: int, ms: float, group=None, device="cpu"):
    """Whole-job ray count (sum) and frame time (max over ranks): REPORT_RAY_STATS for N GPUs."""
    import torch
    import torch.distributed as dist
    r = torch.tensor([float(rays)], dtype=torch.float64, device=device)
    t = torch.tensor([float(ms)], dtype=torch.float64, device=device)
    dist.all_reduce(r, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(r.item()), float(t.item())


def assemble_numpy(gathered: np.ndarray, width: int, height: int, world: int) -> np.ndarray:
    """Host statement of kernel K8 (used by the CPU tests): gathered[rank][local_tile][64*64]
    uint32 -> (height, width) uint32."""
    ntx, _ = num_tiles(width, height)
    slab = padded_tiles(width, height, world) * TILE * TILE
    g = gathered.reshape(world, slab)
    y, x = np.mgrid[0:height, 0:width]
    tile = (y // TILE) * ntx + x // TILE
    return g[tile % world, (tile // world) * TILE * TILE + (y % TILE) * TILE + (x % TILE)]
