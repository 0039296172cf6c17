"""CPU, world_size 2 over gloo: the N>1 host path -- tile partition, gather of the compact tile
buffers to rank 0, K8's un-permutation (numpy statement) and the ray-stat reduction."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, q, async_op=False):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from chameleonrt_amd import multi_gpu as mg
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ntx, _ = mg.num_tiles(w, h)
    tiles = mg.local_tiles(w, h, rank, world)
    slab = mg.padded_tiles(w, h, world) * mg.TILE * mg.TILE
    buf = np.zeros(slab, np.uint32)
    # what K5 writes: tile-major, row-major inside the tile; value = global pixel id (or 0 off-image)
    for lt, t in enumerate(tiles):
        ty, tx = divmod(t, ntx)
        yy, xx = np.mgrid[0:mg.TILE, 0:mg.TILE]
        gx, gy = tx * mg.TILE + xx, ty * mg.TILE + yy
        val = np.where((gx < w) & (gy < h), gy * w + gx + 1, 0).astype(np.uint32)
        buf[lt * mg.TILE ** 2:(lt + 1) * mg.TILE ** 2] = val.reshape(-1)
    if async_op:  # what bench.py does: gather of frame f in flight while frame f+1 is traced, waited on before f+2
        gathered, work = mg.gather_tile_buffers(torch.from_numpy(buf.view(np.int32)), async_op=True)
        work.wait()
    else:
        gathered = mg.gather_tile_buffers(torch.from_numpy(buf.view(np.int32)))
    rays, ms = mg.reduce_ray_stats(1000 * (rank + 1), 5.0 + rank)
    if rank == 0:
        img = mg.assemble_numpy(gathered.numpy().view(np.uint32), w, h, world)
        q.put((img, rays, ms))
    else:
        assert gathered is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("async_op", [False, True], ids=["blocking", "async"])
@pytest.mark.parametrize("size", [(300, 200), (128, 64)])
def test_gather_and_assemble_world2(size, async_op):
    import torch.multiprocessing as mp
    w, h = size
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, w, h, q, async_op)) for r in range(2)]
    for p in procs:
        p.start()
    img, rays, ms = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    expect = (np.arange(w * h, dtype=np.uint32) + 1).reshape(h, w)
    assert np.array_equal(img, expect)
    assert rays == 3000 and ms == 6.0


def test_partition_covers_every_tile_once():
    from chameleonrt_amd import multi_gpu as mg
    for (w, h) in ((1920, 1080), (3840, 2160), (300, 200)):
        ntx, nty = mg.num_tiles(w, h)
        for world in (1, 2, 4, 8):
            seen = sorted(t for r in range(world) for t in mg.local_tiles(w, h, r, world))
            assert seen == list(range(ntx * nty))
            assert all(len(mg.local_tiles(w, h, r, world)) <= mg.padded_tiles(w, h, world) for r in range(world))
