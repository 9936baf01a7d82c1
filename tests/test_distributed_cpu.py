"""The N > 1 path on CPU: two gloo ranks shard a frame by tiles and rank 0 gathers it."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, w, h, tile, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import raytracer_amd as rt
        from raytracer_amd import system
        o = rt.Options(width=w, height=h, tile_size=tile)
        sizes = [system.shard_pixels(o, r, world) for r in range(world)]
        m = system.shard_map(o, rank, world)
        # stand-in for the rendered tiles: a value that names the pixel and the rank
        shard = torch.zeros(max(sizes) * 3, dtype=torch.float64)
        v = shard.view(-1, 3)
        px = torch.from_numpy(m)
        valid = px >= 0
        v[: len(m)][valid, 0] = px[valid].double()
        v[: len(m)][valid, 1] = float(rank)
        v[: len(m)][valid, 2] = (px[valid] % 7).double()
        full = system.gather_frame(o, shard, rank, world)
        if rank == 0:
            img = full.view(h * w, 3).numpy()
            ok = np.array_equal(img[:, 0], np.arange(h * w, dtype=np.float64))
            ok &= np.array_equal(img[:, 2], (np.arange(h * w) % 7).astype(np.float64))
            # tile k belongs to rank k % world
            tiles_x = (w + tile - 1) // tile
            ys, xs = np.divmod(np.arange(h * w), w)
            owner = ((ys // tile) * tiles_x + xs // tile) % world
            ok &= np.array_equal(img[:, 1], owner.astype(np.float64))
            q.put(bool(ok))
        else:
            assert full is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("w,h,tile", [(100, 70, 16), (64, 64, 64)])
def test_two_rank_tile_gather(w, h, tile):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, w, h, tile, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
