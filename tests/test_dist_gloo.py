"""N > 1 path on CPU: world_size-2 (and 3) gloo processes shard the framebuffer by interleaved row tiles,
'render' their tiles (the oracle's counter-RNG integrator stands in for the GPU here -- it is keyed by
global pixel index exactly like the HIP kernel), gather to rank 0 and must reproduce the unsharded image."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, tile_rows, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from spath_amd import scene, view
    from spath_amd.dist import RowTilePlan, gather_to_root
    t, m = scene.open_clutter(60)
    rays = view.Camera(w, h).get_viewport()
    plan = RowTilePlan(w, h, world, tile_rows)
    ids = plan.pixel_ids(rank)
    base, tile_px, stride = plan.shard(rank)
    # the sphip_shard formula of include/spath_hip.h
    k = np.arange(ids.size)
    assert np.array_equal(base + (k // tile_px) * stride + (k % tile_px), ids)
    local = np.zeros((ids.size, 4), dtype=np.uint8)
    # render each tile of this rank with the global pixel index as RNG key
    pos = 0
    for tile in plan.tiles_of(rank):
        p0 = tile * plan.tile_px
        n = min(plan.tile_px, plan.npix - p0)
        img, _, _ = O.render_counter(rays, t, m, 2, seed=9, pix0=p0, npix=n, workers=1)
        local[pos:pos + n] = img
        pos += n
    full = gather_to_root(torch.from_numpy(local), plan, rank)
    if rank == 0:
        np.save(out_path, full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h,tile_rows", [(2, 40, 30, 4), (3, 33, 20, 3), (2, 16, 5, 8)])
def test_sharded_equals_unsharded(tmp_path, O, world, w, h, tile_rows):
    from spath_amd import scene, view
    out = os.path.join(tmp_path, "img.npy")
    mp.spawn(_worker, args=(world, _free_port(), w, h, tile_rows, out), nprocs=world, join=True)
    got = np.load(out)
    t, m = scene.open_clutter(60)
    want, _, _ = O.render_counter(view.Camera(w, h).get_viewport(), t, m, 2, seed=9)
    assert np.array_equal(got, want)


def test_plan_covers_every_pixel_once():
    from spath_amd.dist import RowTilePlan
    for (w, h, g, r) in [(1920, 1080, 8, 8), (3840, 2160, 8, 16), (17, 13, 4, 5), (8, 3, 8, 1), (5, 2, 4, 8)]:
        plan = RowTilePlan(w, h, g, r)
        ids = np.concatenate([plan.pixel_ids(k) for k in range(g)])
        assert np.array_equal(np.sort(ids), np.arange(w * h))
        assert sum(plan.n_rays(k) for k in range(g)) == w * h
        # balance: no rank holds more than one tile more than another
        n = [plan.n_rays(k) for k in range(g)]
        assert max(n) - min(n) <= plan.tile_px


def test_balanced_tile_rows():
    """bench.py's default tile height: every rank gets the same number of pixels when the image allows it."""
    from spath_amd.dist import RowTilePlan, balanced_tile_rows
    for h, g in [(1080, 1), (1080, 2), (1080, 4), (1080, 8), (2160, 8), (720, 8)]:
        tr = balanced_tile_rows(h, g)
        plan = RowTilePlan(64, h, g, tr)
        assert 1 <= tr <= 8 and len({plan.n_rays(r) for r in range(g)}) == 1, (h, g, tr)
    assert balanced_tile_rows(1081, 8) == 8            # nothing balances a prime height: keep the default
